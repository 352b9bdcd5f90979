#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  Merge front end after the adaptive gather window: its test, two default bench lines.
TAG=${1:-s36}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "concurrent_calls or batch_equals or views_end" 2>&1 | tail -3
for REP in 1 2; do
  timeout -s KILL 300 python bench.py --no-cpu-baseline > $OUT/bd_$REP.json 2> $OUT/bd_$REP.err
  python - $OUT/bd_$REP.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d['value'], 1), 'maps/s', d['config']['library_batches'], 'batches of', d['config']['views_per_library_batch'], 'views; frac', round(d['roofline']['frac'], 4))
PY
done
timeout -s KILL 200 python bench.py --no-cpu-baseline --streams 1 --steps-per-call 1 --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1 thread:', round(d['value'],1))"
