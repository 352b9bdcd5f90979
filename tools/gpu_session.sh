#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  The default bench line as the driver runs it.
TAG=${1:-s37}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s.%N); timeout -s KILL 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "wall $(python -c "import time,sys; print(round(time.time()-float(sys.argv[1]),1))" $T0) s"
python - $OUT/bench_default.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d['value'], 1), 'maps/s', 'steps', d['steps'], d['config']['library_batches'], 'batches of', d['config']['views_per_library_batch'], 'views; frac', round(d['roofline']['frac'], 4), 'bulk', round(d['roofline']['per_kernel']['k_optimize<1> (host-visible rounds)']['frac'], 4), 'cpu', d['cpu_baseline']['value'], 'parity', d['parity']['within_bounds'])
PY
