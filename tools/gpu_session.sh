#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4q
mkdir -p $OUT
for I in 1 2 3; do
timeout -s KILL 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-one-call 2>$OUT/bench$I.err > $OUT/bench$I.json
python - $OUT/bench$I.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("driver value %.1f" % d["value"], "repeats", [round(x) for x in d["repeats"]], d["config"]["library_batch_log"])
PY
done
timeout -s KILL 300 python bench.py --steps 60 --repeats 3 --no-cpu-baseline --no-one-call 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default plan value %.1f' % d['value'], [round(x) for x in d['repeats']])"
timeout -s KILL 300 python bench.py --steps 60 --repeats 3 --steps-per-call 1 --no-cpu-baseline --no-one-call 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('6 threads x 20-view calls value %.1f' % d['value'], [round(x) for x in d['repeats']], d['config']['views_per_library_batch'])"
