#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): scratch sets leased from the scene's pool -- the GPU suite, then the driver's command
# six times (each its own process, as the driver runs it).
TAG=${1:-r3z}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
show() {
  python - $1 $2 <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); t = d['roofline']['per_kernel']['k_tail + k_front (tail rounds)']; b = d['roofline']['per_kernel']['k_optimize<1> (host-visible rounds)']
    print('%-14s' % sys.argv[2], round(d['value'], 1), 'maps/s', 'ms/step', round(d['ms_per_step'], 2), 'bulk ms', round(b['avg_launch_ms'] * b['launches'] / d['steps'], 2), 'frac', round(d['roofline']['bulk_kernel_frac'], 3), 'k_front', round(t['k_front_ms'] / d['steps'], 2), d['config'].get('library_batch_log'))
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
for R in 1 2 3 4 5 6; do
  timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-call 2>/dev/null > $OUT/drv_$R.json; show $OUT/drv_$R.json drv_$R
done
timeout -s KILL 300 python bench.py --no-cpu-baseline --no-one-call 2>/dev/null > $OUT/dfl.json; show $OUT/dfl.json default_plan
