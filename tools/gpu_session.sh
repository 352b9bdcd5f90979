#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): with both slots free a leader takes half of the pending calls.
TAG=${1:-r3ad}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
show() {
  python - $1 $2 <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); t = d['roofline']['per_kernel']['k_tail + k_front (tail rounds)']; b = d['roofline']['per_kernel']['k_optimize<1> (host-visible rounds)']
    print('%-14s' % sys.argv[2], round(d['value'], 1), 'maps/s', 'ms/step', round(d['ms_per_step'], 2), 'bulk ms', round(b['avg_launch_ms'] * b['launches'] / d['steps'], 2), 'frac', round(d['roofline']['bulk_kernel_frac'], 3), 'k_front', round(t['k_front_ms'] / d['steps'], 2), d['config'].get('library_batch_log')[:6])
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
drv() { L=$1; shift; env "$@" timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-call 2>/dev/null > $OUT/drv_$L.json; show $OUT/drv_$L.json drv_$L; }
dfl() { L=$1; shift; env "$@" timeout -s KILL 300 python bench.py --no-cpu-baseline --no-one-call 2>/dev/null > $OUT/dfl_$L.json; show $OUT/dfl_$L.json dfl_$L; }
sml() { L=$1; shift; env "$@" timeout -s KILL 200 python bench.py --no-cpu-baseline --no-one-call --streams 6 --steps-per-call 1 --steps 60 --warmup 3 2>/dev/null > $OUT/sml_$L.json; show $OUT/sml_$L.json sml_$L; }
for R in 1 2 3 4 5 6; do drv new_$R; done
timeout -s KILL 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "merged or scratch" 2>&1 | tail -3
