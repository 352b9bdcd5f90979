#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): bulk-kernel occupancy variants + the multi-slot shim test.
TAG=${1:-r3e}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 300 python -m pytest tests/test_gpu_dropin_app.py -m gpu -q 2>&1 | tail -5
one() {  # label, env...
  L=$1; shift
  env "$@" timeout -s KILL 120 python bench.py --no-cpu-baseline --no-one-call --streams 1 --steps-per-call 1 --steps 10 --warmup 2 2>/dev/null > $OUT/b1_$L.json
  python - $OUT/b1_$L.json $L <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d['roofline']['per_kernel']
    t = r['k_tail + k_front (tail rounds)']; b = r['k_optimize<1> (host-visible rounds)']
    print('%-12s' % sys.argv[2], round(d['value'], 1), 'maps/s  ms/step', round(d['ms_per_step'], 2), ' bulk ms', round(b['avg_launch_ms'] * b['launches'] / d['steps'], 2), 'launches', b['launches'] // d['steps'], 'frac', round(b['frac'], 4),
          ' k_tail ms', round(t['k_tail_ms'] / d['steps'], 2), 'launches', t['k_tail_launches'] // d['steps'], ' k_front ms', round(t['k_front_ms'] / d['steps'], 2))
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
drv() {
  L=$1; shift
  env "$@" timeout -s KILL 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-call 2>/dev/null > $OUT/drv_$L.json
  python - $OUT/drv_$L.json $L <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('driver %-8s' % sys.argv[2], round(d['value'], 1), 'maps/s', d['config']['library_batches'], 'batches', 'bulk frac', round(d['roofline']['bulk_kernel_frac'], 4))
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
one w3 MI_DMRECON_FRONT=0
one w4 MI_DMRECON_FRONT=0 MI_DMRECON_LIB=$PWD/build/libmi_dmrecon_w4.so
one w2 MI_DMRECON_FRONT=0 MI_DMRECON_LIB=$PWD/build/libmi_dmrecon_w2.so
drv w3 MI_DMRECON_FRONT=0
drv w4 MI_DMRECON_FRONT=0 MI_DMRECON_LIB=$PWD/build/libmi_dmrecon_w4.so
drv w2 MI_DMRECON_FRONT=0 MI_DMRECON_LIB=$PWD/build/libmi_dmrecon_w2.so
drv w3f8 MI_DMRECON_FRONT=8
