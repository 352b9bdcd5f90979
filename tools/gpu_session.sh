#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): wide-view-set test + front threshold sweep at the driver's call plan.
TAG=${1:-r3h}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "wide" 2>&1 | tail -30 > $OUT/pytest.log
tail -12 $OUT/pytest.log
drv() {
  L=$1; shift
  env "$@" timeout -s KILL 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-call 2>/dev/null > $OUT/drv_$L.json
  python - $OUT/drv_$L.json $L <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); t = d['roofline']['per_kernel']['k_tail + k_front (tail rounds)']
    print('driver %-8s' % sys.argv[2], round(d['value'], 1), 'maps/s', d['config']['library_batches'], 'batches', 'bulk frac', round(d['roofline']['bulk_kernel_frac'], 4), 'k_tail ms/step', round(t['k_tail_ms'] / d['steps'], 2), 'k_front ms/step', round(t['k_front_ms'] / d['steps'], 2))
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
for R in 1 2; do
drv f64_$R MI_DMRECON_FRONT=64
drv f128_$R MI_DMRECON_FRONT=128
drv f256_$R MI_DMRECON_FRONT=256
drv fall_$R MI_DMRECON_FRONT=1000000
drv f32_$R MI_DMRECON_FRONT=32
done
