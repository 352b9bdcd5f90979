#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): wide test; front workgroup size at the driver's plan; a lone 100-view call.
TAG=${1:-r3i}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "wide" 2>&1 | tail -30 > $OUT/pytest.log
grep -E "W1 patches|passed|failed|^E  " $OUT/pytest.log | head
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
big() {
  L=$1; shift
  env "$@" timeout -s KILL 200 python bench.py --streams 1 --steps 20 --warmup 5 --no-cpu-baseline --no-one-call 2>/dev/null > $OUT/big_$L.json
  python - $OUT/big_$L.json $L <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); t = d['roofline']['per_kernel']['k_tail + k_front (tail rounds)']
    print('1 thread x 100 views %-8s' % sys.argv[2], round(d['value'], 1), 'maps/s', 'bulk frac', round(d['roofline']['bulk_kernel_frac'], 4), 'k_tail ms/step', round(t['k_tail_ms'] / d['steps'], 2), 'k_front ms/step', round(t['k_front_ms'] / d['steps'], 2))
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
for R in 1 2; do
drv w8_$R
drv w4_$R MI_DMRECON_LIB=$PWD/build/libmi_dmrecon_fw4.so
done
big f2 MI_DMRECON_FRONT=2
big f0 MI_DMRECON_FRONT=0
big f8 MI_DMRECON_FRONT=8
big fall MI_DMRECON_FRONT=1000000
big fall_w4 MI_DMRECON_FRONT=1000000 MI_DMRECON_LIB=$PWD/build/libmi_dmrecon_fw4.so
