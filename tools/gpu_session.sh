#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): FAST bulk kernel -- tests, then 3 vs 2 wavefronts per SIMD (zero scratch).
TAG=${1:-r3j}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 > $OUT/pytest.log
tail -6 $OUT/pytest.log
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
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); t = d['roofline']['per_kernel']['k_tail + k_front (tail rounds)']
    print('driver %-8s' % sys.argv[2], round(d['value'], 1), 'maps/s', d['config']['library_batches'], 'batches', 'bulk frac', round(d['roofline']['bulk_kernel_frac'], 4), 'k_tail ms/step', round(t['k_tail_ms'] / d['steps'], 2), 'k_front ms/step', round(t['k_front_ms'] / d['steps'], 2))
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
for R in 1 2; do
one w3_$R
one w2_$R MI_DMRECON_LIB=$PWD/build/libmi_dmrecon_w2.so
drv w3_$R
drv w2_$R MI_DMRECON_LIB=$PWD/build/libmi_dmrecon_w2.so
done
