#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): full GPU suite + lone-call bench lines (default build, 16-wave front variant).
TAG=${1:-r3f}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $OUT/pytest.log
tail -25 $OUT/pytest.log
one() {  # label, env...
  L=$1; shift
  env "$@" timeout -s KILL 120 python bench.py --no-cpu-baseline --no-one-call --streams 1 --steps-per-call 1 --steps 10 --warmup 2 2>/dev/null > $OUT/b1_$L.json
  python - $OUT/b1_$L.json $L <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d['roofline']['per_kernel']
    t = r['k_tail + k_front (tail rounds)']; b = r['k_optimize<1> (host-visible rounds)']
    print('%-12s' % sys.argv[2], round(d['value'], 1), 'maps/s  ms/step', round(d['ms_per_step'], 2), ' bulk ms', round(b['avg_launch_ms'] * b['launches'] / d['steps'], 2), 'launches', b['launches'] // d['steps'], 'frac', round(b['frac'], 4),
          ' k_tail ms', round(t['k_tail_ms'] / d['steps'], 2), 'launches', t['k_tail_launches'] // d['steps'], ' k_front ms', round(t['k_front_ms'] / d['steps'], 2), 'rounds', t['k_front_rounds_slowest_view'] // d['steps'])
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
one dflt
one f8 MI_DMRECON_FRONT=8
one f8_w16 MI_DMRECON_FRONT=8 MI_DMRECON_LIB=$PWD/build/libmi_dmrecon_f16.so
one f16_w16 MI_DMRECON_FRONT=16 MI_DMRECON_LIB=$PWD/build/libmi_dmrecon_f16.so
one f4_w16 MI_DMRECON_FRONT=4 MI_DMRECON_LIB=$PWD/build/libmi_dmrecon_f16.so
