#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the measurement session of the day (tests + bench lines + probes in one call).
TAG=${1:-r3c}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 200 python tools/patch_probe.py > $OUT/probe.txt 2>&1
tail -45 $OUT/probe.txt
one() {  # label, env...
  L=$1; shift
  env "$@" timeout -s KILL 120 python bench.py --no-cpu-baseline --no-one-call --streams 1 --steps-per-call 1 --steps 10 --warmup 2 2>/dev/null > $OUT/b1_$L.json
  python - $OUT/b1_$L.json $L <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d['roofline']['per_kernel']
    t = r['k_tail + k_front (tail rounds)']; b = r['k_optimize<1> (host-visible rounds)']
    print('%-12s' % sys.argv[2], round(d['value'], 1), 'maps/s  ms/step', round(d['ms_per_step'], 2), ' bulk ms', round(b['avg_launch_ms'] * b['launches'] / d['steps'], 2), 'launches', b['launches'] // d['steps'], 'frac', round(b['frac'], 4),
          ' k_tail ms', round(t['k_tail_ms'] / d['steps'], 2), 'launches', t['k_tail_launches'] // d['steps'], ' k_front ms', round(t['k_front_ms'] / d['steps'], 2), 'rounds', t['k_front_rounds_slowest_view'] // d['steps'], 'attempts', t['k_front_attempts'] // d['steps'])
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
one f0 MI_DMRECON_FRONT=0
one f0_t32k MI_DMRECON_FRONT=0 MI_DMRECON_TAIL_THRESHOLD=32768
one f0_t64k MI_DMRECON_FRONT=0 MI_DMRECON_TAIL_THRESHOLD=65536
one f0_t128k MI_DMRECON_FRONT=0 MI_DMRECON_TAIL_THRESHOLD=131072
one f0_t256k MI_DMRECON_FRONT=0 MI_DMRECON_TAIL_THRESHOLD=262144
one f0_s4096 MI_DMRECON_FRONT=0 MI_DMRECON_SPECULATE=4096
