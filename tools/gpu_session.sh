#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  Round-2 measurement session.  Output: gpurun_out/$TAG/.
TAG=${1:-s29}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
show() { python - "$1" <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d['roofline']
    n = d['steps']
    line = '  value %.1f maps/s  ms/step %.2f  frac %.4f' % (d['value'], d['ms_per_step'], r['frac'])
    if 'per_kernel' in r:
        line += ' | ' + ' | '.join('%s: %.0f x %.4f = %.2f ms' % (k[:10], v['launches']/n, v['avg_launch_ms'], v['launches']*v['avg_launch_ms']/n) for k, v in r['per_kernel'].items())
    print(line)
except Exception as e:
    print('  (no json)', e)
PY
}
BD="python bench.py --warmup 2 --no-cpu-baseline --steps 30"
rund() { N=$1; shift
  env "$@" timeout -s KILL 300 $BD $EXTRA > $OUT/bd_$N.json 2> $OUT/bd_$N.err; echo -n "$N "; show $OUT/bd_$N.json | cut -c1-170
}
echo "== quick tests"; timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
for REP in 1 2 3; do
  EXTRA="" rund spec_$REP MI_DMRECON_SPECULATE_SHARED=1
  EXTRA="" rund auto_$REP
done
EXTRA="--streams 2" rund auto_t2
EXTRA="--streams 3" rund auto_t3
EXTRA="--streams 4" rund auto_t4
EXTRA="--streams 8 --steps 40" rund auto_t8
EXTRA="--streams 1 --steps-per-call 1 --steps 10" rund auto_1
