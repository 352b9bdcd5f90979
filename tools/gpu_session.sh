#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  Round-2 measurement session.  Output: gpurun_out/$TAG/.
TAG=${1:-s23}
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
  env "$@" timeout -s KILL 300 $BD $EXTRA > $OUT/bd_$N.json 2> $OUT/bd_$N.err; echo -n "$N "; show $OUT/bd_$N.json | cut -c1-150
}
for REP in 1 2 3; do
  EXTRA="" rund base_$REP
  EXTRA="" rund bulkprio_$REP MI_DMRECON_TAIL_PRIORITY=-1
  EXTRA="" rund thr4096_$REP MI_DMRECON_TAIL_THRESHOLD=4096
done
EXTRA="" rund bulkprio_notoken MI_DMRECON_TAIL_PRIORITY=-1 MI_DMRECON_BULK_TOKEN=0
EXTRA="" rund thr2048 MI_DMRECON_TAIL_THRESHOLD=2048
EXTRA="" rund thr32768 MI_DMRECON_TAIL_THRESHOLD=32768
EXTRA="" rund gvsdev MI_DMRECON_GVS_DEVICE=1
EXTRA="--streams 3" rund gvsdev_t3 MI_DMRECON_GVS_DEVICE=1
