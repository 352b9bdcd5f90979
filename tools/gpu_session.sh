#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  Launch-size sweep of the bulk kernel on one host thread.
TAG=${1:-s31}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for SPC in 10 20; do
  timeout -s KILL 300 python bench.py --streams 1 --steps-per-call $SPC --steps $((SPC*3)) --warmup $SPC --no-cpu-baseline > $OUT/b1_spc$SPC.json 2> $OUT/b1_spc$SPC.err
  python - $OUT/b1_spc$SPC.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d['roofline']
print(d['config']['steps_per_call'], 'steps/call:', round(d['value'], 1), 'maps/s', {k[:12]: (round(v['avg_launch_ms'], 3), round(v['frac'], 4)) for k, v in r['per_kernel'].items()})
PY
done
