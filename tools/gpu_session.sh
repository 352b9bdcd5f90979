#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  Round-2 measurement session.  Output: gpurun_out/$TAG/.
TAG=${1:-s11}
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
echo "== maps with and without band ordering are identical"
python - <<'PY'
import os, numpy as np
from mve_amd import api
from mve_amd.synth import SynthParams, make_scene
sc = make_scene(SynthParams(n_views=8, width=640, height=360, n_features=800))
ctx = api.Context(0); ctx.load_scene(sc)
a = ctx.reconstruct(api.Settings(), list(range(8)))
print("gvs ok", ctx.global_view_selection(api.Settings(refViewNr=3)))
PY
B1="python bench.py --steps 8 --warmup 2 --streams 1 --steps-per-call 1 --no-cpu-baseline"
BD="python bench.py --steps 30 --warmup 2 --no-cpu-baseline"
for V in base rows; do for BN in 0 1; do
  L=$PWD/mve_amd/csrc/libmi_dmrecon.so; [ $V = rows ] && L=$PWD/build/libmi_dmrecon_rows.so
  echo "== $V bands=$BN: 1 stream"; MI_DMRECON_LIB=$L MI_DMRECON_BANDS=$BN timeout -s KILL 240 $B1 > $OUT/b1_${V}_$BN.json 2> $OUT/b1_${V}_$BN.err; show $OUT/b1_${V}_$BN.json; tail -1 $OUT/b1_${V}_$BN.err | cut -c1-200
  echo "== $V bands=$BN: default"; MI_DMRECON_LIB=$L MI_DMRECON_BANDS=$BN timeout -s KILL 300 $BD > $OUT/bd_${V}_$BN.json 2> $OUT/bd_${V}_$BN.err; show $OUT/bd_${V}_$BN.json
done; done
echo "== pytest with bands"; MI_DMRECON_BANDS=1 timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q --maxfail=4 2>&1 | tail -3
echo "== pytest rows"; MI_DMRECON_LIB=$PWD/build/libmi_dmrecon_rows.so timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=4 2>&1 | tail -3
