#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  Round-2 measurement session.  Output: gpurun_out/$TAG/.
TAG=${1:-s12}
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
echo "== view selection on the device: tests"
timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q --maxfail=4 -k "view_selection or c3_inv or batch_equals" 2>&1 | tail -5
echo "== view selection: host vs device time, C3 and C5 geometry (small images, same cameras / features)"
python - <<'PY' 2>&1 | tail -12
import os, time, numpy as np
from mve_amd import api
from mve_amd.synth import CONFIGS, SynthParams, make_scene
for name, w, h in (("C3", 480, 270), ("C5", 504, 378)):
    cfg = CONFIGS[name]; p = SynthParams(**{**cfg["params"].__dict__, "width": w, "height": h})
    sc = make_scene(p); ctx = api.Context(0); ctx.load_scene(sc)
    st = api.Settings(scale=0, nrReconNeighbors=cfg["local_neighbors"]); refs = list(range(p.n_views))
    for mode in ("0", "1"):
        os.environ["MI_DMRECON_GVS_DEVICE"] = mode
        ts = []
        for k in range(4):
            ctx.reconstruct(st, refs, want_normal=False); s = ctx.last_stats
            ts.append((s["ms_plan_gvs"], s["ms_plan_seeds"], s["ms_total"]))
        print(name, p.n_views, "views", p.n_features, "features; device" if mode == "1" else "features; host  ", "gvs/seeds/total ms:", ["%.2f/%.2f/%.1f" % t for t in ts])
    sel = {}
    for mode in ("0", "1"):
        os.environ["MI_DMRECON_GVS_DEVICE"] = mode
        sel[mode] = [ctx.global_view_selection(api.Settings(refViewNr=r), r) for r in refs]
    print(name, "identical selections:", sel["0"] == sel["1"])
PY
B1="python bench.py --steps 8 --warmup 2 --streams 1 --steps-per-call 1 --no-cpu-baseline"
BD="python bench.py --steps 30 --warmup 2 --no-cpu-baseline"
for G in 0 1; do
  echo "== gvs_device=$G: 1 stream"; MI_DMRECON_GVS_DEVICE=$G timeout -s KILL 240 $B1 > $OUT/b1_$G.json 2> $OUT/b1_$G.err; show $OUT/b1_$G.json; tail -1 $OUT/b1_$G.err | cut -c1-200
  echo "== gvs_device=$G: default"; MI_DMRECON_GVS_DEVICE=$G timeout -s KILL 300 $BD > $OUT/bd_$G.json 2> $OUT/bd_$G.err; show $OUT/bd_$G.json
done
echo "== C5 full"; timeout -s KILL 500 python tools/c5_full.py > $OUT/c5_full.json 2> $OUT/c5_full.err; python -c "
import json; d=json.load(open('$OUT/c5_full.json')); print(d['one_call_all_views']); print(d['one_call_view_selection_on_host']); print(d['parity_view2_vs_oracle'])"
