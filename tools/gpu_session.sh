#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  Session r4e: blind phase A, trimmed sample address math, lane activity.
export TMPDIR=/tmp
OUT=gpurun_out/r4e
mkdir -p $OUT
timeout -s KILL 900 python -m pytest tests -q -m gpu 2>&1 | tail -150 > $OUT/pytest.txt; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest.txt
timeout -s KILL 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>$OUT/bench_driver.err > $OUT/bench_driver.json
python - $OUT/bench_driver.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
oc = d["one_call"]
print("value %.1f" % d["value"], "repeats", [round(x) for x in d["repeats"]], "bulk frac %.3f valu %.3f" % (d["roofline"]["bulk_kernel_frac"], d["roofline"]["secondary_roofs"]["valu_issue"]["frac"]))
print("one_call", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in oc.items() if k != "what"})
PY
timeout -s KILL 120 python tools/trace_c3.py > $OUT/round_trace_c3.txt 2>&1; grep -E "phase|total|optimise launch" $OUT/round_trace_c3.txt | head -60
MI_DMRECON_LIB=$PWD/build/libmi_dmrecon_act.so timeout -s KILL 200 python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from mve_amd import api
from mve_amd.synth import CONFIGS, make_scene
cfg = CONFIGS["C3"]; sc = make_scene(cfg["params"])
ctx = api.Context(0); ctx.load_scene(sc)
st = api.Settings(scale=cfg["scale"], nrReconNeighbors=cfg["local_neighbors"])
for refs in (list(range(20)), list(range(20)) * 5):
    ctx.reconstruct(st, refs, want_normal=False)
    s = ctx.last_stats
    print("views %d: patch-turns %d wave-turns x16 %d -> lane activity %.3f ; n_pass/4 %d" % (len(refs), s["n_patch_turns"], s["n_wave_turns"], s["n_patch_turns"] / max(s["n_wave_turns"], 1), s["n_pass"] // 4))
os.environ["MI_DMRECON_ONE_LAUNCH"] = "0"
ctx.reconstruct(st, list(range(20)), want_normal=False); s = ctx.last_stats
print("two-launch rounds only, 20 views: lane activity %.3f" % (s["n_patch_turns"] / max(s["n_wave_turns"], 1)))
PY
