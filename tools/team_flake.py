#!/usr/bin/env python3
"""How often does the sequence of tests/test_gpu_parity.py::test_front_team_gives_up_and_the_views_finish differ from the
one-workgroup-per-view maps, and where?  (A development probe: N repetitions of [teams of 8 with a member that vanishes ->
fallback] and [teams of 8 that write through their L2s], fault as in the test.)
usage (on the GPU box): python tools/team_flake.py [N] [fault] [wt]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mve_amd import api
from conftest import scene_from_golden, GOLDEN

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
fault = sys.argv[2] if len(sys.argv) > 2 else "0:7"
wt = sys.argv[3] if len(sys.argv) > 3 else "1"
os.environ.update(MI_DMRECON_VIEW_HANDOVER="1000000000", MI_DMRECON_FRONT="1000000", MI_DMRECON_TEAM_WAIT_US="3000")
ctx = api.Context(0)
scenes = [(scene_from_golden(dict(np.load(os.path.join(GOLDEN, "g1_5views_160x120.npz")))), [0, 1, 2, 3, 4]),
          (scene_from_golden(dict(np.load(os.path.join(GOLDEN, "h1_hard_9views_208x156.npz")))), list(range(9)))]
bad = {"fallback": 0, "again": 0}
for it in range(N):
    for si, (scene, refs) in enumerate(scenes):
        ctx.load_scene(scene)
        os.environ["MI_DMRECON_FRONT_TEAM"] = "1"
        ref = ctx.reconstruct(api.Settings(), refs)
        os.environ["MI_DMRECON_FRONT_TEAM"] = "8"
        stages = []
        if fault != "none":
            os.environ["MI_DMRECON_DEBUG_FRONT_FAULT"] = fault
            stages.append(("fallback", ctx.reconstruct(api.Settings(), refs), dict(ctx.last_stats)))
            del os.environ["MI_DMRECON_DEBUG_FRONT_FAULT"]
        if wt != "none":
            os.environ["MI_DMRECON_DEBUG_TEAM_WT"] = wt
        stages.append(("again", ctx.reconstruct(api.Settings(), refs), dict(ctx.last_stats)))
        os.environ.pop("MI_DMRECON_DEBUG_TEAM_WT", None)
        for name, got, st in stages:
            for v, (a, b) in enumerate(zip(got, ref)):
                d = (a["depth"] != b["depth"]) | (a["conf"] != b["conf"])
                if d.any():
                    bad[name] += 1
                    ys, xs = np.nonzero(d)
                    print("iteration %d scene %d stage %s view %d: %d pixels differ, first (%d, %d): depth %r vs %r, conf %r vs %r; fallbacks %d team %d" % (
                        it, si, name, v, d.sum(), xs[0], ys[0], a["depth"][ys[0], xs[0]], b["depth"][ys[0], xs[0]],
                        a["conf"][ys[0], xs[0]], b["conf"][ys[0], xs[0]], st["front_fallbacks"], st["front_team"]), flush=True)
print("%d iterations, fault %s, wt %s: mismatching views %r" % (N, fault, wt, bad))
