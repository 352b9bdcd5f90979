#!/usr/bin/env python3
"""GPU box: the sixteen-slot layout at the size of BASELINE config 3 (20 views 1920 x 1080, scale 2) -- two reference views with ten
and with sixteen local views out of the 19 global ones: that the call ends (host-visible rounds only), what it fills, how long it
takes, and that every filled pixel carries exactly K ascending views.  With four local views for comparison."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mve_amd import api
from mve_amd.synth import CONFIGS, make_scene

cfg = CONFIGS["C3"]
sc = make_scene(cfg["params"])
ctx = api.Context(0); ctx.load_scene(sc)
for k in (4, 10, 16):
    st = api.Settings(scale=cfg["scale"], nrReconNeighbors=k)
    for rep in range(2):
        t0 = time.perf_counter()
        res = ctx.reconstruct(st, [0, 7], want_views=True)
        dt = time.perf_counter() - t0
    s = ctx.last_stats
    for v, r in zip((0, 7), res):
        f = r["conf"] > 0
        vv = r["views"][f]
        ok = ((vv >= 0).sum(1) == k).all() and (np.diff(vv[:, :k], axis=1) > 0).all()
        print("K = %2d view %d: filled %.3f, exactly K ascending views per pixel: %s, conf median %.3f" % (k, v, f.mean(), ok, float(np.median(r["conf"][f])) if f.any() else 0))
    print("   call %.1f ms, rounds %d, launches %d, truncated %d, patches %d" % (1e3 * dt, s["n_rounds"], s["n_launches"], s["truncated"], s["n_patch"]), flush=True)
ctx.close()
