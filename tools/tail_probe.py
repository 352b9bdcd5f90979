"""Latency probe of the patch kernel through the parity hook (run under rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mve_amd import api
from mve_amd.synth import SynthParams, make_scene
sc = make_scene(SynthParams(n_views=8, width=640, height=360, n_features=800))
ctx = api.Context(0); ctx.load_scene(sc)
st = api.Settings(refViewNr=0)
r = ctx.reconstruct(st, [0], want_views=True)[0]
ys, xs = np.nonzero(r["conf"] > 0.9)
sel = np.random.RandomState(0).permutation(len(xs))[:4096]
xy = np.stack([xs[sel], ys[sel]], 1)
hyp = np.stack([r["depth"][ys[sel], xs[sel]] * 1.002, r["dz"][ys[sel], xs[sel], 0], r["dz"][ys[sel], xs[sel], 1]], 1)
loc = r["views"][ys[sel], xs[sel]]
for lpv in (16, 1):
    os.environ["MI_DMRECON_HOOK_LPV"] = str(lpv)
    for n in (1, 16, 256, 4096):
        for rep in range(3):
            out, _ = ctx.patch_optimize(st, 0, xy[:n], hyp[:n], loc[:n])
        print("lpv", lpv, "n", n, "conf>0", int((out[:, 0] > 0).sum()), "iters mean", out[:, 7].mean())
