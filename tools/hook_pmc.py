"""Known amount of sampling work through the parity hook (one k_optimize<1> launch of 16384 patches = 1024
wavefronts), for PMC runs: prints the number of wave-level samples so that counters can be put per sample.
Patches are taken (a) sorted in raster order = neighbouring pixels per wavefront, (b) shuffled."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mve_amd import api
from mve_amd.synth import CONFIGS, make_scene
cfg = CONFIGS["C3"]
sc = make_scene(cfg["params"])
ctx = api.Context(0); ctx.load_scene(sc)
st = api.Settings(refViewNr=0, scale=cfg["scale"])
r = ctx.reconstruct(st, [0], want_views=True)[0]
ys, xs = np.nonzero(r["conf"] > 0.9)
n = 16384
for label, order in (("raster", np.arange(len(xs))[:n]), ("shuffled", np.random.RandomState(0).permutation(len(xs))[:n])):
    sel = order
    xy = np.stack([xs[sel], ys[sel]], 1)
    hyp = np.stack([r["depth"][ys[sel], xs[sel]] * 1.002, r["dz"][ys[sel], xs[sel], 0], r["dz"][ys[sel], xs[sel], 1]], 1)
    loc = r["views"][ys[sel], xs[sel]]
    os.environ["MI_DMRECON_HOOK_LPV"] = "1"
    out, _ = ctx.patch_optimize(st, 0, xy, hyp, loc)
    it = out[:, 7].reshape(-1, 16)
    turns = it.max(1) + 2            # first pass + one pass per iteration + the last check, per wavefront
    print("%s: wavefronts %d, iterations mean %.2f, wave-level samples ~%d" % (label, len(turns), out[:, 7].mean(), int(turns.sum() * 25)))
