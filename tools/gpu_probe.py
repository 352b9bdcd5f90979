"""Development probe (run through gpurun): HIP path vs CPU oracle on a small synthetic scene."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mve_amd.synth import SynthParams, make_scene
from mve_amd import api
from oracle import oracle as orc

W, H, NV = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (320, 240, 5))]
scale = int(sys.argv[4]) if len(sys.argv) > 4 else 0
p = SynthParams(n_views=NV, width=W, height=H, n_features=600)
sc = make_scene(p)
print("devices", api.device_count())
ctx = api.Context(0)
t = time.time(); ctx.load_scene(sc); print("load_scene %.3fs" % (time.time() - t))
S = orc.OracleScene(sc)
# pyramid
for lvl in range(ctx.num_levels(0)):
    g, pj, ipj = ctx.get_level(0, lvl)
    o, opj, oipj = S.pyramid_level(0, lvl)
    print("level", lvl, g.shape, "bytes equal", np.array_equal(g, o), "mismatch", int((g != o).sum()),
          "proj eq", np.array_equal(pj, opj), np.array_equal(ipj, oipj))
st = api.Settings(refViewNr=0, scale=scale)
ost = orc.make_settings(ref_view=0, scale=scale)
print("gvs gpu", ctx.global_view_selection(st), "oracle", S.global_vs(ost))
# patch eval
from mve_amd.synth import true_depth
lw, lh = ctx.level_size(0, scale)
img, pj, ipj = S.pyramid_level(0, scale)
for (x, y) in [(lw // 2, lh // 2), (lw // 3, lh // 3), (50, 40)]:
    # crude depth: 10
    d0 = 10.0
    ge = ctx.patch_eval(st, 0, x, y, d0)
    oe = S.patch_eval(ost, x, y, d0)
    print("eval", x, y, "master", ge["master"], oe["master"])
    print("  ncc gpu", ge["ncc"], "\n  ncc orc", oe["ncc"], "\n  ok", ge["ok"], oe["ok"], "lvl", ge["level"], oe["level"])
    okb = (ge["ok"] > 0) & (oe["ok"] > 0)
    if okb.any():
        print("  col maxdiff", np.abs(ge["col"][okb] - oe["col"][okb]).max(), "deriv maxdiff",
              np.abs(ge["deriv"][okb] - oe["deriv"][okb]).max(), "deriv scale", np.abs(oe["deriv"][okb]).max())
# patch optimize at seeds = features projected
rng = np.random.RandomState(0)
xy = np.stack([rng.randint(3, lw - 3, 200), rng.randint(3, lh - 3, 200)], 1)
hyp = np.stack([10.0 + rng.uniform(-0.3, 0.3, 200), np.zeros(200), np.zeros(200)], 1)
t = time.time(); go, gl = ctx.patch_optimize(st, 0, xy, hyp); print("gpu patch_opt %.3fs" % (time.time() - t))
t = time.time(); oo, ol = S.patch_optimize(ost, xy, hyp); print("orc patch_opt %.3fs" % (time.time() - t))
both = (go[:, 0] > 0) & (oo[:, 0] > 0)
print("conf>0: gpu", (go[:, 0] > 0).sum(), "orc", (oo[:, 0] > 0).sum(), "both", both.sum())
if both.any():
    print("  rel depth diff median %.3g p99 %.3g max %.3g" % tuple(np.percentile(np.abs(go[both, 1] - oo[both, 1]) / oo[both, 1], [50, 99, 100])))
    print("  conf diff median %.3g max %.3g" % (np.median(np.abs(go[both, 0] - oo[both, 0])), np.abs(go[both, 0] - oo[both, 0]).max()))
    print("  same local set frac", (gl[both] == ol[both]).all(1).mean(), "iters gpu/orc mean", go[both, 7].mean(), oo[both, 7].mean())
# full reconstruct
t = time.time(); r = ctx.reconstruct(st, [0])[0]; tg = time.time() - t
print("gpu reconstruct %.3fs stats %s" % (tg, ctx.last_stats))
t = time.time(); o = S.reconstruct(ost); to = time.time() - t
print("orc reconstruct %.3fs stats %s" % (to, o["stats"]))
mg, mo = r["depth"] > 0, o["depth"] > 0
both = mg & mo
print("filled gpu %d orc %d IoU %.4f" % (mg.sum(), mo.sum(), (mg & mo).sum() / max((mg | mo).sum(), 1)))
rel = np.abs(r["depth"][both] - o["depth"][both]) / o["depth"][both]
print("rel depth diff median %.3g p90 %.3g p99 %.3g max %.3g" % tuple(np.percentile(rel, [50, 90, 99, 100])))
cd = np.abs(r["conf"][both] - o["conf"][both])
print("conf diff median %.3g p99 %.3g max %.3g" % tuple(np.percentile(cd, [50, 99, 100])))
gt = true_depth(p, sc.cameras[0], lw, lh, ppoint=None)
print("median abs err vs truth: gpu %.4g orc %.4g" % (np.median(np.abs(r["depth"][mg] - gt[mg])), np.median(np.abs(o["depth"][mo] - gt[mo]))))
# batch of all views
t = time.time(); rr = ctx.reconstruct(st, list(range(NV))); tb = time.time() - t
print("gpu batch of %d: %.3fs stats %s" % (NV, tb, ctx.last_stats))
