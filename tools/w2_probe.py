#!/usr/bin/env python3
"""GPU box: scene W2 (tests/golden/w2_wider_100views_96x72.npz) -- where the HIP path and the reference's maps differ in which
pixels get a depth, and what the reference's own patch optimisation (the restatement's hook) says about the pixels only the
HIP path fills: run with the HIP result's own hypothesis and local view set."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import scene_from_golden, GOLDEN, map_parity
from mve_amd import api
from oracle import oracle as orc

g = dict(np.load(os.path.join(GOLDEN, "w2_wider_100views_96x72.npz")))
sc = scene_from_golden(g)
ctx = api.Context(0); ctx.load_scene(sc)
S = orc.OracleScene(sc)
for tag, k, ng in (("k10n80", 10, 80), ("k16n20", 16, 20), ("k4n80", 4, 80), ("k8n80", 8, 80)):
    st = api.Settings(refViewNr=0, nrReconNeighbors=k, globalVSMax=ng)
    r = ctx.reconstruct(st, [0], want_views=True)[0]
    if tag + "_depth" in g:
        print(tag, "default (the reference's seed semantics)", map_parity(r["depth"], r["conf"], g[tag + "_depth"], g[tag + "_conf"]), flush=True)
    os.environ["MI_DMRECON_SEED_REOPT"] = "0"                   # every seed propagates at once: the rest of the probe looks at that form
    r = ctx.reconstruct(st, [0], want_views=True)[0]
    del os.environ["MI_DMRECON_SEED_REOPT"]
    if tag + "_depth" in g:
        rd, rc = g[tag + "_depth"], g[tag + "_conf"]
    else:
        o = S.reconstruct(orc.make_settings(ref_view=0, local_neighbors=k, global_max=ng)); rd, rc = o["depth"], o["conf"]
    m = map_parity(r["depth"], r["conf"], rd, rc)
    print(tag, m, "rounds", ctx.last_stats["n_rounds"], "ms", round(ctx.last_stats["ms_total"], 1), flush=True)
    ga, gb = r["depth"] > 0, rd > 0
    only_g, only_r = ga & ~gb, gb & ~ga
    ys, xs = np.nonzero(only_g)
    print("  only HIP: %d px; rows %s cols %s; conf min/med/max %s" % (only_g.sum(), (ys.min(), ys.max()) if len(ys) else None,
          (xs.min(), xs.max()) if len(xs) else None, np.round(np.percentile(r["conf"][only_g], [0, 50, 100]), 3) if len(ys) else None))
    print("  only reference: %d px" % only_r.sum())
    v = r["views"][ga]
    print("  views per filled pixel: %s" % np.bincount((v >= 0).sum(1), minlength=17))
    if len(ys):
        xy = np.stack([xs, ys], 1).astype(np.int32)
        hyp = np.stack([r["depth"][only_g], r["dz"][only_g][:, 0], r["dz"][only_g][:, 1]], 1).astype(np.float32)
        loc = r["views"][only_g]
        ost = orc.make_settings(ref_view=0, local_neighbors=k, global_max=ng)
        out, oloc = S.patch_optimize(ost, xy, hyp, loc)
        print("  the reference's PatchOptimization on those hypotheses + view sets: conf > 0 for %d of %d; |dconf| med %.4f; same views %d"
              % ((out[:, 0] > 0).sum(), len(xy), float(np.median(np.abs(out[:, 0] - r["conf"][only_g]))),
                 int((oloc[:, :loc.shape[1]] == loc).all(1).sum())))
        # ... and without a propagated set (the view selection from scratch)
        out2, oloc2 = S.patch_optimize(ost, xy, hyp, None)
        print("  ... with the view selection from scratch: conf > 0 for %d" % (out2[:, 0] > 0).sum())
        # neighbours: is an only-HIP pixel next to a reference-filled pixel?
        pad = np.pad(gb, 1)
        nb = pad[ys, xs + 1] | pad[ys + 2, xs + 1] | pad[ys + 1, xs] | pad[ys + 1, xs + 2]
        print("  next to a pixel the reference fills: %d of %d" % (nb.sum(), len(ys)))
ctx.close()
