"""Maps of the C3 scene (a batch of `copies` x the 20 views: the plan whose large rounds run the FAST kernel) from two builds of the
library, compared pixel by pixel: python tools/maps_diff.py dump <out.npz> (under MI_DMRECON_LIB=...), then
python tools/maps_diff.py cmp a.npz b.npz.    (GPU box)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if sys.argv[1] == "dump":
    from mve_amd import api
    from mve_amd.synth import CONFIGS, make_scene
    cfg = CONFIGS["C3"]; p = cfg["params"]
    sc = make_scene(p, gpu=True)
    ctx = api.Context(0); ctx.load_scene(sc)
    st = api.Settings(scale=cfg["scale"], nrReconNeighbors=cfg["local_neighbors"])
    res = ctx.reconstruct(st, list(range(p.n_views)) * 5, want_normal=True, want_views=True)
    np.savez(sys.argv[2], **{"%s_%d" % (k, i): r[k] for i, r in enumerate(res[:p.n_views]) for k in ("depth", "conf", "dz", "normal", "views")})
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    out = {}
    for k in a.files:
        x, y = a[k], b[k]
        if np.array_equal(x, y):
            continue
        d = np.abs(x.astype(np.float64) - y.astype(np.float64))
        nz = d > 0
        out[k] = {"differing": int(nz.sum()), "of": int(nz.size), "max_abs": float(d.max()),
                  "max_rel": float((d[nz] / np.maximum(np.abs(x.astype(np.float64))[nz], 1e-30)).max())}
    print(json.dumps({"differing_arrays": len(out), "of_arrays": len(a.files), "detail": dict(list(out.items())[:12])}))
