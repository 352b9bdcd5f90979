"""A digest of the maps of one call over all reference views of a config's scene (and of a merged batch of `copies` such
calls' worth of views): two builds of the library give the same digests iff their maps are the same bits.  Prints one JSON object
(with the window counters of a -DMI_LDS_WINDOW build: wavefront-passes on LDS windows / on global gathers).

    MI_DMRECON_LIB=build/libmi_dmrecon_<variant>.so python tools/maps_digest.py [CONFIG] [copies]      (GPU box)
"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mve_amd import api
from mve_amd.synth import CONFIGS, make_scene

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = CONFIGS[name]
p = cfg["params"]
sc = make_scene(p, gpu=True)
ctx = api.Context(0); ctx.load_scene(sc)
st = api.Settings(scale=cfg["scale"], nrReconNeighbors=cfg["local_neighbors"])
out = {"config": name, "lib": os.environ.get("MI_DMRECON_LIB", "product")}
for label, views in (("one_call", list(range(p.n_views))), ("batch_x%d" % copies, list(range(p.n_views)) * copies)):
    res = ctx.reconstruct(st, views, want_normal=True)
    h = hashlib.sha256()
    for r in res[:p.n_views]:
        for k in ("depth", "conf", "dz", "normal", "views"):
            if k in r and r[k] is not None:
                h.update(np.ascontiguousarray(r[k]).tobytes())
    s = ctx.last_stats
    out[label] = {"sha256": h.hexdigest(), "n_patch": int(s.get("n_patch", 0)), "n_pass": int(s.get("n_pass", 0)),
                  "window_passes": int(s.get("n_patch_turns", 0)), "gather_passes": int(s.get("n_wave_turns", 0)),
                  "ms_bulk_kernel": s.get("ms_bulk_kernel"), "filled": int(sum(int((r["depth"] > 0).sum()) for r in res[:p.n_views]))}
print(json.dumps(out))
