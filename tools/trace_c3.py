import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mve_amd import api
from mve_amd.synth import CONFIGS, make_scene
cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
nref = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["params"].n_views
sc = make_scene(cfg["params"])
ctx = api.Context(0); ctx.load_scene(sc)
st = api.Settings(scale=cfg["scale"], nrReconNeighbors=cfg["local_neighbors"])
ctx.reconstruct(st, list(range(nref)), want_normal=False)
os.environ["MI_DMRECON_TRACE"] = "1"
t = time.time(); ctx.reconstruct(st, list(range(nref)), want_normal=False); print("wall", time.time() - t, ctx.last_stats)
