"""The FIRST reconstruct call of a process (what every view of the drop-in app's first batch pays): phases of a cold
one-view call, then of the same call again.  usage: python tools/cold_call.py [config] [views]"""
import os, sys, time
os.environ["MI_DMRECON_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mve_amd import api
from mve_amd.synth import CONFIGS, make_scene
cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
nref = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sc = make_scene(cfg["params"])
t = time.time(); ctx = api.Context(0); print("context %.1f ms" % (1e3 * (time.time() - t)), file=sys.stderr)
t = time.time(); ctx.load_scene(sc); print("scene staged %.1f ms" % (1e3 * (time.time() - t)), file=sys.stderr)
st = api.Settings(scale=cfg["scale"], nrReconNeighbors=cfg["local_neighbors"])
for rep in ("cold", "again", "fork cold", "fork again"):
    c = ctx.fork() if rep == "fork cold" else (c if rep == "fork again" else ctx)
    t = time.time(); c.reconstruct(st, list(range(nref)), want_normal=False)
    print("== %s call: %.1f ms" % (rep, 1e3 * (time.time() - t)), file=sys.stderr)
