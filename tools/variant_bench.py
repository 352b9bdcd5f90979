import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mve_amd import api
if len(sys.argv) > 1 and sys.argv[1] != "default":
    api.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", sys.argv[1])
from mve_amd.synth import CONFIGS, make_scene
cfg = CONFIGS["C3"]
sc = make_scene(cfg["params"])
ctx = api.Context(0); ctx.load_scene(sc)
st = api.Settings(scale=cfg["scale"])
refs = list(range(20))
out = ctx.alloc_outputs(st, refs, want_normal=False, pinned=True)
ctx.reconstruct(st, refs, want_normal=False, out=out)
ts = []
for i in range(4):
    t = time.time(); ctx.reconstruct(st, refs, want_normal=False, out=out); ts.append(time.time() - t)
print(sys.argv[1] if len(sys.argv) > 1 else "default", "ms/step %.1f" % (1000 * min(ts)), "opt kernel ms %.1f" % ctx.last_stats["ms_opt_kernel"])
