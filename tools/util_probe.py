import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mve_amd import api
api.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "libmi_util.so")
from mve_amd.synth import CONFIGS, make_scene
cfg = CONFIGS["C3"]
sc = make_scene(cfg["params"])
ctx = api.Context(0); ctx.load_scene(sc)
st = api.Settings(scale=cfg["scale"])
ctx.reconstruct(st, list(range(20)), want_normal=False)
s = ctx.last_stats
tsum = s["n_seeds_ok"] - 32503 if s["n_seeds_ok"] > 1e6 else s["n_seeds_ok"]
hi = s["n_filled"] >> 32
print("sum of quad turns", s["n_seeds_ok"], "16*max per wave summed", hi, "utilisation", s["n_seeds_ok"] / max(hi, 1))
