"""Lane divergence of the throughput layout on C3 (MI_HIST build of the library: build/libmi_dmrecon_hist.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mve_amd import api
api.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "libmi_dmrecon_hist.so")
from mve_amd.synth import CONFIGS, make_scene
cfg = CONFIGS["C3"]
sc = make_scene(cfg["params"])
ctx = api.Context(0); ctx.load_scene(sc)
L = api.load_library()
L.mi_dmrecon_debug_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.mi_dmrecon_debug_timing(None, 0)
st = api.Settings(scale=cfg["scale"], nrReconNeighbors=cfg["local_neighbors"])
buf = np.zeros(500, np.uint64)
ctx.reconstruct(st, list(range(20)), want_normal=False)
L.mi_dmrecon_debug_timing(ctypes.c_void_p(buf.ctypes.data), 500)
ctx.reconstruct(st, list(range(20)), want_normal=False)
L.mi_dmrecon_debug_timing(ctypes.c_void_p(buf.ctypes.data), 500)
h = buf[:32].astype(np.int64)
print("turns histogram (patches):", " ".join("%d:%d" % (i, v) for i, v in enumerate(h) if v))
pt, wt, nw = int(buf[32]), int(buf[33]), int(buf[34])
print("patches %d  sum patch turns %d  wave-optimisations %d  sum wave max turns %d" % (h.sum(), pt, nw, wt))
print("mean turns per patch %.2f, mean max per wave %.2f, lane activity %.1f %%" % (pt / max(h.sum(), 1), wt / max(nw, 1), 100.0 * pt / max(16 * wt, 1)))
wt2, vs_w, vs_q, act_q = int(buf[40]), int(buf[41]), int(buf[42]), int(buf[43])
print("wave-turns %d; with a view selection %d (%.1f %%); quads running VS %d; active quads per wave-turn %.2f" % (wt2, vs_w, 100.0 * vs_w / max(wt2, 1), vs_q, act_q / max(wt2, 1)))
print("wave-turns by number of distinct pass variants:", {k: int(buf[44 + k]) for k in range(5)})
