"""Per-phase shader-clock stamps of one patch (MI_TIMING build of the library)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mve_amd import api
api.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "libmi_dmrecon_timing.so")
from mve_amd.synth import SynthParams, make_scene
sc = make_scene(SynthParams(n_views=8, width=640, height=360, n_features=800))
ctx = api.Context(0); ctx.load_scene(sc)
L = api.load_library()
st = api.Settings(refViewNr=0)
r = ctx.reconstruct(st, [0], want_views=True)[0]
ys, xs = np.nonzero(r["conf"] > 0.9)
sel = np.random.RandomState(0).permutation(len(xs))[:64]
xy = np.stack([xs[sel], ys[sel]], 1)
hyp = np.stack([r["depth"][ys[sel], xs[sel]] * 1.002, r["dz"][ys[sel], xs[sel], 0], r["dz"][ys[sel], xs[sel], 1]], 1)
loc = r["views"][ys[sel], xs[sel]]
L.mi_dmrecon_debug_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.mi_dmrecon_debug_timing(None, 0)
buf = np.zeros(500, np.uint64)
for lpv, win in ((16, 0), (16, 1), (1, 0), (1, 2)):
    os.environ["MI_DMRECON_HOOK_LPV"] = str(lpv)
    os.environ["MI_DMRECON_WIN"] = str(win)
    for rep in range(3):
        L.mi_dmrecon_debug_timing(ctypes.c_void_p(buf.ctypes.data), 500)
        out, _ = ctx.patch_optimize(st, 0, xy[:1] if lpv == 16 else xy[:16], hyp[:1] if lpv == 16 else hyp[:16], loc[:1] if lpv == 16 else loc[:16])
    L.mi_dmrecon_debug_timing(ctypes.c_void_p(buf.ctypes.data), 500)
    ids, ts = buf[0::2], buf[1::2]
    n = int((ts > 0).sum())
    print("lpv", lpv, "win", win, "iters", out[0, 7], "stamps", n)
    prev = None
    line = []
    for i in range(n):
        d = int(ts[i] - ts[0]); dd = d - prev if prev is not None else 0; prev = d
        line.append("%d:+%d" % (ids[i], dd))
    print(" ".join(line))
    print("total cycles", int(ts[n - 1] - ts[0]))
    # cycles spent in the interval that ENDS at each stamp id, summed over the patch
    agg = {}
    for i in range(1, n):
        agg[int(ids[i])] = agg.get(int(ids[i]), 0) + int(ts[i] - ts[i - 1])
    print("by closing stamp:", " ".join("%d=%d" % kv for kv in sorted(agg.items())))
