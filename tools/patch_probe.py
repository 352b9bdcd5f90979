"""Where does a patch optimisation of the front kernel spend its time?  (MI_PROBE build: make -C mve_amd/csrc probe)
Prints, over the recorded attempts of one C3 call: microseconds per attempt (100 MHz clock), shader clock per attempt,
and the mean cycles between consecutive stamps grouped by (from id -> to id)."""
import ctypes, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mve_amd import api
api.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "libmi_dmrecon_probe.so")
from mve_amd.synth import CONFIGS, make_scene
cfg = CONFIGS["C3"]
sc = make_scene(cfg["params"])
ctx = api.Context(0); ctx.load_scene(sc)
L = api.load_library()
LOG = 96                                   # MI_PROBE_LOG of dmrecon_device.hip
NW = 8 + 400 * (LOG + 4)
L.mi_dmrecon_debug_buffer.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.mi_dmrecon_debug_buffer(None, NW)
st = api.Settings(scale=cfg["scale"], nrReconNeighbors=cfg["local_neighbors"])
buf = np.zeros(NW, np.uint64)
ctx.reconstruct(st, list(range(20)), want_normal=False)
L.mi_dmrecon_debug_buffer(ctypes.c_void_p(buf.ctypes.data), NW)       # drop the warm-up call's records
ctx.reconstruct(st, list(range(20)), want_normal=False)
print({k: ctx.last_stats[k] for k in ("ms_total", "ms_bulk_kernel", "ms_tail_kernel", "ms_front_kernel", "n_front_rounds_max", "n_front_attempts")})
L.mi_dmrecon_debug_buffer(ctypes.c_void_p(buf.ctypes.data), NW)
n = min(int(buf[0]), 400)
print("records", int(buf[0]), "kept", n)
us, cyc, seg, iters_h = [], [], collections.defaultdict(list), collections.Counter()
for r in range(n):
    o = buf[8 + r * (LOG + 4): 8 + (r + 1) * (LOG + 4)]
    ticks, np_, meta = int(o[1]), int(o[2]), int(o[3])
    ids = [int(v >> np.uint64(56)) for v in o[4:4 + np_]]
    ts = [int(v & np.uint64(0x00FFFFFFFFFFFFFF)) for v in o[4:4 + np_]]
    if np_ < 2:
        continue
    us.append(ticks / 100.0); cyc.append(ts[-1] - ts[0]); iters_h[(meta & 0xFFFF, meta >> 32)] += 1
    for k in range(1, np_):
        seg[(ids[k - 1], ids[k])].append(ts[k] - ts[k - 1])
    if r < 6:
        print("rec", r, "view", int(o[0]) & 0xFFFF, "round", (int(o[0]) >> 16) & 0xFFFF, "entries", (int(o[0]) >> 32) & 0xFFFF, "attempts in pass", int(o[0]) >> 48,
              "us", ticks / 100.0, "cycles", ts[-1] - ts[0], " ".join("%d:+%d" % (ids[k], ts[k] - ts[k - 1]) for k in range(1, np_)))
us, cyc = np.array(us), np.array(cyc)
print("attempts %d: us median %.2f mean %.2f p90 %.2f | cycles median %d | clock %.2f GHz" % (len(us), np.median(us), us.mean(), np.percentile(us, 90), np.median(cyc), np.median(cyc / us) / 1000.0))
print("(iterations, succeeded) histogram:", dict(iters_h))
tot = sum(sum(v) for v in seg.values())
for k, v in sorted(seg.items(), key=lambda kv: -sum(kv[1])):
    print("  %3d -> %3d : n %5d  mean %7.0f cycles  share %5.1f %%" % (k[0], k[1], len(v), np.mean(v), 100.0 * sum(v) / tot))
