"""One library call of N x the C3 scene's 20 reference views, repeated: where a large batch's time goes and how it varies."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mve_amd import api
from mve_amd.synth import CONFIGS, make_scene
cfg = CONFIGS["C3"]; sc = make_scene(cfg["params"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = api.Context(0); ctx.load_scene(sc)
st = api.Settings(scale=cfg["scale"], nrReconNeighbors=cfg["local_neighbors"])
refs = list(range(20)) * n
out = ctx.alloc_outputs(st, refs, want_normal=False, pinned=True)
for i in range(10):
    t0 = time.perf_counter(); ctx.reconstruct(st, refs, want_normal=False, out=out); t = time.perf_counter() - t0
    s = ctx.last_stats
    print("call %d: %.1f ms (%.0f maps/s) plan gvs %.1f seeds %.1f | bulk kernels %.1f front %.1f sweeps %.1f | rounds %d ffr %d lat rounds %d gvs_dev %d" % (
        i, 1e3 * t, len(refs) / t, s["ms_plan_gvs"], s["ms_plan_seeds"], s["ms_bulk_kernel"], s["ms_front_kernel"], s["ms_sweep_kernels"],
        s["n_rounds"], s["front_first_round"], s["n_latency_rounds"], s["gvs_on_device"]))
