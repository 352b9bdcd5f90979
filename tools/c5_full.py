"""BASELINE config 5 at its stated size, once: 100 views of 4032 x 3024, reconstructed at scale 3 (504 x 378 maps).

Measures, on one MI355X: scene residency in HBM, the upload (blocking from pageable memory vs the asynchronous
path through two page-locked buffers), depth-maps/s of the whole scene (one call of 100 reference views, and two
host threads x 50), and compares one view with the CPU oracle.  Prints one JSON line (kept under profiles/).

    python tools/c5_full.py [n_views]          (GPU box; ~3-4 minutes, most of it rendering the synthetic images)
"""
import json, os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mve_amd import api
from mve_amd.synth import CONFIGS, SynthParams, make_scene, true_depth

cfg = CONFIGS["C5"]
p = cfg["params"]
if len(sys.argv) > 1:
    p = SynthParams(**{**p.__dict__, "n_views": int(sys.argv[1])})
t0 = time.time(); scene = make_scene(p); t_render = time.time() - t0
import subprocess
def hbm_used():
    try:
        out = subprocess.run(["rocm-smi", "--showmeminfo", "vram", "--json"], capture_output=True, text=True, timeout=30).stdout
        j = json.loads(out); k = sorted(j)[0]
        return int(j[k]["VRAM Total Used Memory (B)"])
    except Exception:
        return None
m0 = hbm_used()
a = api.Context(0)
t0 = time.time(); a.load_scene(scene, pinned_staging=True); t_async = time.time() - t0
m1 = hbm_used()
st = api.Settings(scale=cfg["scale"], nrReconNeighbors=cfg["local_neighbors"])
refs = list(range(p.n_views))
out = a.alloc_outputs(st, refs, want_normal=False, pinned=True)
a.reconstruct(st, refs, want_normal=False, out=out)                       # warm-up (builds the scene tables too)
t0 = time.time(); res = a.reconstruct(st, refs, want_normal=False, out=out); t_one = time.time() - t0
stats = dict(a.last_stats)
os.environ["MI_DMRECON_GVS_DEVICE"] = "0"                                  # the same call with the view selection on the host
a.reconstruct(st, refs, want_normal=False, out=out)
t0 = time.time(); a.reconstruct(st, refs, want_normal=False, out=out); t_one_host = time.time() - t0
stats_host = dict(a.last_stats)
del os.environ["MI_DMRECON_GVS_DEVICE"]
fill = float(np.mean([(r["conf"] > 0).mean() for r in res]))
# two host threads, half the views each
f = a.fork()
halves = [refs[0::2], refs[1::2]]
outs = [c.alloc_outputs(st, h, want_normal=False, pinned=True) for c, h in zip((a, f), halves)]
for c, h, o in zip((a, f), halves, outs):
    c.reconstruct(st, h, want_normal=False, out=o)
ths = [threading.Thread(target=lambda c=c, h=h, o=o: c.reconstruct(st, h, want_normal=False, out=o)) for c, h, o in zip((a, f), halves, outs)]
t0 = time.time(); [t.start() for t in ths]; [t.join() for t in ths]; t_two = time.time() - t0
# accuracy against the analytic truth of three views; parity of one view against the CPU oracle
errs = []
for v in (0, p.n_views // 2, p.n_views - 1):
    w, h = a.level_size(v, cfg["scale"])
    gt = true_depth(p, scene.cameras[v], w, h)
    m = res[v]["conf"] > 0
    errs.append(float(np.median(np.abs(res[v]["depth"][m] - gt[m]))))
par = None
try:
    from oracle import oracle as orc
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from conftest import map_parity
    S = orc.OracleScene(scene)
    t0 = time.time(); o = S.reconstruct(orc.make_settings(ref_view=2, scale=cfg["scale"], local_neighbors=cfg["local_neighbors"])); t_orc = time.time() - t0
    par = map_parity(res[2]["depth"], res[2]["conf"], o["depth"], o["conf"]); par["oracle_seconds_one_view"] = t_orc
except Exception as e:
    par = {"error": repr(e)}
print(json.dumps({
    "workload": "C5: %d views %dx%d, scale %d (%dx%d maps), 2000 features" % (p.n_views, p.width, p.height, cfg["scale"], res[0]["depth"].shape[1], res[0]["depth"].shape[0]),
    "render_seconds": t_render, "upload_async_pinned_seconds": t_async,
    "host_image_bytes": int(sum(im.nbytes for im in scene.images)),
    "hbm_used_before_after_upload": [m0, m1], "hbm_scene_bytes": (m1 - m0) if (m0 is not None and m1 is not None) else None,
    "one_call_all_views": {"seconds": t_one, "depth_maps_per_s": p.n_views / t_one, "n_rounds": stats["n_rounds"],
                           "ms_bulk_kernel": stats["ms_bulk_kernel"], "ms_tail_kernel": stats["ms_tail_kernel"],
                           "n_patch": stats["n_patch"], "n_eval": stats["n_eval"], "n_filled": stats["n_filled"],
                           "gvs_on_device": stats["gvs_on_device"], "ms_plan_gvs": stats["ms_plan_gvs"], "ms_plan_seeds": stats["ms_plan_seeds"]},
    "one_call_view_selection_on_host": {"seconds": t_one_host, "depth_maps_per_s": p.n_views / t_one_host,
                                        "ms_plan_gvs": stats_host["ms_plan_gvs"], "ms_plan_seeds": stats_host["ms_plan_seeds"]},
    "two_threads_half_each": {"seconds": t_two, "depth_maps_per_s": p.n_views / t_two},
    "mean_fill": fill, "median_abs_depth_error_views_first_mid_last": errs, "parity_view2_vs_oracle": par}))
