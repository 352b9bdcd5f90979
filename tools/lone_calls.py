"""Lone calls of N reference views of the C3 scene (a rank's share of the scene in strong scaling, BASELINE config 4, when the
rank has its GPU to itself): median / min wall time of a library call and where it goes, for N = 1, 2, 3, 5, 7, 10, 20.
Prints one JSON object (kept as profiles/r<N>_lone_calls.json: bench.py's `strong_scaling.predicted` reads it).

    python tools/lone_calls.py [CONFIG] [calls per N]         (GPU box)
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mve_amd import api
from mve_amd.dist import shard_views
from mve_amd.synth import CONFIGS, make_scene

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cfg = CONFIGS[name]
p = cfg["params"]
sc = make_scene(p)
ctx = api.Context(0); ctx.load_scene(sc)
st = api.Settings(scale=cfg["scale"], nrReconNeighbors=cfg["local_neighbors"])
allv = list(range(p.n_views))
out = {"config": name, "calls_per_size": reps, "sizes": {}}
for world in (20, 10, 8, 4, 3, 2, 1):
    views = shard_views(allv, 0, world)                      # rank 0's share with `world` ranks
    n = len(views)
    if str(n) in out["sizes"]:
        continue
    o = ctx.alloc_outputs(st, views, want_normal=False, pinned=True)
    for _ in range(2):
        ctx.reconstruct(st, views, want_normal=False, out=o)
    ts, acc = [], {}
    for _ in range(reps):
        t0 = time.perf_counter(); ctx.reconstruct(st, views, want_normal=False, out=o); ts.append(time.perf_counter() - t0)
        for k, v in ctx.last_stats.items():
            acc[k] = acc.get(k, 0) + v
    out["sizes"][str(n)] = {"views": n, "ranks_this_is_the_share_of": world, "ms_median": 1000 * float(np.median(ts)), "ms_min": 1000 * float(np.min(ts)),
                            "ms_bulk_kernel": acc["ms_bulk_kernel"] / reps, "ms_tail_kernel": acc["ms_tail_kernel"] / reps,
                            "ms_front_kernel": acc["ms_front_kernel"] / reps, "ms_plan": (acc["ms_plan_gvs"] + acc["ms_plan_seeds"]) / reps,
                            "front_team": acc["front_team"] / reps, "rounds": acc["n_rounds"] / reps,
                            "front_first_round": acc["front_first_round"] / reps, "n_latency_rounds": acc["n_latency_rounds"] / reps}
print(json.dumps(out))
