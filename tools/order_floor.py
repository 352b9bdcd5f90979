#!/usr/bin/env python3
"""How far apart two valid orders of the REFERENCE ALGORITHM are: the CPU restatement (oracle/, bit-identical to the compiled
reference in its own order) on full-size views of a BASELINE configuration, once in the reference's pop order and once per
alternative order -- ORC_QUEUE_ORDER = reverse (worst seed first), random:<seed> (uniformly random), jitter:<seed> (the
reference's order with the confidences perturbed by +-0.02 for the ORDER only: other tie-breaks, the orders closest to the
reference's own).  Every alternative against the reference order: fill IoU, relative depth median / p99, confidence median /
p99 -- the floor the map-level bounds of the GPU sweep (another re-ordering of the same algorithm) are set against.

  python tools/order_floor.py C3 [views, default all] -> tests/golden/order_floor_<config>.json   (CPU only; a fixture:
  tests/test_gpu_fullsize.py and bench.py read their fill-mask bound from it)

Runs here (no GPU needed), one process per view."""
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

ORDERS = ["reverse", "random:1", "random:2", "random:3", "jitter:1", "jitter:2", "jitter:3"]


def one_view(args):
    name, v = args
    from conftest import map_parity
    from mve_amd.synth import CONFIGS
    from oracle import oracle as orc
    cfg = CONFIGS[name]
    scene = _SCENE
    S = orc.OracleScene(scene)
    st = orc.make_settings(ref_view=v, scale=cfg["scale"], local_neighbors=cfg["local_neighbors"])
    os.environ.pop("ORC_QUEUE_ORDER", None)
    t0 = time.time()
    base = S.reconstruct(st)
    out = {"seconds_reference_order": time.time() - t0, "filled_reference_order": int((base["conf"] > 0).sum()), "orders": {}}
    for o in ORDERS:
        os.environ["ORC_QUEUE_ORDER"] = o                      # (read by the restatement at every reconstruction)
        r = S.reconstruct(st)
        out["orders"][o] = {k: float(x) for k, x in map_parity(r["depth"], r["conf"], base["depth"], base["conf"]).items()}
    os.environ.pop("ORC_QUEUE_ORDER", None)
    print("view %d: %s" % (v, {o: round(m["iou"], 4) for o, m in out["orders"].items()}), flush=True)
    return v, out


def main():
    global _SCENE
    name = sys.argv[1] if len(sys.argv) > 1 else "C3"
    from mve_amd.synth import CONFIGS, make_scene
    cfg = CONFIGS[name]
    views = [int(a) for a in sys.argv[2:]] or list(range(cfg["params"].n_views))
    _SCENE = make_scene(cfg["params"])                          # (inherited by the forked workers)
    with mp.get_context("fork").Pool(min(len(views), os.cpu_count() or 1)) as pool:
        res = dict(pool.map(one_view, [(name, v) for v in views]))
    worst = {}
    for v, r in res.items():
        for k, lo in (("iou", True), ("rel_med", False), ("rel_p99", False), ("conf_med", False), ("conf_p99", False)):
            vals = [m[k] for m in r["orders"].values()]
            worst.setdefault(k, {})[str(v)] = min(vals) if lo else max(vals)
    out = {"config": name, "what": "the reference algorithm (oracle restatement) against ITSELF: every alternative pop order of the "
                                   "queue against the reference's own order, per view; `worst` = per view the minimum fill IoU / "
                                   "the maximum of the other metrics over the alternative orders",
           "orders": ORDERS, "views": {str(v): res[v] for v in sorted(res)}, "worst": worst,
           "min_fill_iou": min(worst["iou"].values()), "max_rel_depth_p99": max(worst["rel_p99"].values()),
           "max_conf_abs_p99": max(worst["conf_p99"].values())}
    dst = os.path.join(ROOT, "tests", "golden", "order_floor_%s.json" % name.lower())
    json.dump(out, open(dst, "w"), indent=1)
    print("wrote", dst, "min IoU %.4f, max rel depth p99 %.2e, max conf p99 %.2e" % (out["min_fill_iou"], out["max_rel_depth_p99"], out["max_conf_abs_p99"]))


if __name__ == "__main__":
    main()
