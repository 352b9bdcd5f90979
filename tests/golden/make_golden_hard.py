"""Generates tests/golden/h1_hard_9views_208x156.npz from the REAL reference (oracle/_ref, built by
oracle/Makefile from /root/reference).  Run in the authoring container only:

    python tests/golden/make_golden_hard.py

Scene H1 (mve_amd.synth.make_hard_scene): a background with a depth step, a foreground occluder plate, a
textureless band and one view with little overlap -- what the smooth height-field scenes never reach:
failed samplings of neighbour views, replaceViews / the iteration-14 rule (patch_optimization.cc:207-239),
local view sets that must be re-selected (local_view_selection.cc:149-160), unfilled regions.

Holds: the scene; the reference's depth / conf / dz maps of reference views 0 (central) and 8 (low overlap);
patch-level results of 260 hypotheses dumped from the reference's own PatchOptimization class by
oracle/ref_patch_driver.cc, half of them with a propagated local view set.
"""
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import run_maps, scene_arrays  # noqa: E402
from mve_amd.scene_io import write_scene  # noqa: E402
from mve_amd.synth import HardParams, hard_render, make_hard_scene  # noqa: E402
from oracle import oracle as orc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    hp = HardParams()
    sc = make_hard_scene(hp)
    work = tempfile.mkdtemp(prefix="golden_hard_")
    sdir = os.path.join(work, "h1")
    write_scene(sdir, sc)
    g = scene_arrays(sc)
    for v in (0, 8):
        m = run_maps(sdir, 0, v, 4)
        for k, val in m.items():
            g["s0v%d_%s" % (v, k)] = val
        print("view", v, "filled", int((m["depth"] > 0).sum()), "of", m["depth"].size)
    # patch hypotheses in reference view 0: near the true surface, many of them at the occluder's edge, the depth
    # step and the textureless band
    _, truth, sid = hard_render(hp, sc.cameras[0], hp.width, hp.height)
    rng = np.random.RandomState(17)
    n = 260
    edge = np.zeros_like(sid, bool)
    edge[1:-1, 1:-1] = (sid[1:-1, 1:-1] != sid[1:-1, :-2]) | (sid[1:-1, 1:-1] != sid[1:-1, 2:]) \
        | (sid[1:-1, 1:-1] != sid[:-2, 1:-1]) | (sid[1:-1, 1:-1] != sid[2:, 1:-1])
    ey, ex = np.nonzero(edge)
    xs, ys = rng.randint(2, hp.width - 2, n), rng.randint(2, hp.height - 2, n)
    for i in list(range(0, 60)) + list(range(n // 2, n // 2 + 90)):   # within 9 px of a discontinuity
        j = rng.randint(len(ex))
        xs[i] = np.clip(ex[j] + rng.randint(-9, 10), 2, hp.width - 3)
        ys[i] = np.clip(ey[j] + rng.randint(-9, 10), 2, hp.height - 3)
    depth = truth[ys, xs] * (1.0 + rng.uniform(-0.01, 0.01, n))
    dzi, dzj = rng.uniform(-0.01, 0.01, n), rng.uniform(-0.01, 0.01, n)
    seeds = [[int(xs[i]), int(ys[i]), float(np.float32(depth[i])), float(np.float32(dzi[i])), float(np.float32(dzj[i]))]
             for i in range(n)]
    lines = orc.run_reference_patch_driver(sdir, 0, 0, 4, "opt", seeds[:1])
    gvs = [int(v) for v in lines[0][1:]]
    local = np.full((n, 4), -1, np.int32)
    # which global view sees the true surface point of each seed pixel (not occluded, inside the image)?
    from mve_amd.synth import hard_trace, project
    t0, _, X, Y, Z, _ = hard_trace(hp, sc.cameras[0], hp.width, hp.height)
    vis = np.zeros((n, len(sc.cameras)), bool)
    dmaps = [hard_render(hp, c, hp.width, hp.height)[1] for c in sc.cameras]
    for v in gvs:
        P = np.stack([X[ys, xs], Y[ys, xs], Z[ys, xs]], 1)
        u, vv, zc = project(sc.cameras[v], hp.width, hp.height, P)
        inside = (zc > 0) & (u > 3) & (vv > 3) & (u < hp.width - 4) & (vv < hp.height - 4)
        ui, vi = np.clip(np.rint(u).astype(int), 0, hp.width - 1), np.clip(np.rint(vv).astype(int), 0, hp.height - 1)
        d = np.linalg.norm(P - sc.cameras[v].position(), axis=1)
        vis[:, v] = inside & (np.abs(dmaps[v][vi, ui] - d) < 0.05)
    n_mixed = 0
    for i in range(n // 2, n):
        # a propagated set of four global views; where possible one or two of them do NOT see the point (occluded by
        # the plate / across the step) while enough others do: the optimisation must replace them
        seen = [v for v in gvs if vis[i, v]]
        hid = [v for v in gvs if not vis[i, v]]
        if len(hid) >= 1 and len(seen) >= 5 and i % 4 != 3:
            k_hid = 1 + (i % 2 if len(hid) >= 2 else 0)
            pick = list(rng.choice(hid, k_hid, replace=False)) + list(rng.choice(seen, 4 - k_hid, replace=False))
            n_mixed += 1
        else:
            pick = list(rng.choice(gvs, 4, replace=False))
        local[i] = sorted(int(v) for v in pick)
        seeds[i] += [int(v) for v in local[i]]
    print("propagated sets with a view that cannot see the point:", n_mixed)
    lines = orc.run_reference_patch_driver(sdir, 0, 0, 4, "opt", seeds)
    opt = np.zeros((n, 8), np.float32)
    opt_local = np.full((n, 4), -1, np.int32)
    for ln in lines[1:]:
        assert ln[0] == "P"
        i = int(ln[1])
        opt[i, :7] = [np.float32(v) for v in ln[2:9]]
        nl = int(ln[9])
        opt_local[i, :nl] = [int(v) for v in ln[10:10 + nl]]
    ok = opt[:, 0] > 0
    changed = ok[n // 2:] & (opt_local[n // 2:] != local[n // 2:]).any(1)
    print("patches: %d of %d succeed; %d of the %d propagated sets were changed by the optimisation"
          % (ok.sum(), n, changed.sum(), n - n // 2))
    g.update(gvs=np.asarray(gvs, np.int32), seeds_xy=np.stack([xs, ys], 1).astype(np.int32),
             seeds_hyp=np.stack([depth, dzi, dzj], 1).astype(np.float32), seeds_local=local, opt=opt, opt_local=opt_local)
    np.savez_compressed(os.path.join(OUT, "h1_hard_9views_208x156.npz"), **g)
    shutil.rmtree(work)
    print("h1_hard_9views_208x156.npz", os.path.getsize(os.path.join(OUT, "h1_hard_9views_208x156.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
