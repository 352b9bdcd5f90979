"""Generates tests/golden/scene2pset_g1_opts.npz from the UNMODIFIED apps/scene2pset (oracle/_ref/scene2pset_ref): the
options mve_amd/scene2pset.py gained last -- --mask (silhouette clipping, scene2pset.cc:406-465), --correspondence
(:65-118, :374, :481) and the .npts / .bnpts / .off writers behind mve::geom::save_mesh (libs/mve/mesh_io_npts.cc:66-98,
mesh_io_off.cc).  Run in the authoring container only, after make_golden.py and make_golden_pset.py:

    python tests/golden/make_golden_pset_opts.py

Scene: fixture G1 with the two depth maps of scene2pset_g1_F0.npz (views 0 and 3) and a one-channel `mask` embedding on
every view (zero inside a disc and a border strip that differ per view).  One OpenMP thread."""
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import scene_from_golden  # noqa: E402
from mve_amd.scene2pset import read_ply_points  # noqa: E402
from mve_amd.scene_io import view_dir, write_mvei, write_png, write_scene  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
APP = os.path.join(ROOT, "oracle", "_ref", "scene2pset_ref")


def make_masks(n_views, h, w):
    ys, xs = np.mgrid[0:h, 0:w]
    masks = []
    for v in range(n_views):
        m = np.full((h, w), 255, np.uint8)
        cx, cy, r = 40 + 20 * v, 30 + 12 * v, 18 + 3 * v
        m[(xs - cx) ** 2 + (ys - cy) ** 2 < r * r] = 0
        m[:, : 6 + 2 * v] = 0
        masks.append(m)
    return np.stack(masks)


def main():
    work = tempfile.mkdtemp(prefix="golden_pset_opts_")
    g1 = dict(np.load(os.path.join(OUT, "g1_5views_160x120.npz")))
    fx = dict(np.load(os.path.join(OUT, "scene2pset_g1_F0.npz")))
    sc = scene_from_golden(g1)
    sdir = os.path.join(work, "scene")
    write_scene(sdir, sc)
    h, w = g1["s0v0_depth"].shape
    write_mvei(os.path.join(view_dir(sdir, 0), "depth-L0.mvei"), g1["s0v0_depth"].astype(np.float32)[:, :, None])
    write_mvei(os.path.join(view_dir(sdir, 3), "depth-L0.mvei"), fx["depth_v3"][:, :, None])
    masks = make_masks(len(sc.cameras), h, w)
    for v in range(len(sc.cameras)):
        write_png(os.path.join(view_dir(sdir, v), "mask.png"), masks[v])
    env = dict(os.environ, OMP_NUM_THREADS="1")
    run = lambda *a: subprocess.run([APP] + list(a), check=True, env=env, stdout=subprocess.DEVNULL)
    out = dict(masks=masks)
    # --mask: -F0 so that normals, scale and confidence are there too
    ply = os.path.join(work, "masked.ply")
    run("-F0", "-mmask", sdir, ply)
    ref = read_ply_points(ply)
    out.update({"masked_" + k: v for k, v in ref.items()})
    # --correspondence (no mask, no box)
    ply = os.path.join(work, "corr.ply")
    run("-n", "-C", sdir, ply)
    out["corr_pos"] = read_ply_points(ply)["pos"]
    out["corr_data"] = np.loadtxt(ply + "_correspondence-data.csv", delimiter=",", skiprows=1, dtype=np.int64).reshape(-1, 2)
    out["corr_meta"] = np.loadtxt(ply + "_correspondence-metadata.csv", delimiter=",", skiprows=1, dtype=np.int64).reshape(-1, 4)
    out["corr_data_header"] = np.array(open(ply + "_correspondence-data.csv").readline())
    out["corr_meta_header"] = np.array(open(ply + "_correspondence-metadata.csv").readline())
    # .npts / .bnpts / .off (the app forces normals on and scale / confidence off for the first two)
    for ext in ("npts", "bnpts", "off"):
        path = os.path.join(work, "pts." + ext)
        run("-n", sdir, path)
        raw = open(path, "rb").read()
        out[ext + "_sha256"] = np.array(hashlib.sha256(raw).hexdigest())
        out[ext + "_head"] = np.frombuffer(raw[:600], np.uint8)
        out[ext + "_size"] = np.int64(len(raw))
    ply = os.path.join(work, "plain.ply")
    run("-n", sdir, ply)                                   # the same points as the three files above, parsed
    plain = read_ply_points(ply)
    out["plain_pos"], out["plain_normal"] = plain["pos"], plain["normal"]
    np.savez_compressed(os.path.join(OUT, "scene2pset_g1_opts.npz"), **out)
    print("masked: %d of %d points; correspondence rows %d; npts %d B, bnpts %d B, off %d B"
          % (len(out["masked_pos"]), len(out["plain_pos"]), len(out["corr_data"]), out["npts_size"], out["bnpts_size"], out["off_size"]))
    shutil.rmtree(work)


if __name__ == "__main__":
    main()
