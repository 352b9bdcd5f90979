"""Generates tests/golden/pset_*.npz from the REAL reference functions behind apps/scene2pset
(oracle/_ref/ref_pset_driver = oracle/ref_pset_driver.cc linked with the unmodified libs/mve objects).
Run in the authoring container only, after tests/golden/make_golden.py:

    python tests/golden/make_golden_pset.py

Cases: the reference's own depth maps of fixture G1 (view 0, scale 0, RGB) and G1b (view 2, scale 1, RGB level
image), and a 96x64 stress map (random holes, steps and spikes: ragged borders, depth discontinuities, COMPLEX
vertices; no colour image).  Each fixture holds the input depth map and the reference's vertices
(pixel, pos, normal, color, scale, conf) in the reference's vertex order.
"""
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
from mve_amd.scene_io import view_dir, write_mvei, write_png, write_scene  # noqa: E402
from oracle.pset_oracle import read_ref_dump  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_pset_driver")


def run_case(scene, view, depth, color, name, work):
    sdir = os.path.join(work, name)
    write_scene(sdir, scene)
    vd = view_dir(sdir, view)
    write_mvei(os.path.join(vd, "dm.mvei"), depth.astype(np.float32)[:, :, None])
    img = "none"
    if color is not None:
        write_png(os.path.join(vd, "col.png"), color)
        img = "col"
    dump = os.path.join(work, name + ".bin")
    subprocess.run([DRIVER, sdir, str(view), "dm", img, "2.5", dump], check=True)
    ref = read_ref_dump(dump)
    out = dict(depth=depth.astype(np.float32), view=np.int32(view), **{"ref_" + k: v for k, v in ref.items()})
    if color is not None:
        out["color"] = color
    np.savez_compressed(os.path.join(OUT, "pset_%s.npz" % name), **out)
    print(name, depth.shape, "vertices", len(ref["pixel"]))


def main():
    work = tempfile.mkdtemp(prefix="golden_pset_")
    g1 = dict(np.load(os.path.join(OUT, "g1_5views_160x120.npz")))
    sc1 = scene_from_golden(g1)
    run_case(sc1, 0, g1["s0v0_depth"], sc1.images[0], "g1_v0_s0", work)
    g1b = dict(np.load(os.path.join(OUT, "g1b_5views_322x241_scale1.npz")))
    sc1b = scene_from_golden(g1b)
    run_case(sc1b, 2, g1b["s1v2_depth"], g1b["s1v2_undist"], "g1b_v2_s1", work)
    # stress map on G1's camera 1
    rng = np.random.RandomState(11)
    h, w = 64, 96
    ys, xs = np.mgrid[0:h, 0:w]
    d = 8.0 + 0.02 * xs + 0.5 * np.sin(ys / 5.0) + rng.normal(0, 0.002, (h, w))
    d[20:40, 30:60] += 1.5                                  # a step: discontinuities on its rim
    spikes = rng.rand(h, w) < 0.02
    d[spikes] *= 1.2                                        # isolated spikes
    holes = rng.rand(h, w) < 0.15                           # holes -> ragged borders, COMPLEX vertices
    holes[4:36, 62:94] = False                              # ... and one intact region with interior vertices
    d[4:36, 62:94] = (8.0 + 0.02 * xs + 0.5 * np.sin(ys / 5.0))[4:36, 62:94]
    d[holes] = 0
    d[:, :3] = 0; d[50:, 70:] = 0
    run_case(sc1, 1, d.astype(np.float32), None, "stress_96x64", work)
    scene2pset_case(sc1, g1, work)
    shutil.rmtree(work)


def scene2pset_case(sc1, g1, work):
    """The unmodified apps/scene2pset (oracle/_ref/scene2pset_ref) with -F0 on scene G1 holding two depth maps: the
    reference's own map of view 0 and a synthetic one (holes, a step) on view 3.  One OpenMP thread: the reference
    appends views in completion order.  Fixture: the parsed vertices + the sha256 of the PLY file."""
    import hashlib
    from mve_amd.scene2pset import read_ply_points
    sdir = os.path.join(work, "s2p")
    write_scene(sdir, sc1)
    h, w = g1["s0v0_depth"].shape
    rng = np.random.RandomState(5)
    ys, xs = np.mgrid[0:h, 0:w]
    d3 = (9.5 + 0.004 * xs + 0.3 * np.cos(ys / 9.0)).astype(np.float32)
    d3[40:80, 50:110] += 0.8
    d3[rng.rand(h, w) < 0.08] = 0
    d3[:, :4] = 0
    write_mvei(os.path.join(view_dir(sdir, 0), "depth-L0.mvei"), g1["s0v0_depth"].astype(np.float32)[:, :, None])
    write_mvei(os.path.join(view_dir(sdir, 3), "depth-L0.mvei"), d3[:, :, None])
    ply = os.path.join(work, "s2p.ply")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "scene2pset_ref"), "-F0", sdir, ply], check=True, env=env,
                   stdout=subprocess.DEVNULL)
    ref = read_ply_points(ply)
    sha = hashlib.sha256(open(ply, "rb").read()).hexdigest()
    n0 = int(np.load(os.path.join(OUT, "pset_g1_v0_s0.npz"))["ref_pixel"].shape[0])
    np.savez_compressed(os.path.join(OUT, "scene2pset_g1_F0.npz"), depth_v3=d3, n_view0=np.int32(n0), sha256=np.array(sha),
                        **{"ref_" + k: v for k, v in ref.items()})
    print("scene2pset -F0: %d points (view 0: %d), sha256 %s" % (len(ref["pos"]), n0, sha[:16]))


if __name__ == "__main__":
    main()
