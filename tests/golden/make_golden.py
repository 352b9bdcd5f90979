"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref, built by oracle/Makefile
from /root/reference).  Run in the authoring container only:

    python tests/golden/make_golden.py

The fixtures pin the CPU restatement (oracle/dmrecon_oracle.cc) and, through it and directly,
the HIP path.  They hold: the scene (cameras, RGB8 images, features), the reference's own
depth / conf / dz maps (apps/dmrecon, -O2 -ffp-contract=off build), its pyramid level
("undist-L1"), and per-patch values dumped by oracle/ref_patch_driver.cc.
"""
import os
import re
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mve_amd.scene_io import read_mvei, read_png, write_scene  # noqa: E402
from mve_amd.synth import SynthParams, make_scene  # noqa: E402
from oracle import oracle as orc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def scene_arrays(sc):
    cams = np.stack([c.as_f32() for c in sc.cameras])
    imgs = np.stack(sc.images)
    fpos = np.asarray([f.pos for f in sc.features], np.float32)
    foff = np.zeros(len(sc.features) + 1, np.int32)
    foff[1:] = np.cumsum([len(f.view_ids) for f in sc.features])
    fref = np.asarray([v for f in sc.features for v in f.view_ids], np.int32)
    return dict(cams=cams, imgs=imgs, fpos=fpos, foff=foff, fref=fref)


def run_maps(scene_dir, scale, master, k):
    dst, _ = orc.run_reference_app(scene_dir, scale, local_neighbors=k, master=master, flavour="strict")
    vd = os.path.join(dst, "views", "view_%04d.mve" % master)
    out = dict(depth=read_mvei(os.path.join(vd, "depth-L%d.mvei" % scale))[:, :, 0],
               conf=read_mvei(os.path.join(vd, "conf-L%d.mvei" % scale))[:, :, 0],
               dz=read_mvei(os.path.join(vd, "dz-L%d.mvei" % scale)))
    if scale > 0:
        out["undist"] = read_png(os.path.join(vd, "undist-L%d.png" % scale))
    shutil.rmtree(os.path.dirname(dst))
    return out


def main():
    # --- G1: 5 views 160x120, reconstructed at scale 0 (view 0) and scale 1 (view 2)
    p = SynthParams(n_views=5, width=160, height=120, n_features=300)
    sc = make_scene(p)
    work = tempfile.mkdtemp(prefix="golden_")
    sdir = os.path.join(work, "g1")
    write_scene(sdir, sc)
    g = scene_arrays(sc)
    m0 = run_maps(sdir, 0, 0, 4)
    for k, v in m0.items():
        g["s0v0_" + k] = v
    # --- G1b: same cameras at 320x240 so that scale 1 has a 160x120 master level
    p2 = SynthParams(n_views=5, width=322, height=241, n_features=300)   # odd sizes: ppoint rescale path
    sc2 = make_scene(p2)
    sdir2 = os.path.join(work, "g1b")
    write_scene(sdir2, sc2)
    g2 = scene_arrays(sc2)
    m1 = run_maps(sdir2, 1, 2, 4)
    for k, v in m1.items():
        g2["s1v2_" + k] = v
    # --- patch-level vectors from the reference classes (scene G1, ref view 0, scale 0)
    rng = np.random.RandomState(5)
    n = 48
    xs, ys = rng.randint(2, 158, n), rng.randint(2, 118, n)
    xs[:4] = [0, 1, 159, 80]; ys[:4] = [60, 60, 60, 119]           # border patches must fail
    depth = 10.0 + rng.uniform(-0.5, 0.5, n)
    dzi, dzj = rng.uniform(-0.01, 0.01, n), rng.uniform(-0.01, 0.01, n)
    dzi[:24] = 0; dzj[:24] = 0
    seeds = [[int(xs[i]), int(ys[i]), float(np.float32(depth[i])), float(np.float32(dzi[i])), float(np.float32(dzj[i]))]
             for i in range(n)]
    # half of them with a propagated local view set
    for i in range(n // 2, n):
        seeds[i] += [1, 2, 3, 4]
    lines = orc.run_reference_patch_driver(sdir, 0, 0, 4, "opt", seeds)
    gvs = [int(v) for v in lines[0][1:]]
    opt = np.zeros((n, 8), np.float32)
    opt_local = np.full((n, 4), -1, np.int32)
    for ln in lines[1:]:
        assert ln[0] == "P"
        i = int(ln[1])
        vals = [np.float32(v) for v in ln[2:9]]
        opt[i, :7] = vals
        nl = int(ln[9])
        opt_local[i, :nl] = [int(v) for v in ln[10:10 + nl]]
    lines = orc.run_reference_patch_driver(sdir, 0, 0, 4, "eval", seeds)
    ev_master = np.zeros((n, 5), np.float32)
    ev_ncc = np.zeros((n, len(gvs)), np.float32)
    ev_ok = np.zeros((n, len(gvs)), np.int32)
    ev_col = np.zeros((n, len(gvs), 25, 3), np.float32)
    ev_der = np.zeros((n, len(gvs), 25, 3), np.float32)
    cur = -1
    for ln in lines[1:]:
        if ln[0] == "S":
            cur = int(ln[1])
            ev_master[cur, 0] = int(ln[2])
            if int(ln[2]):
                ev_master[cur, 1:5] = [np.float32(v) for v in ln[3:7]]
        else:
            gi = gvs.index(int(ln[1]))
            ev_ncc[cur, gi] = np.float32(ln[2])
            ev_ok[cur, gi] = int(ln[3])
            if int(ln[3]):
                vals = np.asarray([np.float32(v) for v in ln[4:]], np.float32)
                ev_col[cur, gi] = vals[:75].reshape(25, 3)
                ev_der[cur, gi] = vals[75:].reshape(25, 3)
    g.update(gvs=np.asarray(gvs, np.int32), seeds_xy=np.stack([xs, ys], 1).astype(np.int32),
             seeds_hyp=np.stack([depth, dzi, dzj], 1).astype(np.float32),
             seeds_local=np.asarray([[1, 2, 3, 4] if i >= n // 2 else [-1] * 4 for i in range(n)], np.int32),
             opt=opt, opt_local=opt_local, ev_master=ev_master, ev_ncc=ev_ncc, ev_ok=ev_ok, ev_col=ev_col, ev_der=ev_der)
    # --- the sRGB table literal of libs/dmrecon/mvs_tools.cc:30-93
    src = open("/root/reference/libs/dmrecon/mvs_tools.cc").read()
    blk = src[src.index("srgb2lin[256] = {"):]
    blk = blk[:blk.index("};")]
    lut = np.asarray([np.float32(x.rstrip("f")) for x in re.findall(r"[0-9][0-9.e-]*f", blk)], np.float32)
    assert lut.shape == (256,)
    g["srgb2lin"] = lut
    # --- C1-like: 2 views, local-neighbors 1 (BASELINE config 1, reduced size)
    p3 = SynthParams(n_views=2, width=160, height=120, n_features=300)
    sc3 = make_scene(p3)
    sdir3 = os.path.join(work, "g2")
    write_scene(sdir3, sc3)
    g3 = scene_arrays(sc3)
    m3 = run_maps(sdir3, 0, 0, 1)
    for k, v in m3.items():
        g3["s0v0_" + k] = v
    np.savez_compressed(os.path.join(OUT, "g1_5views_160x120.npz"), **g)
    np.savez_compressed(os.path.join(OUT, "g1b_5views_322x241_scale1.npz"), **g2)
    np.savez_compressed(os.path.join(OUT, "g2_2views_160x120_k1.npz"), **g3)
    shutil.rmtree(work)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")
    print("filled s0v0:", int((m0["depth"] > 0).sum()), "s1v2:", int((m1["depth"] > 0).sum()), "k1:", int((m3["depth"] > 0).sum()))


if __name__ == "__main__":
    main()
