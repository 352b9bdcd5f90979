"""scene2pset on the GPU: the point-set export of apps/scene2pset/scene2pset.cc over the C ABI.

    python -m mve_amd.scene2pset [OPTS] SCENE_DIR MESH_OUT.ply

Per view: mi_dmrecon_pointset (depth map -> oriented points: scene2pset.cc:262-356); around it this module mirrors the
driver: option names and defaults (scene2pset.cc:159-235), view filters (:268-300), AABB filter (:394-415), the PLY
layout of mve::geom::save_ply_mesh (libs/mve/mesh_io_ply.cc:740-815: x y z [nx ny nz] [red green blue] [confidence]
[value], binary little endian).  Views are appended in view order, the points of a view in ascending pixel order
(the reference appends views in OpenMP completion order and numbers a view's points by first use: as a SET the
output is the reference's; the rows of the --correspondence table follow the same order).  --mask (silhouette
clipping, :406-465), --correspondence (:65-118) and the .npts / .bnpts / .off outputs of mve::geom::save_mesh
(mesh_io_npts.cc:66-98, mesh_io_off.cc) are pinned to the unmodified app by tests/golden/scene2pset_g1_opts.npz.
Depth maps are .mvei embeddings, colour and mask images .png embeddings of the view directories (scene_io.py).
"""
from __future__ import annotations

import argparse
import os
import sys
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import api, scene_io


def view_entries(scene_dir: str):
    """(view id, view directory, camera) of every view directory, in id order (mve::Scene::get_views)."""
    vroot = os.path.join(scene_dir, "views")
    out = []
    for e in sorted(x for x in os.listdir(vroot) if x.endswith(".mve")):
        ini = scene_io._parse_ini(os.path.join(vroot, e, "meta.ini"))
        cam = scene_io.Camera()
        if "camera.focal_length" in ini:
            cam.flen = float(np.float32(ini["camera.focal_length"]))
        if "camera.pixel_aspect" in ini:
            cam.paspect = float(np.float32(ini["camera.pixel_aspect"]))
        if "camera.principal_point" in ini:
            cam.ppoint = [float(np.float32(x)) for x in ini["camera.principal_point"].split()]
        if "camera.rotation" in ini:
            cam.rot = [float(np.float32(x)) for x in ini["camera.rotation"].split()]
        if "camera.translation" in ini:
            cam.trans = [float(np.float32(x)) for x in ini["camera.translation"].split()]
        out.append((int(ini.get("view.id", len(out))), os.path.join(vroot, e), cam))
    out.sort(key=lambda t: t[0])
    return out


def scene_to_pointset(scene_dir: str, dmname: str = "depth-L0", image: str = "undistorted", ids: Optional[Sequence[int]] = None,
                      aabb: Optional[Sequence[float]] = None, min_fraction: float = 0.0, scale_factor: float = 2.5,
                      poisson_normals: bool = False, ctx: Optional[api.Context] = None, verbose: bool = True) -> Dict[str, np.ndarray]:
    """All views' points: dict(pos, normal, color (uint8 or None), scale, conf, view) -- scene2pset.cc:255-420."""
    own = ctx is None
    if own:
        ctx = api.Context(0)
    parts: Dict[str, List[np.ndarray]] = dict(pos=[], normal=[], color=[], scale=[], conf=[], view=[], pixel=[])
    meta: List[List[int]] = []                                       # --correspondence: view id, width, height, first vertex
    have_color = True
    for vid, vdir, cam in view_entries(scene_dir):
        if ids and vid not in ids:                                   # :271-274
            continue
        if cam.flen == 0.0:                                          # :276-278
            continue
        dpath = os.path.join(vdir, dmname + ".mvei")
        if not os.path.exists(dpath):                                # :280-282
            continue
        depth = scene_io.read_mvei(dpath)[:, :, 0]
        if min_fraction > 0.0:                                       # :284-300
            fraction = float(np.float32(np.count_nonzero(depth > 0)) / np.float32(depth.size))
            if fraction < min_fraction:
                if verbose:
                    print("View %d: Fill status %.2f%%, skipping." % (vid, fraction * 100.0))
                continue
        color = None
        ipath = os.path.join(vdir, image + ".png")
        if image and os.path.exists(ipath):
            color = scene_io.read_png(ipath)
            if color.shape[:2] != depth.shape:
                raise ValueError("Color image dimension mismatch")   # depthmap.cc:331-332
        if verbose:
            print('Processing view %d%s...' % (vid, " (with colors)" if color is not None else ""))
        ps = ctx.pointset(cam, depth, color, scale_factor=scale_factor)
        n = len(ps["pixel"])
        keep = np.ones(n, bool)
        if aabb is not None:                                         # math::geom::point_box_overlap, :397-399
            lo, hi = np.asarray(aabb[:3], np.float32), np.asarray(aabb[3:], np.float32)
            keep = np.all((ps["pos"] >= lo) & (ps["pos"] <= hi), axis=1)
        nrm = ps["normal"]
        if poisson_normals:                                          # poisson_scale_normals, :119-127
            nrm = nrm * ps["conf"][:, None]
        parts["pos"].append(ps["pos"][keep]); parts["normal"].append(nrm[keep].astype(np.float32))
        parts["scale"].append(ps["scale"][keep]); parts["conf"].append(ps["conf"][keep])
        parts["view"].append(np.full(int(keep.sum()), vid, np.int32))
        meta.append([vid, depth.shape[1], depth.shape[0], sum(len(p) for p in parts["pixel"])])
        parts["pixel"].append(ps["pixel"][keep].astype(np.int64))
        if color is None:
            have_color = False
        else:
            px = ps["pixel"][keep]
            c = color.reshape(depth.size, -1)[px]
            parts["color"].append((c[:, :3] if c.shape[1] >= 3 else np.repeat(c[:, :1], 3, 1)).astype(np.uint8))
    if own:
        ctx.close()
    cat = lambda k, shape, dt: np.concatenate(parts[k]) if parts[k] else np.zeros(shape, dt)
    out = dict(pos=cat("pos", (0, 3), np.float32), normal=cat("normal", (0, 3), np.float32),
               scale=cat("scale", (0,), np.float32), conf=cat("conf", (0,), np.float32), view=cat("view", (0,), np.int32),
               pixel=cat("pixel", (0,), np.int64), meta=np.asarray(meta, np.int64).reshape(-1, 4))
    # the reference writes colours only if every contributing view had an image (save_ply_mesh: sizes must match)
    out["color"] = cat("color", (0, 3), np.uint8) if have_color and parts["color"] else None
    return out


def mask_filter(scene_dir: str, ps: Dict[str, np.ndarray], maskname: str, verbose: bool = True) -> Dict[str, np.ndarray]:
    """Silhouette clipping (scene2pset.cc:406-465): a point is dropped if it projects onto a zero pixel of ANY view's
    one-channel `maskname` image.  The projection is the reference's float arithmetic, operation by operation
    (Matrix4f::mult(v, 1), Matrix3f * v: math/matrix.h:475-492; CameraInfo::fill_calibration, camera.cc:125-144)."""
    from PIL import Image
    f32 = np.float32
    pos = ps["pos"].astype(f32)
    dead = np.zeros(len(pos), bool)
    for vid, vdir, cam in view_entries(scene_dir):
        if cam.flen == 0.0:
            continue
        mpath = os.path.join(vdir, maskname + ".png")
        if not os.path.exists(mpath):
            if verbose:
                print('Mask not found for image "view %d", skipping.' % vid)
            continue
        with Image.open(mpath) as im:
            if len(im.getbands()) != 1:
                if verbose:
                    print('Expected 1-channel mask for image "view %d", skipping.' % vid)
                continue
            mask = np.asarray(im)
        h, w = mask.shape
        rot, t = np.asarray(cam.rot, f32), np.asarray(cam.trans, f32)
        fw, fh, flen, pa = f32(w), f32(h), f32(cam.flen), f32(cam.paspect)
        if (fw / fh) * pa < f32(1.0):                                  # portrait
            ax, ay = flen * fh / pa, flen * fh
        else:
            ax, ay = flen * fw, flen * fw * pa
        cx, cy = fw * f32(cam.ppoint[0]), fh * f32(cam.ppoint[1])
        x, y, z = pos[:, 0], pos[:, 1], pos[:, 2]
        zero = f32(0.0)
        c = [((zero + rot[3 * r] * x) + rot[3 * r + 1] * y) + rot[3 * r + 2] * z + f32(1.0) * t[r] for r in range(3)]
        px = ((zero + ax * c[0]) + zero * c[1]) + cx * c[2]
        py = ((zero + zero * c[0]) + ay * c[1]) + cy * c[2]
        pz = ((zero + zero * c[0]) + zero * c[1]) + f32(1.0) * c[2]
        with np.errstate(divide="ignore", invalid="ignore"):
            u, v = px / pz, py / pz
        inside = ~((u < 0) | (v < 0) | (u >= fw) | (v >= fh)) & np.isfinite(u) & np.isfinite(v) & ~dead
        ix = np.where(inside, u, 0).astype(np.int64); iy = np.where(inside, v, 0).astype(np.int64)
        hit = inside & (mask[np.clip(iy, 0, h - 1), np.clip(ix, 0, w - 1)] == 0)
        dead |= hit
    if verbose:
        print("Filtered a total of %d points." % int(dead.sum()))
    keep = ~dead
    return {k: (v[keep] if isinstance(v, np.ndarray) and k != "meta" and len(v) == len(keep) else v) for k, v in ps.items()}


def write_correspondence(mesh_out: str, ps: Dict[str, np.ndarray]) -> None:
    """<out>_correspondence-data.csv (pixel x, y of every vertex) and -metadata.csv (scene2pset.cc:85-118)."""
    with open(mesh_out + "_correspondence-data.csv", "w") as f:
        f.write("x, y\n")
        off = 0
        for vid, w, h, first in ps["meta"]:
            nxt = [m[3] for m in ps["meta"] if m[3] > first]
            end = min(nxt) if nxt else len(ps["pixel"])
            px = ps["pixel"][first:end]
            f.write("".join("%d, %d\n" % (p % w, p // w) for p in px))
    with open(mesh_out + "_correspondence-metadata.csv", "w") as f:
        f.write("View_ID, Width, Height, First_Vertex_Index\n")
        f.write("".join("%d, %d, %d, %d\n" % tuple(m) for m in ps["meta"]))


def write_npts(path: str, ps: Dict[str, np.ndarray], binary: bool) -> None:
    """mve::geom::save_npts_mesh (mesh_io_npts.cc:66-98): position + normal per point, six floats raw or as text."""
    if len(ps["pos"]) == 0:
        raise ValueError("Input mesh is empty")
    rows = np.concatenate([ps["pos"].astype("<f4"), ps["normal"].astype("<f4")], axis=1)
    with open(path, "wb") as f:
        if binary:
            f.write(rows.tobytes())
        else:                                                        # operator<< of float: printf %g
            f.write("".join("%g %g %g %g %g %g\n" % tuple(float(x) for x in r) for r in rows).encode("ascii"))


def write_off(path: str, ps: Dict[str, np.ndarray]) -> None:
    """mve::geom::save_off_mesh for a mesh without faces (mesh_io_off.cc): fixed notation, seven digits."""
    with open(path, "wb") as f:
        f.write(("OFF\n%d 0 0\n" % len(ps["pos"])).encode("ascii"))
        f.write("".join("%.7f %.7f %.7f\n" % tuple(float(x) for x in r) for r in ps["pos"].astype(np.float32)).encode("ascii"))


def write_ply(path: str, ps: Dict[str, np.ndarray], with_normals: bool, with_scale: bool, with_conf: bool) -> None:
    """mve::geom::save_ply_mesh for a point set (mesh_io_ply.cc:740-815), binary little endian."""
    n = len(ps["pos"])
    fields = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")]
    hdr = ["ply", "format binary_little_endian 1.0", "comment Export generated by libmve", "element vertex %d" % n,
           "property float x", "property float y", "property float z"]
    if with_normals:
        fields += [("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4")]
        hdr += ["property float nx", "property float ny", "property float nz"]
    if ps.get("color") is not None:
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
        hdr += ["property uchar red", "property uchar green", "property uchar blue"]
    if with_conf:
        fields.append(("confidence", "<f4")); hdr.append("property float confidence")
    if with_scale:
        fields.append(("value", "<f4")); hdr.append("property float value")
    hdr.append("end_header")
    rec = np.zeros(n, dtype=np.dtype(fields))
    rec["x"], rec["y"], rec["z"] = ps["pos"][:, 0], ps["pos"][:, 1], ps["pos"][:, 2]
    if with_normals:
        rec["nx"], rec["ny"], rec["nz"] = ps["normal"][:, 0], ps["normal"][:, 1], ps["normal"][:, 2]
    if ps.get("color") is not None:
        rec["red"], rec["green"], rec["blue"] = ps["color"][:, 0], ps["color"][:, 1], ps["color"][:, 2]
    if with_conf:
        rec["confidence"] = ps["conf"]
    if with_scale:
        rec["value"] = ps["scale"]
    with open(path, "wb") as f:
        f.write(("\n".join(hdr) + "\n").encode("ascii"))
        f.write(rec.tobytes())


def read_ply_points(path: str) -> Dict[str, np.ndarray]:
    """Minimal reader for the vertex element of a binary little-endian PLY as written by libmve / write_ply."""
    raw = open(path, "rb").read()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    n, fields, in_vertex = 0, [], False
    tmap = {"float": "<f4", "uchar": "u1", "int": "<i4", "uint": "<u4", "double": "<f8"}
    for line in raw[:end].decode("ascii").splitlines():
        t = line.split()
        if t[:2] == ["element", "vertex"]:
            n, in_vertex = int(t[2]), True
        elif t[:1] == ["element"]:
            in_vertex = False
        elif t[:1] == ["property"] and in_vertex:
            fields.append((t[2], tmap[t[1]]))
    rec = np.frombuffer(raw, dtype=np.dtype(fields), count=n, offset=end)
    out = dict(pos=np.stack([rec["x"], rec["y"], rec["z"]], 1))
    names = rec.dtype.names
    if "nx" in names:
        out["normal"] = np.stack([rec["nx"], rec["ny"], rec["nz"]], 1)
    if "red" in names:
        out["color"] = np.stack([rec["red"], rec["green"], rec["blue"]], 1)
    if "confidence" in names:
        out["conf"] = rec["confidence"].copy()
    if "value" in names:
        out["scale"] = rec["value"].copy()
    return out


def main(argv: Optional[Sequence[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="scene2pset", description="Generates a pointset from the scene by projecting "
                                 "reconstructed depth values in the world coordinate system (GPU build).")
    ap.add_argument("scene_dir"); ap.add_argument("mesh_out")
    ap.add_argument("-d", "--depthmap", default="depth-L0", help="Name of depth map to use [depth-L0]")
    ap.add_argument("-i", "--image", default="undistorted", help="Name of color image to use [undistorted]")
    ap.add_argument("-n", "--with-normals", action="store_true"); ap.add_argument("-s", "--with-scale", action="store_true")
    ap.add_argument("-c", "--with-conf", action="store_true")
    ap.add_argument("-v", "--views", default="", help="View IDs to use, e.g. 0,2,5-7 [all]")
    ap.add_argument("-b", "--bounding-box", default="", help="Six comma separated values used as AABB (write --bounding-box=-1,... when the first value is negative).")
    ap.add_argument("-f", "--min-fraction", type=float, default=0.0)
    ap.add_argument("-p", "--poisson-normals", action="store_true")
    ap.add_argument("-S", "--scale-factor", type=float, default=2.5)
    ap.add_argument("-F", "--fssr", type=int, default=None, help="FSSR output, sets -nsc and -di with scale ARG")
    ap.add_argument("-m", "--mask", default="", help="Name of mask/silhouette image to clip 3D points []")
    ap.add_argument("-C", "--correspondence", action="store_true", help="Output correspondences (in absence of -m and -b only)")
    a = ap.parse_args(argv)
    ext = os.path.splitext(a.mesh_out)[1]
    if ext not in (".ply", ".npts", ".bnpts", ".off"):               # mve::geom::save_mesh, mesh_io.cc:48-66
        ap.error("Extension not recognized")
    if a.fssr is not None:                                           # scene2pset.cc:205-217
        a.with_conf = a.with_normals = a.with_scale = True
        a.depthmap = "depth-L%d" % a.fssr
        a.image = "undistorted" if a.fssr == 0 else "undist-L%d" % a.fssr
    if ext in (".npts", ".bnpts"):                                   # scene2pset.cc:225-231
        a.with_normals, a.with_scale, a.with_conf = True, False, False
    if a.poisson_normals:                                            # :229-233
        a.with_normals = a.with_conf = True
    ids: List[int] = []
    for tok in filter(None, a.views.split(",")):                     # util::Arguments::get_ids_from_string
        if "-" in tok:
            lo, hi = tok.split("-"); ids += list(range(int(lo), int(hi) + 1))
        else:
            ids.append(int(tok))
    aabb = None
    if a.bounding_box:
        aabb = [float(x) for x in a.bounding_box.split(",")]
        if len(aabb) != 6:
            print("Error: Invalid AABB given", file=sys.stderr)
            return 1
    print('Using depthmap "%s" and color image "%s"' % (a.depthmap, a.image))
    ps = scene_to_pointset(a.scene_dir, a.depthmap, a.image, ids or None, aabb, a.min_fraction, a.scale_factor, a.poisson_normals)
    if a.mask:
        print("Filtering points using silhouette masks...")
        ps = mask_filter(a.scene_dir, ps, a.mask)
    print("Writing final point set (%d points)..." % len(ps["pos"]))
    if ext == ".ply":
        write_ply(a.mesh_out, ps, a.with_normals, a.with_scale, a.with_conf)
    elif ext == ".off":
        write_off(a.mesh_out, ps)
    else:
        write_npts(a.mesh_out, ps, binary=(ext == ".bnpts"))
    if a.correspondence and aabb is None and not a.mask:             # scene2pset.cc:374, :481
        write_correspondence(a.mesh_out, ps)
    return 0


if __name__ == "__main__":
    sys.exit(main())
