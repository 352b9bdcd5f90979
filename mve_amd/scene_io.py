"""MVE scene directory I/O (the formats on either side of the dmrecon hot path).

Only what the depth-map path consumes and produces, per SURVEY.md section 8f/#3:

* view directory ``views/view_%04d.mve/`` with ``meta.ini`` ([view] id,name;
  [camera] focal_length, pixel_aspect, principal_point, rotation, translation)
  -- reference: libs/mve/view.cc:581-623 (load_meta_data), view.h:22-34
* ``undistorted.png`` RGB8 input embedding -- libs/mve/view.cc:656-709
* MVEI float/uint8 images (``depth-L<s>.mvei`` ...): 11-byte signature
  ``\\x89MVE_IMAGE\\n`` + int32 w,h,c,type + raw row-major interleaved data
  -- libs/mve/image_io.cc:44-47,1260-1321, image_base.h:27-43
* Photosynther bundle ``synth_0.out`` -- libs/mve/bundle_io.cc:276-401

This module is host-side harness code (tests, bench, scene generator); the
product's compute path is the HIP library behind include/mi_dmrecon.h.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

MVEI_SIGNATURE = b"\x89MVE_IMAGE\n"
# mve::ImageType (libs/mve/image_base.h:27-43)
_MVEI_TYPES = {1: np.uint8, 2: np.uint16, 3: np.uint32, 4: np.uint64,
               5: np.int8, 6: np.int16, 7: np.int32, 8: np.int64,
               9: np.float32, 10: np.float64}
_MVEI_TYPE_IDS = {np.dtype(v): k for k, v in _MVEI_TYPES.items()}


@dataclass
class Camera:
    """Mirror of the fields of mve::CameraInfo the path reads (camera.h)."""
    flen: float = 0.0
    paspect: float = 1.0
    ppoint: Sequence[float] = (0.5, 0.5)
    rot: Sequence[float] = (1, 0, 0, 0, 1, 0, 0, 0, 1)   # world->cam, row-major
    trans: Sequence[float] = (0, 0, 0)

    def as_f32(self) -> np.ndarray:
        """[flen, paspect, pp0, pp1, rot(9), trans(3)] as float32 (16 values)."""
        return np.asarray([self.flen, self.paspect, *self.ppoint, *self.rot,
                           *self.trans], dtype=np.float32)

    def position(self) -> np.ndarray:
        r = np.asarray(self.rot, np.float64).reshape(3, 3)
        return -(r.T @ np.asarray(self.trans, np.float64))


@dataclass
class Feature:
    pos: Sequence[float]
    view_ids: Sequence[int]
    color: Sequence[int] = (128, 128, 128)


@dataclass
class SceneData:
    """A whole (small) scene in memory: what apps/dmrecon reads from disk."""
    cameras: List[Camera]
    images: List[Optional[np.ndarray]]          # uint8 HxWx3 per view (level 0)
    features: List[Feature] = field(default_factory=list)

    @property
    def n_views(self) -> int:
        return len(self.cameras)


def _f32s(vals) -> str:
    return " ".join("%.9g" % float(np.float32(v)) for v in vals)


def write_mvei(path: str, arr: np.ndarray) -> None:
    a = np.ascontiguousarray(arr)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, c = a.shape
    with open(path, "wb") as f:
        f.write(MVEI_SIGNATURE)
        f.write(struct.pack("<4i", w, h, c, _MVEI_TYPE_IDS[a.dtype]))
        f.write(a.tobytes())


def read_mvei(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        sig = f.read(len(MVEI_SIGNATURE))
        if sig != MVEI_SIGNATURE:
            raise ValueError("%s: not an MVEI file" % path)
        w, h, c, t = struct.unpack("<4i", f.read(16))
        dt = np.dtype(_MVEI_TYPES[t])
        data = np.frombuffer(f.read(w * h * c * dt.itemsize), dtype=dt)
    return data.reshape(h, w, c)


def write_png(path: str, img: np.ndarray) -> None:
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(img)).save(path, compress_level=1)


def read_png(path: str) -> np.ndarray:
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))


def view_dir(scene_path: str, view_id: int) -> str:
    return os.path.join(scene_path, "views", "view_%04d.mve" % view_id)


def write_scene(scene_path: str, scene: SceneData, embedding: str = "undistorted", raw: bool = False) -> None:
    """Write an MVE scene directory that the unmodified apps/dmrecon accepts.  A view whose image is None gets its
    meta.ini only: MVE then has no such embedding for it and dmrecon skips it as a neighbour (dmrecon.cc:62-79).
    raw=True stores the embedding as <embedding>.mvei (uncompressed; MVE finds image files by extension,
    libs/mve/view.cc:656-709) -- for scenes of gigabytes, where PNG encoding and decoding would dominate."""
    os.makedirs(os.path.join(scene_path, "views"), exist_ok=True)
    for vid, (cam, img) in enumerate(zip(scene.cameras, scene.images)):
        d = view_dir(scene_path, vid)
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "meta.ini"), "w") as f:
            f.write("# MVE view meta data is stored in INI-file syntax.\n")
            f.write("# This file is generated, formatting will get lost.\n\n")
            f.write("[camera]\n")
            f.write("focal_length = %s\n" % _f32s([cam.flen]))
            f.write("pixel_aspect = %s\n" % _f32s([cam.paspect]))
            f.write("principal_point = %s\n" % _f32s(cam.ppoint))
            f.write("rotation = %s\n" % _f32s(cam.rot))
            f.write("translation = %s\n" % _f32s(cam.trans))
            f.write("\n[view]\n")
            f.write("id = %d\n" % vid)
            f.write("name = synth%04d\n" % vid)
        if img is not None:
            if raw:
                write_mvei(os.path.join(d, embedding + ".mvei"), img)
            else:
                write_png(os.path.join(d, embedding + ".png"), img)
    with open(os.path.join(scene_path, "synth_0.out"), "w") as f:
        f.write("drews 1.0\n")
        f.write("%d %d\n" % (scene.n_views, len(scene.features)))
        for cam in scene.cameras:
            f.write("%s 0 0\n" % _f32s([cam.flen]))
            r = list(cam.rot)
            for k in range(3):
                f.write(_f32s(r[3 * k:3 * k + 3]) + "\n")
            f.write(_f32s(cam.trans) + "\n")
        for ft in scene.features:
            f.write(_f32s(ft.pos) + "\n")
            f.write("%d %d %d\n" % tuple(ft.color))
            f.write("%d" % len(ft.view_ids))
            for k, v in enumerate(ft.view_ids):
                f.write(" %d %d 0" % (v, k))
            f.write("\n")


def _parse_ini(path: str) -> dict:
    out, sec = {}, ""
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line[0] in "#;":
                continue
            if line.startswith("[") and line.endswith("]"):
                sec = line[1:-1].strip()
                continue
            if "=" in line:
                k, v = line.split("=", 1)
                out[sec + "." + k.strip()] = v.strip()
    return out


def read_scene(scene_path: str, embedding: str = "undistorted",
               load_images: bool = True) -> SceneData:
    vroot = os.path.join(scene_path, "views")
    entries = sorted(e for e in os.listdir(vroot) if e.endswith(".mve"))
    metas = []
    for e in entries:
        ini = _parse_ini(os.path.join(vroot, e, "meta.ini"))
        metas.append((int(ini.get("view.id", len(metas))), e, ini))
    n = max(m[0] for m in metas) + 1 if metas else 0
    cams: List[Camera] = [Camera() for _ in range(n)]
    imgs: List[Optional[np.ndarray]] = [None] * n
    for vid, e, ini in metas:
        c = Camera()
        if "camera.focal_length" in ini:
            c.flen = float(np.float32(ini["camera.focal_length"]))
        if "camera.pixel_aspect" in ini:
            c.paspect = float(np.float32(ini["camera.pixel_aspect"]))
        if "camera.principal_point" in ini:
            c.ppoint = [float(np.float32(x)) for x in ini["camera.principal_point"].split()]
        if "camera.rotation" in ini:
            c.rot = [float(np.float32(x)) for x in ini["camera.rotation"].split()]
        if "camera.translation" in ini:
            c.trans = [float(np.float32(x)) for x in ini["camera.translation"].split()]
        cams[vid] = c
        p = os.path.join(vroot, e, embedding + ".png")
        if load_images and os.path.exists(p):
            imgs[vid] = read_png(p)
    feats: List[Feature] = []
    bpath = os.path.join(scene_path, "synth_0.out")
    if os.path.exists(bpath):
        with open(bpath) as f:
            if f.readline().strip() != "drews 1.0":
                raise ValueError("unsupported bundle signature")
            tok = f.read().split()
        it = iter(tok)
        ncam, nfeat = int(next(it)), int(next(it))
        for _ in range(ncam * 15):
            next(it)
        for _ in range(nfeat):
            pos = [float(np.float32(next(it))) for _ in range(3)]
            col = [int(float(next(it))) for _ in range(3)]
            nref = int(next(it))
            vids = []
            for _ in range(nref):
                vids.append(int(next(it)))
                next(it)
                next(it)
            feats.append(Feature(pos, vids, col))
    return SceneData(cams, imgs, feats)
