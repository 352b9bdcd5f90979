"""TEST INFRASTRUCTURE ONLY: ctypes wrapper around oracle/liboracle_dmrecon.so
(the CPU restatement, oracle/dmrecon_oracle.cc) and helpers to run the
compiled reference under oracle/_ref/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product (mve_amd) never does.
"""
from __future__ import annotations

import ctypes
import os
import shutil
import subprocess
import tempfile
from typing import List, Optional, Sequence

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle_dmrecon.so")
REF_DIR = os.path.join(HERE, "_ref")


class OrcCamera(ctypes.Structure):
    _fields_ = [("flen", ctypes.c_float), ("paspect", ctypes.c_float), ("ppoint", ctypes.c_float * 2),
                ("rot", ctypes.c_float * 9), ("trans", ctypes.c_float * 3)]


class OrcSettings(ctypes.Structure):
    _fields_ = [("refViewNr", ctypes.c_int32), ("filterWidth", ctypes.c_int32),
                ("minNCC", ctypes.c_float), ("minParallax", ctypes.c_float),
                ("acceptNCC", ctypes.c_float), ("minRefineDiff", ctypes.c_float),
                ("maxIterations", ctypes.c_int32), ("nrReconNeighbors", ctypes.c_int32),
                ("globalVSMax", ctypes.c_int32), ("scale", ctypes.c_int32),
                ("useColorScale", ctypes.c_int32),
                ("aabbMin", ctypes.c_float * 3), ("aabbMax", ctypes.c_float * 3)]


class OrcStats(ctypes.Structure):
    _fields_ = [("n_patch", ctypes.c_int64), ("n_eval", ctypes.c_int64), ("n_filled", ctypes.c_int64),
                ("n_seeds_ok", ctypes.c_int64), ("n_seeds", ctypes.c_int64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            subprocess.check_call(["make", "-s", "-C", HERE, "restatement"])
        L = ctypes.CDLL(LIB_PATH)
        L.orc_scene_create.restype = ctypes.c_void_p
        L.orc_scene_create.argtypes = [ctypes.c_int]
        L.orc_scene_destroy.argtypes = [ctypes.c_void_p]
        L.orc_settings_default.argtypes = [ctypes.POINTER(OrcSettings)]
        L.orc_scene_set_view.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(OrcCamera),
                                         ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.orc_scene_set_features.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3
        L.orc_pyramid_levels.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_pyramid_get.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4
        L.orc_global_vs.argtypes = [ctypes.c_void_p, ctypes.POINTER(OrcSettings), ctypes.c_void_p]
        L.orc_reconstruct.argtypes = [ctypes.c_void_p, ctypes.POINTER(OrcSettings)] + [ctypes.c_void_p] * 4 + [ctypes.POINTER(OrcStats)]
        L.orc_patch_optimize.argtypes = [ctypes.c_void_p, ctypes.POINTER(OrcSettings), ctypes.c_int] + [ctypes.c_void_p] * 5
        L.orc_patch_eval.argtypes = [ctypes.c_void_p, ctypes.POINTER(OrcSettings), ctypes.c_int, ctypes.c_int,
                                     ctypes.c_float, ctypes.c_float, ctypes.c_float] + [ctypes.c_void_p] * 6
        _lib = L
    return _lib


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else None


def make_settings(ref_view=0, scale=0, local_neighbors=4, global_max=20, **kw) -> OrcSettings:
    s = OrcSettings()
    lib().orc_settings_default(ctypes.byref(s))
    s.refViewNr, s.scale, s.nrReconNeighbors, s.globalVSMax = ref_view, scale, local_neighbors, global_max
    for k, v in kw.items():
        setattr(s, k, v)
    return s


class OracleScene:
    """Scene handle for the CPU restatement; takes an mve_amd.scene_io.SceneData."""

    def __init__(self, scene):
        self.n_views = scene.n_views
        self.h = lib().orc_scene_create(scene.n_views)
        for vid, (cam, img) in enumerate(zip(scene.cameras, scene.images)):
            if img is None:
                continue
            c = OrcCamera()
            c.flen, c.paspect = cam.flen, cam.paspect
            c.ppoint[:] = list(cam.ppoint)
            c.rot[:] = list(cam.rot)
            c.trans[:] = list(cam.trans)
            im = np.ascontiguousarray(img, np.uint8)
            lib().orc_scene_set_view(self.h, vid, ctypes.byref(c), im.shape[1], im.shape[0], _ptr(im))
        pos = np.asarray([f.pos for f in scene.features], np.float32).reshape(-1, 3)
        off = np.zeros(len(scene.features) + 1, np.int32)
        off[1:] = np.cumsum([len(f.view_ids) for f in scene.features])
        refs = np.asarray([v for f in scene.features for v in f.view_ids], np.int32)
        self._keep = (pos, off, refs)
        lib().orc_scene_set_features(self.h, len(scene.features), _ptr(pos), _ptr(off), _ptr(refs))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_scene_destroy(self.h)
            self.h = None

    def pyramid_levels(self, view: int) -> int:
        return lib().orc_pyramid_levels(self.h, view)

    def pyramid_level(self, view: int, level: int):
        wh = np.zeros(2, np.int32)
        proj = np.zeros(9, np.float32)
        inv = np.zeros(9, np.float32)
        lib().orc_pyramid_get(self.h, view, level, _ptr(wh), None, _ptr(proj), _ptr(inv))
        img = np.zeros((wh[1], wh[0], 3), np.uint8)
        lib().orc_pyramid_get(self.h, view, level, None, _ptr(img), None, None)
        return img, proj, inv

    def global_vs(self, st: OrcSettings) -> List[int]:
        ids = np.zeros(max(self.n_views, 1), np.int32)
        n = lib().orc_global_vs(self.h, ctypes.byref(st), _ptr(ids))
        if n < 0:
            raise RuntimeError("oracle global_vs failed")
        return [int(v) for v in ids[:n]]

    def reconstruct(self, st: OrcSettings):
        img, _, _ = self.pyramid_level(st.refViewNr, st.scale)
        h, w = img.shape[:2]
        depth = np.zeros((h, w), np.float32)
        normal = np.zeros((h, w, 3), np.float32)
        dz = np.zeros((h, w, 2), np.float32)
        conf = np.zeros((h, w), np.float32)
        stats = OrcStats()
        rc = lib().orc_reconstruct(self.h, ctypes.byref(st), _ptr(depth), _ptr(normal), _ptr(dz), _ptr(conf),
                                   ctypes.byref(stats))
        if rc != 0:
            raise RuntimeError("oracle reconstruct failed rc=%d" % rc)
        return dict(depth=depth, normal=normal, dz=dz, conf=conf,
                    stats={k: getattr(stats, k) for k, _ in OrcStats._fields_})

    def patch_optimize(self, st: OrcSettings, xy, hyp, local=None):
        xy = np.ascontiguousarray(xy, np.int32).reshape(-1, 2)
        n = len(xy)
        hyp = np.ascontiguousarray(hyp, np.float32).reshape(n, 3)
        loc = np.full((n, 16), -1, np.int32)                    # the hook carries up to sixteen local views (ORC_MAX_LOCAL)
        if local is not None:
            local = np.ascontiguousarray(local, np.int32).reshape(n, -1)
            loc[:, :local.shape[1]] = local
        out = np.zeros((n, 8), np.float32)
        out_local = np.zeros((n, 16), np.int32)
        rc = lib().orc_patch_optimize(self.h, ctypes.byref(st), n, _ptr(xy), _ptr(hyp), _ptr(loc), _ptr(out), _ptr(out_local))
        if rc != 0:
            raise RuntimeError("oracle patch_optimize failed")
        return out, out_local[:, :(16 if st.nrReconNeighbors > 8 else 8 if st.nrReconNeighbors > 4 else 4)]

    def patch_eval(self, st: OrcSettings, x, y, depth, dzi=0.0, dzj=0.0):
        g = max(self.n_views, 1)
        master = np.zeros(5, np.float32)
        ncc = np.zeros(g, np.float32)
        ok = np.zeros(g, np.int32)
        ns = int(st.filterWidth) ** 2
        col = np.zeros((g, ns, 3), np.float32)
        der = np.zeros((g, ns, 3), np.float32)
        lvl = np.zeros(g, np.int32)
        n = lib().orc_patch_eval(self.h, ctypes.byref(st), x, y, depth, dzi, dzj, _ptr(master), _ptr(ncc), _ptr(ok),
                                 _ptr(col), _ptr(der), _ptr(lvl))
        if n < 0:
            raise RuntimeError("oracle patch_eval failed")
        return dict(master=master, ncc=ncc[:n], ok=ok[:n], col=col[:n], deriv=der[:n], level=lvl[:n])


# ---------------------------------------------------------------------------
# The compiled reference (oracle/_ref), when present.

def ref_available(which: str = "dmrecon_ref_strict") -> bool:
    return os.path.exists(os.path.join(REF_DIR, which))


def run_reference_app(scene_dir: str, scale: int, local_neighbors: int = 4, master: Optional[int] = None,
                      flavour: str = "strict", threads: Optional[int] = None, extra: Sequence[str] = ()):
    """Run the unmodified apps/dmrecon (built by oracle/Makefile) on a COPY of scene_dir.
    Returns (copy_dir, wall_seconds)."""
    import time
    exe = os.path.join(REF_DIR, "dmrecon_ref_" + flavour)
    work = tempfile.mkdtemp(prefix="refscene_")
    dst = os.path.join(work, "scene")
    shutil.copytree(scene_dir, dst)
    cmd = [exe, "-s%d" % scale, "--local-neighbors=%d" % local_neighbors, "--keep-conf", "--keep-dz",
           "--force", "--progress=silent"]
    if master is not None:
        cmd.append("-m%d" % master)
    cmd += list(extra) + [dst]
    env = dict(os.environ)
    if threads:
        env["OMP_NUM_THREADS"] = str(threads)
    t0 = time.time()
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, env=env)
    return dst, time.time() - t0


def run_reference_patch_driver(scene_dir: str, ref_view: int, scale: int, local_neighbors: int, mode: str,
                               seeds: Sequence[Sequence[float]], filter_width: int = 5, global_max: int = 20) -> List[List[str]]:
    exe = os.path.join(REF_DIR, "ref_patch_driver")
    with tempfile.TemporaryDirectory() as td:
        sp, op = os.path.join(td, "seeds.txt"), os.path.join(td, "out.txt")
        with open(sp, "w") as f:
            for s in seeds:
                x, y, d, dzi, dzj = s[:5]
                loc = [int(v) for v in s[5:] if int(v) >= 0]
                f.write("%d %d %.9g %.9g %.9g %d %s\n" % (x, y, d, dzi, dzj, len(loc), " ".join(map(str, loc))))
        subprocess.run([exe, scene_dir, str(ref_view), str(scale), str(local_neighbors), mode, sp, op],
                       check=True, stdout=subprocess.DEVNULL,
                       env=dict(os.environ, REF_FILTER_WIDTH=str(filter_width), REF_GLOBAL_VS_MAX=str(global_max)))
        with open(op) as f:
            return [ln.split() for ln in f if ln.strip()]
