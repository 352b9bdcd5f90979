"""ctypes binding of the C ABI in include/mi_dmrecon.h (mve_amd/csrc/libmi_dmrecon.so).

This is the Python face of the same boundary the C++ shim (mve_amd/host/, class
``mvs::DMRecon``) uses: :class:`Settings` mirrors ``mvs::Settings``
(libs/dmrecon/settings.h:22-52), :class:`DMRecon` mirrors ``mvs::DMRecon``
(libs/dmrecon/dmrecon.h:40-68: ctor checks, ``start()``, ``getProgress()``,
``getRefViewNr()``), with the reference's exception types mapped onto Python's:
``std::invalid_argument`` -> ValueError, ``std::runtime_error`` -> RuntimeError.

There is no CPU fallback: if the HIP library is missing or no GPU is visible,
everything that computes raises.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from .scene_io import SceneData

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI_DMRECON_LIB") or os.path.join(_HERE, "csrc", "libmi_dmrecon.so")   # env: another build of the same ABI

ABI_VERSION = 6                # MI_DMRECON_ABI_VERSION of the header the ctypes mirrors below restate
MAX_GLOBAL_VIEWS = 128
MAX_LOCAL_VIEWS = 16


def local_view_channels(n_local: int) -> int:
    """Channels of the `views` maps / of patch_optimize's local-view arrays: 4, 8 for nrReconNeighbors > 4, 16 above 8
    (mi_dmrecon_local_view_channels)."""
    return 16 if n_local > 8 else 8 if n_local > 4 else 4


E_INVAL, E_GVS, E_DEVICE, E_CANCELLED, E_FOOTPRINT, E_NOIMAGE = -1, -2, -3, -4, -5, -6

EXPORTS = [
    "mi_dmrecon_abi_version", "mi_dmrecon_device_count", "mi_dmrecon_local_view_channels", "mi_dmrecon_last_error", "mi_dmrecon_settings_default",
    "mi_dmrecon_ctx_create", "mi_dmrecon_ctx_destroy", "mi_dmrecon_ctx_fork", "mi_dmrecon_ctx_stream",
    "mi_dmrecon_host_alloc", "mi_dmrecon_host_free",
    "mi_dmrecon_set_view", "mi_dmrecon_set_view_async", "mi_dmrecon_sync", "mi_dmrecon_evict_view", "mi_dmrecon_set_features",
    "mi_dmrecon_num_levels", "mi_dmrecon_level_size", "mi_dmrecon_get_level",
    "mi_dmrecon_global_view_selection", "mi_dmrecon_reconstruct",
    "mi_dmrecon_patch_optimize", "mi_dmrecon_patch_eval", "mi_dmrecon_pointset",
]


class CCamera(ctypes.Structure):
    _fields_ = [("flen", ctypes.c_float), ("paspect", ctypes.c_float), ("ppoint", ctypes.c_float * 2),
                ("rot", ctypes.c_float * 9), ("trans", ctypes.c_float * 3)]


class CSettings(ctypes.Structure):
    _fields_ = [("filterWidth", ctypes.c_int32), ("minNCC", ctypes.c_float), ("minParallax", ctypes.c_float),
                ("acceptNCC", ctypes.c_float), ("minRefineDiff", ctypes.c_float),
                ("maxIterations", ctypes.c_int32), ("nrReconNeighbors", ctypes.c_int32),
                ("globalVSMax", ctypes.c_int32), ("scale", ctypes.c_int32), ("useColorScale", ctypes.c_int32),
                ("aabbMin", ctypes.c_float * 3), ("aabbMax", ctypes.c_float * 3)]


class CPointsetOptions(ctypes.Structure):
    _fields_ = [("dd_factor", ctypes.c_float), ("scale_factor", ctypes.c_float), ("conf_iterations", ctypes.c_int32)]


class CProgress(ctypes.Structure):
    _fields_ = [("status", ctypes.c_int32), ("filled", ctypes.c_uint64), ("queueSize", ctypes.c_uint64),
                ("start_time", ctypes.c_uint64), ("cancelled", ctypes.c_int32)]


class CMaps(ctypes.Structure):
    _fields_ = [("depth", ctypes.c_void_p), ("normal", ctypes.c_void_p), ("dz", ctypes.c_void_p),
                ("conf", ctypes.c_void_p), ("views", ctypes.c_void_p)]


class CStats(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_int64),
                ("n_patch", ctypes.c_int64), ("n_eval", ctypes.c_int64), ("n_filled", ctypes.c_int64),
                ("n_seeds", ctypes.c_int64), ("n_seeds_ok", ctypes.c_int64), ("n_rounds", ctypes.c_int64),
                ("n_launches", ctypes.c_int64), ("ms_total", ctypes.c_double),
                ("ms_opt_kernel", ctypes.c_double), ("ms_sweep_kernels", ctypes.c_double),
                ("ms_bulk_kernel", ctypes.c_double), ("ms_tail_kernel", ctypes.c_double),
                ("n_bulk_launches", ctypes.c_int64), ("n_tail_launches", ctypes.c_int64),
                ("n_pass", ctypes.c_int64), ("truncated", ctypes.c_int64),
                ("n_eval_bulk", ctypes.c_int64), ("n_patch_bulk", ctypes.c_int64), ("n_filled_bulk", ctypes.c_int64),
                ("n_view_replaced", ctypes.c_int64), ("n_iter14", ctypes.c_int64),
                ("gvs_on_device", ctypes.c_int64), ("ms_plan_gvs", ctypes.c_double), ("ms_plan_seeds", ctypes.c_double),
                ("n_merged_calls", ctypes.c_int64), ("merged_into_other_call", ctypes.c_int64),
                ("ms_front_kernel", ctypes.c_double), ("ms_front_view_max", ctypes.c_double),
                ("n_front_launches", ctypes.c_int64), ("front_first_round", ctypes.c_int64),
                ("n_front_views", ctypes.c_int64), ("n_front_rounds_max", ctypes.c_int64),
                ("n_front_rounds_sum", ctypes.c_int64), ("n_front_attempts", ctypes.c_int64),
                ("n_front_entries", ctypes.c_int64), ("front_team", ctypes.c_int64),
                ("front_fallbacks", ctypes.c_int64), ("n_latency_rounds", ctypes.c_int64),
                ("ms_wall_setup", ctypes.c_double), ("ms_wall_rounds", ctypes.c_double), ("ms_wall_front", ctypes.c_double), ("ms_wall_download", ctypes.c_double),
                ("n_patch_turns", ctypes.c_int64), ("n_wave_turns", ctypes.c_int64),
                ("front_team_max", ctypes.c_int64), ("n_sparse_records", ctypes.c_int64),
                ("n_eval_by_kernel", ctypes.c_int64 * 8), ("n_pass_by_kernel", ctypes.c_int64 * 8), ("n_patch_by_kernel", ctypes.c_int64 * 8),
                ("n_pass_executed_by_kernel", ctypes.c_int64 * 8),
                ("ms_latency_rounds", ctypes.c_double), ("n_latency_entries", ctypes.c_int64), ("shader_clock_mhz", ctypes.c_double),
                ("clk_shader_cycles", ctypes.c_int64), ("clk_real_ticks", ctypes.c_int64), ("clk_real_mhz", ctypes.c_double)]


KERNEL_KINDS = ("fast", "follow", "seed", "loop", "spec", "latency", "tail", "front")      # index of the *_by_kernel arrays


_lib = None


def load_library() -> ctypes.CDLL:
    """Load libmi_dmrecon.so (fails loudly if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("HIP library %s is missing - build it with `make -C mve_amd/csrc` "
                           "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float
    L.mi_dmrecon_device_count.restype = ctypes.c_int
    L.mi_dmrecon_abi_version.restype = ctypes.c_int
    if L.mi_dmrecon_abi_version() != ABI_VERSION:
        raise RuntimeError("%s has ABI version %d, mve_amd/api.py mirrors version %d of include/mi_dmrecon.h: rebuild the library"
                           % (LIB_PATH, L.mi_dmrecon_abi_version(), ABI_VERSION))
    L.mi_dmrecon_last_error.restype = ctypes.c_char_p
    L.mi_dmrecon_settings_default.argtypes = [ctypes.POINTER(CSettings)]
    L.mi_dmrecon_settings_default.restype = None
    L.mi_dmrecon_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.mi_dmrecon_ctx_destroy.argtypes = [vp]
    L.mi_dmrecon_ctx_fork.argtypes = [vp, ctypes.POINTER(vp)]
    L.mi_dmrecon_ctx_destroy.restype = None
    L.mi_dmrecon_ctx_stream.argtypes = [vp]
    L.mi_dmrecon_host_alloc.argtypes = [ctypes.c_size_t]
    L.mi_dmrecon_host_alloc.restype = vp
    L.mi_dmrecon_host_free.argtypes = [vp]
    L.mi_dmrecon_host_free.restype = None
    L.mi_dmrecon_ctx_stream.restype = vp
    L.mi_dmrecon_set_view.argtypes = [vp, i32, ctypes.POINTER(CCamera), i32, i32, i32, vp]
    L.mi_dmrecon_set_view_async.argtypes = [vp, i32, ctypes.POINTER(CCamera), i32, i32, i32, vp]
    L.mi_dmrecon_sync.argtypes = [vp]
    L.mi_dmrecon_evict_view.argtypes = [vp, i32]
    L.mi_dmrecon_set_features.argtypes = [vp, i32, vp, vp, vp]
    L.mi_dmrecon_num_levels.argtypes = [vp, i32]
    L.mi_dmrecon_level_size.argtypes = [vp, i32, i32, ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.mi_dmrecon_get_level.argtypes = [vp, i32, i32, vp, vp, vp]
    L.mi_dmrecon_global_view_selection.argtypes = [vp, ctypes.POINTER(CSettings), i32, vp, ctypes.POINTER(i32)]
    L.mi_dmrecon_reconstruct.argtypes = [vp, ctypes.POINTER(CSettings), i32, vp, ctypes.POINTER(CMaps),
                                         ctypes.POINTER(CProgress), vp, ctypes.POINTER(CStats)]
    L.mi_dmrecon_patch_optimize.argtypes = [vp, ctypes.POINTER(CSettings), i32, i32, vp, vp, vp, i32, vp, vp]
    L.mi_dmrecon_patch_eval.argtypes = [vp, ctypes.POINTER(CSettings), i32, i32, i32, f32, f32, f32,
                                        vp, vp, vp, vp, vp, vp]
    L.mi_dmrecon_pointset.argtypes = [vp, ctypes.POINTER(CCamera), i32, i32, vp, vp, i32,
                                      ctypes.POINTER(CPointsetOptions), i32, vp, vp, vp, vp, vp, vp,
                                      ctypes.POINTER(i32)]
    L.mi_dmrecon_debug_inject_footprint.argtypes = [ctypes.c_int]          # test hook, not in the public header
    L.mi_dmrecon_debug_inject_footprint.restype = None
    L.mi_dmrecon_debug_scratch_sets.argtypes = [vp, ctypes.POINTER(ctypes.c_longlong)]  # test hook, not in the public header
    L.mi_dmrecon_debug_region_mark.argtypes = [vp, ctypes.c_int]                        # profiling hook (mi_dmrecon_debug.h)
    L.mi_dmrecon_debug_plan_views_host.argtypes = [i32, vp, vp, vp, i32, vp, vp, vp, ctypes.POINTER(CSettings), i32, i32, i32, vp,
                                                   ctypes.POINTER(i32), ctypes.POINTER(ctypes.c_double), i32, vp, vp,
                                                   ctypes.POINTER(i32)]                                     # test hook
    _lib = L
    return L


def debug_inject_footprint(view_id: int) -> None:
    """Test hook: the reference view `view_id` gets a negative pixel footprint in the calls that follow (-1: none) --
    the condition under which the reference's PatchSampler throws std::out_of_range (patch_sampler.cc:78-82)."""
    load_library().mi_dmrecon_debug_inject_footprint(int(view_id))


def front_teams(n_views: int, n_cus: int = 256, want: int = 32, empty=None):
    """Test hook: the block map of a front launch with teams for n_views reference views on a device of n_cus compute
    units (the library's build_front_teams; no GPU needed).  Returns dict(grid, team_min, team_max, map=[(job, member, team) or None])."""
    L = load_library()
    L.mi_dmrecon_debug_front_teams.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32),
                                               ctypes.POINTER(ctypes.c_int32)]
    cap = 16384
    m = np.zeros(cap, np.uint32)
    e = None if empty is None else np.ascontiguousarray(empty, np.int64)
    g, tmin, tmax = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
    rc = L.mi_dmrecon_debug_front_teams(int(n_views), int(n_cus), int(want), _ptr(e), _ptr(m), cap, ctypes.byref(g), ctypes.byref(tmin),
                                        ctypes.byref(tmax))
    if rc != 0:
        _raise(rc)
    blocks = [None if v == 0xFFFFFFFF else (int(v & 0xFFFF), int((v >> 16) & 0xFF), int(v >> 24)) for v in m[:g.value]]
    return dict(grid=g.value, team_min=tmin.value, team_max=tmax.value, map=blocks)


def plan_views_host(scene: "SceneData", st: "Settings", ref_view: int, tables: bool = True, repeats: int = 1, seeds: bool = False):
    """Test hook: the HOST half of a call's planning -- the global view selection of `ref_view` as a reconstruct call runs
    it when it does not use the device for it (from the scene tables, or directly: tables=False) -- on the scene's cameras,
    image sizes and features alone.  Needs no GPU.  Returns (view ids, milliseconds of `repeats - 1` further selections),
    with seeds=True also the view's seeds as (xy [n, 2], depth [n])."""
    L = load_library()
    n = len(scene.cameras)
    cams = (CCamera * n)()
    widths, heights = np.zeros(n, np.int32), np.zeros(n, np.int32)
    for i, (cam, img) in enumerate(zip(scene.cameras, scene.images)):
        if img is None:
            widths[i], heights[i] = 2, 2                 # (flen 0 below: an invalid view, as one that was never set)
            continue
        cams[i].flen, cams[i].paspect = cam.flen, cam.paspect
        cams[i].ppoint[:] = list(cam.ppoint)
        cams[i].rot[:] = list(cam.rot)
        cams[i].trans[:] = list(cam.trans)
        heights[i], widths[i] = img.shape[0], img.shape[1]
    feats = scene.features
    pos = np.asarray([f.pos for f in feats], np.float32).reshape(-1, 3)
    off = np.zeros(len(feats) + 1, np.int32)
    off[1:] = np.cumsum([len(f.view_ids) for f in feats])
    refs = np.asarray([v for f in feats for v in f.view_ids], np.int32)
    if refs.size == 0:
        refs = np.zeros(1, np.int32)
    out = np.zeros(max(n, 1), np.int32)
    n_out, ms = ctypes.c_int32(0), ctypes.c_double(0.0)
    cs = st.to_c()
    cap = len(feats) if seeds else 0
    sxy, sd, ns = np.zeros((max(cap, 1), 2), np.int32), np.zeros(max(cap, 1), np.float32), ctypes.c_int32(0)
    rc = L.mi_dmrecon_debug_plan_views_host(n, ctypes.cast(cams, ctypes.c_void_p), _ptr(widths), _ptr(heights), len(feats), _ptr(pos),
                                            _ptr(off), _ptr(refs), ctypes.byref(cs), int(ref_view), 1 if tables else 0,
                                            int(repeats), _ptr(out), ctypes.byref(n_out), ctypes.byref(ms),
                                            cap, _ptr(sxy), _ptr(sd), ctypes.byref(ns) if seeds else None)
    if rc != 0:
        _raise(rc)
    ids = [int(v) for v in out[:n_out.value]]
    if seeds:
        return ids, ms.value, (sxy[:ns.value].copy(), sd[:ns.value].copy())
    return ids, ms.value


def device_count() -> int:
    return int(load_library().mi_dmrecon_device_count())


class PinnedArray:
    """A numpy view on page-locked host memory from mi_dmrecon_host_alloc (freed with the object)."""

    def __init__(self, shape, dtype):
        self._L = load_library()
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._p = self._L.mi_dmrecon_host_alloc(max(n, 1))
        if not self._p:
            raise RuntimeError(self._L.mi_dmrecon_last_error().decode())
        buf = (ctypes.c_char * max(n, 1)).from_address(self._p)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def __del__(self):
        if getattr(self, "_p", None):
            self._L.mi_dmrecon_host_free(self._p)
            self._p = None


def _ptr(a: Optional[np.ndarray]):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else None


def _raise(rc: int):
    msg = load_library().mi_dmrecon_last_error().decode("utf-8", "replace")
    if rc == E_INVAL:
        raise ValueError(msg)                 # std::invalid_argument
    if rc == E_FOOTPRINT:
        raise IndexError(msg)                 # std::out_of_range
    if rc == E_CANCELLED:
        raise InterruptedError(msg)
    raise RuntimeError(msg)                   # std::runtime_error


@dataclass
class Settings:
    """mvs::Settings (libs/dmrecon/settings.h:22-52), same names and defaults."""
    refViewNr: int = 0
    imageEmbedding: str = "undistorted"
    filterWidth: int = 5
    minNCC: float = 0.3
    minParallax: float = 10.0
    acceptNCC: float = 0.6
    minRefineDiff: float = 0.001
    maxIterations: int = 20
    nrReconNeighbors: int = 4
    globalVSMax: int = 20
    scale: int = 0
    useColorScale: bool = True
    writePlyFile: bool = False
    aabbMin: Sequence[float] = field(default_factory=lambda: [-np.finfo(np.float32).max] * 3)
    aabbMax: Sequence[float] = field(default_factory=lambda: [np.finfo(np.float32).max] * 3)
    plyPath: str = ""
    keepDzMap: bool = False
    keepConfidenceMap: bool = False
    quiet: bool = False

    def to_c(self) -> CSettings:
        s = CSettings()
        s.filterWidth, s.minNCC, s.minParallax = self.filterWidth, self.minNCC, self.minParallax
        s.acceptNCC, s.minRefineDiff, s.maxIterations = self.acceptNCC, self.minRefineDiff, self.maxIterations
        s.nrReconNeighbors, s.globalVSMax, s.scale = self.nrReconNeighbors, self.globalVSMax, self.scale
        s.useColorScale = 1 if self.useColorScale else 0
        s.aabbMin[:] = [float(v) for v in self.aabbMin]
        s.aabbMax[:] = [float(v) for v in self.aabbMax]
        return s


class Context:
    """One GPU context (= one scene cache on one device).  Not thread-safe."""

    def __init__(self, device: int = 0):
        self._L = load_library()
        h = ctypes.c_void_p()
        rc = self._L.mi_dmrecon_ctx_create(device, ctypes.byref(h))
        if rc != 0:
            _raise(rc)
        self._h = h
        self.device = device
        self.n_views = 0
        self._keep = None

    def fork(self) -> "Context":
        """A sibling context sharing this one's resident scene, with its own stream (one per host thread)."""
        h = ctypes.c_void_p()
        rc = self._L.mi_dmrecon_ctx_fork(self._h, ctypes.byref(h))
        if rc != 0:
            _raise(rc)
        c = Context.__new__(Context)
        c._L, c._h, c.device, c.n_views, c._keep = self._L, h, self.device, self.n_views, None
        return c

    def debug_scratch_sets(self):
        """Test hook: (free scratch sets of this context's scene, pixel capacity of the largest)."""
        px = ctypes.c_longlong(0)
        n = self._L.mi_dmrecon_debug_scratch_sets(self._h, ctypes.byref(px))
        return int(n), int(px.value)

    def debug_region_mark(self, tag=0):
        """Profiling hook: an empty one-lane kernel (k_region_mark) on this context's stream, waited for -- where
        tools/trace_regions.py cuts a rocprofv3 kernel trace."""
        rc = self._L.mi_dmrecon_debug_region_mark(self._h, int(tag))
        if rc != 0:
            _raise(rc)

    def close(self):
        if getattr(self, "_h", None):
            self._L.mi_dmrecon_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    @property
    def stream(self) -> int:
        return int(self._L.mi_dmrecon_ctx_stream(self._h) or 0)

    # -- scene upload --------------------------------------------------------
    def set_view(self, view_id: int, cam, image: np.ndarray, asynchronous: bool = False):
        """asynchronous=True only enqueues (image must stay alive, ideally pinned, until sync())."""
        c = CCamera()
        c.flen, c.paspect = cam.flen, cam.paspect
        c.ppoint[:] = list(cam.ppoint)
        c.rot[:] = list(cam.rot)
        c.trans[:] = list(cam.trans)
        img = np.ascontiguousarray(image, np.uint8)
        if img.ndim == 2:
            img = img[:, :, None]
        fn = self._L.mi_dmrecon_set_view_async if asynchronous else self._L.mi_dmrecon_set_view
        rc = fn(self._h, view_id, ctypes.byref(c), img.shape[1], img.shape[0], img.shape[2], _ptr(img))
        if rc != 0:
            _raise(rc)
        self.n_views = max(self.n_views, view_id + 1)

    def set_view_camera_only(self, view_id: int, cam, width: int, height: int):
        """Registers a view with its camera and image size only (mi_dmrecon_set_view with pixels = NULL): a candidate of the
        view selections whose image could not be loaded; a reference view that selects it ends with E_NOIMAGE."""
        c = CCamera()
        c.flen, c.paspect = cam.flen, cam.paspect
        c.ppoint[:] = list(cam.ppoint)
        c.rot[:] = list(cam.rot)
        c.trans[:] = list(cam.trans)
        rc = self._L.mi_dmrecon_set_view(self._h, view_id, ctypes.byref(c), int(width), int(height), 3, None)
        if rc != 0:
            _raise(rc)
        self.n_views = max(self.n_views, view_id + 1)

    def sync(self):
        rc = self._L.mi_dmrecon_sync(self._h)
        if rc != 0:
            _raise(rc)

    def set_features(self, features):
        pos = np.asarray([f.pos for f in features], np.float32).reshape(-1, 3)
        off = np.zeros(len(features) + 1, np.int32)
        off[1:] = np.cumsum([len(f.view_ids) for f in features])
        refs = np.asarray([v for f in features for v in f.view_ids], np.int32)
        if refs.size == 0:
            refs = np.zeros(1, np.int32)
        rc = self._L.mi_dmrecon_set_features(self._h, len(features), _ptr(pos), _ptr(off), _ptr(refs))
        if rc != 0:
            _raise(rc)

    def load_scene(self, scene: SceneData, pinned_staging: bool = False):
        """Upload every view (pyramids are built on the device) and the features.
        pinned_staging=True streams the images through two page-locked buffers with asynchronous
        uploads: the host copies view i+1 into one buffer while the GPU ingests view i from the other."""
        if not pinned_staging:
            for vid, (cam, img) in enumerate(zip(scene.cameras, scene.images)):
                if img is not None:
                    self.set_view(vid, cam, img)
        else:
            nmax = max(int(np.prod(im.shape)) for im in scene.images if im is not None)
            ring = [PinnedArray((nmax,), np.uint8) for _ in range(2)]
            k = 0
            for vid, (cam, img) in enumerate(zip(scene.cameras, scene.images)):
                if img is None:
                    continue
                if k >= 2:
                    self.sync()                  # the buffer about to be reused has been consumed
                buf = ring[k % 2].array[:img.size].reshape(img.shape)
                np.copyto(buf, img)
                self.set_view(vid, cam, buf, asynchronous=True)
                k += 1
            self.sync()
        self.set_features(scene.features)

    # -- queries -------------------------------------------------------------
    def num_levels(self, view_id: int) -> int:
        n = self._L.mi_dmrecon_num_levels(self._h, view_id)
        if n < 0:
            _raise(n)
        return n

    def level_size(self, view_id: int, level: int):
        w, h = ctypes.c_int32(), ctypes.c_int32()
        rc = self._L.mi_dmrecon_level_size(self._h, view_id, level, ctypes.byref(w), ctypes.byref(h))
        if rc != 0:
            _raise(rc)
        return w.value, h.value

    def get_level(self, view_id: int, level: int):
        w, h = self.level_size(view_id, level)
        rgb = np.zeros((h, w, 3), np.uint8)
        proj = np.zeros(9, np.float32)
        inv = np.zeros(9, np.float32)
        rc = self._L.mi_dmrecon_get_level(self._h, view_id, level, _ptr(rgb), _ptr(proj), _ptr(inv))
        if rc != 0:
            _raise(rc)
        return rgb, proj, inv

    def global_view_selection(self, st: Settings, ref_view: Optional[int] = None) -> List[int]:
        ids = np.zeros(MAX_GLOBAL_VIEWS, np.int32)
        n = ctypes.c_int32()
        cs = st.to_c()
        rc = self._L.mi_dmrecon_global_view_selection(self._h, ctypes.byref(cs),
                                                      st.refViewNr if ref_view is None else ref_view,
                                                      _ptr(ids), ctypes.byref(n))
        if rc != 0:
            _raise(rc)
        return [int(v) for v in ids[:n.value]]

    # -- the hot path ----------------------------------------------------------
    def alloc_outputs(self, st: Settings, ref_views: Sequence[int], want_normal=True, want_views=False,
                      pinned=False) -> List[Dict]:
        """Output buffers for reconstruct(out=...); pinned=True puts them in page-locked memory."""
        out = []
        for r in ref_views:
            w, h = self.level_size(int(r), st.scale)
            spec = dict(depth=((h, w), np.float32), dz=((h, w, 2), np.float32), conf=((h, w), np.float32))
            if want_normal:
                spec["normal"] = ((h, w, 3), np.float32)
            if want_views:
                spec["views"] = ((h, w, local_view_channels(st.nrReconNeighbors)), np.int32)
            d = {}
            for k, (shape, dt) in spec.items():
                if pinned:
                    pa = PinnedArray(shape, dt)
                    d["_pin_" + k] = pa
                    d[k] = pa.array
                else:
                    d[k] = np.zeros(shape, dt)
            out.append(d)
        return out

    def reconstruct(self, st: Settings, ref_views: Sequence[int], want_normal=True, want_views=False,
                    progress: Optional[ctypes.Array] = None, out: Optional[List[Dict]] = None) -> List[Dict]:
        """DMRecon::start() for a batch of reference views; returns one dict of maps per view."""
        n = len(ref_views)
        refs = np.asarray(ref_views, np.int32)
        cs = st.to_c()
        maps = (CMaps * n)()
        if out is None:
            out = self.alloc_outputs(st, ref_views, want_normal, want_views)
        for i, r in enumerate(ref_views):
            d = out[i]
            maps[i].depth, maps[i].dz, maps[i].conf = _ptr(d["depth"]), _ptr(d["dz"]), _ptr(d["conf"])
            maps[i].normal = _ptr(d.get("normal"))
            maps[i].views = _ptr(d.get("views"))
        status = np.zeros(n, np.int32)
        stats = CStats()
        stats.struct_size = ctypes.sizeof(CStats)             # the library fills what the caller has room for (mi_dmrecon.h)
        rc = self._L.mi_dmrecon_reconstruct(self._h, ctypes.byref(cs), n, _ptr(refs), maps, progress,
                                            _ptr(status), ctypes.byref(stats))
        if rc != 0:
            _raise(rc)
        self.last_stats = {}
        for k, _ in CStats._fields_:
            v = getattr(stats, k)
            if isinstance(v, ctypes.Array):               # the per-kernel-template arrays: one key per template (KERNEL_KINDS)
                for name, x in zip(KERNEL_KINDS, v):
                    self.last_stats["%s.%s" % (k, name)] = x
            else:
                self.last_stats[k] = v
        for i in range(n):
            out[i]["status"] = int(status[i])
        return out

    def patch_optimize(self, st: Settings, ref_view: int, xy, hyp, local=None, lanes_per_view: int = 1):
        xy = np.ascontiguousarray(xy, np.int32).reshape(-1, 2)
        n = len(xy)
        hyp = np.ascontiguousarray(hyp, np.float32).reshape(n, 3)
        nch = local_view_channels(st.nrReconNeighbors)
        loc = None
        if local is not None:
            local = np.ascontiguousarray(local, np.int32).reshape(n, -1)
            loc = np.full((n, nch), -1, np.int32)
            loc[:, :local.shape[1]] = local                      # (a 4-column array also serves an 8-channel call)
        out = np.zeros((n, 8), np.float32)
        out_local = np.zeros((n, nch), np.int32)
        cs = st.to_c()
        rc = self._L.mi_dmrecon_patch_optimize(self._h, ctypes.byref(cs), ref_view, n, _ptr(xy), _ptr(hyp),
                                               _ptr(loc), int(lanes_per_view), _ptr(out), _ptr(out_local))
        if rc != 0:
            _raise(rc)
        return out, out_local

    def patch_eval(self, st: Settings, ref_view: int, x: int, y: int, depth: float, dzi=0.0, dzj=0.0):
        g = MAX_GLOBAL_VIEWS
        master = np.zeros(5, np.float32)
        ncc = np.zeros(g, np.float32)
        ok = np.zeros(g, np.int32)
        ns = int(st.filterWidth) ** 2
        col = np.zeros((g, ns, 3), np.float32)
        der = np.zeros((g, ns, 3), np.float32)
        lvl = np.zeros(g, np.int32)
        cs = st.to_c()
        n = self._L.mi_dmrecon_patch_eval(self._h, ctypes.byref(cs), ref_view, x, y, depth, dzi, dzj,
                                          _ptr(master), _ptr(ncc), _ptr(ok), _ptr(col), _ptr(der), _ptr(lvl))
        if n < 0:
            _raise(n)
        return dict(master=master, ncc=ncc[:n], ok=ok[:n], col=col[:n], deriv=der[:n], level=lvl[:n])

    def pointset(self, cam, depth: np.ndarray, color: Optional[np.ndarray] = None, dd_factor: float = 5.0,
                 scale_factor: float = 2.5, conf_iterations: int = 4):
        """apps/scene2pset per-view body (scene2pset.cc:262-356): depth map -> oriented points.

        ``cam`` is a scene_io/synth camera (flen, paspect, ppoint, rot, trans).  Returns a dict of arrays
        pixel, pos, normal, color, scale, conf in ascending pixel order."""
        depth = np.ascontiguousarray(depth, np.float32)
        h, w = depth.shape
        cc = CCamera()
        cc.flen = float(cam.flen); cc.paspect = float(cam.paspect)
        cc.ppoint[:] = [float(v) for v in cam.ppoint]
        cc.rot[:] = [float(v) for v in np.asarray(cam.rot, np.float32).reshape(9)]
        cc.trans[:] = [float(v) for v in np.asarray(cam.trans, np.float32).reshape(3)]
        nch = 0
        cptr = None
        if color is not None:
            color = np.ascontiguousarray(color, np.uint8).reshape(h, w, -1)
            nch = color.shape[2]
            cptr = _ptr(color)
        opt = CPointsetOptions(dd_factor, scale_factor, conf_iterations)
        cap = int(np.count_nonzero(depth > 0))
        out = dict(pixel=np.zeros(cap, np.int32), pos=np.zeros((cap, 3), np.float32),
                   normal=np.zeros((cap, 3), np.float32), color=np.zeros((cap, 3), np.float32),
                   scale=np.zeros(cap, np.float32), conf=np.zeros(cap, np.float32))
        n = ctypes.c_int32(0)
        rc = self._L.mi_dmrecon_pointset(self._h, ctypes.byref(cc), w, h, _ptr(depth), cptr, nch, ctypes.byref(opt),
                                         cap, _ptr(out["pixel"]), _ptr(out["pos"]), _ptr(out["normal"]),
                                         _ptr(out["color"]), _ptr(out["scale"]), _ptr(out["conf"]), ctypes.byref(n))
        if rc < 0:
            _raise(rc)
        return {k: v[:n.value] for k, v in out.items()}


class DMRecon:
    """mvs::DMRecon (libs/dmrecon/dmrecon.h:40-68) over the C ABI.

    ``scene`` is a :class:`Context` that already holds the views and features
    (the role ``mve::Scene::Ptr`` plays in the reference).  ``start()`` stores the
    maps under the embedding names the reference uses (dmrecon.cc:119-145).
    """

    def __init__(self, scene: Context, settings: Settings):
        self.scene = scene
        self.settings = settings
        self.progress = CProgress()
        self.images: Dict[str, np.ndarray] = {}
        # ctor checks of dmrecon.cc:37-46,74-75
        if settings.refViewNr < 0 or settings.refViewNr >= scene.n_views:
            raise ValueError("Master view index out of bounds")
        if settings.scale < 0:
            raise ValueError("Invalid scale factor")
        if not settings.imageEmbedding:
            raise ValueError("Invalid image embedding")
        try:
            self.width, self.height = scene.level_size(settings.refViewNr, settings.scale)
        except ValueError:
            raise ValueError("Invalid master view")

    def getRefViewNr(self) -> int:
        return self.settings.refViewNr

    def getProgress(self) -> CProgress:
        return self.progress

    def start(self):
        st = self.settings
        arr = (CProgress * 1)(self.progress)
        try:
            res = self.scene.reconstruct(st, [st.refViewNr], progress=arr)[0]
        except InterruptedError:
            self.progress.status = 5      # RECON_CANCELLED, nothing is written (dmrecon.cc:101-105)
            return
        finally:
            for k, _ in CProgress._fields_:
                if k != "cancelled":
                    setattr(self.progress, k, getattr(arr[0], k))
        name = "-L%d" % st.scale
        self.images["depth" + name] = res["depth"]
        if st.keepDzMap:
            self.images["dz" + name] = res["dz"]
        if st.keepConfidenceMap:
            self.images["conf" + name] = res["conf"]
        if st.scale != 0:
            self.images["undist" + name] = self.scene.get_level(st.refViewNr, st.scale)[0]
        self.normal = res.get("normal")
        self.progress.filled = int((res["conf"] > 0).sum())
