"""Deterministic synthetic MVE scenes for the dmrecon hot path.

A textured smooth height field z = h(x, y) near z = 0 is observed by N pinhole
cameras placed on a ring around the optical axis at distance ~10, all looking
roughly along +z.  Images are rendered by exact ray / height-field intersection
(fixed-point iteration), so analytic ground-truth *radial* depth (reference
quirk Q1: libs/mve/depthmap.cc:149-156 normalises view rays) is available.

Choices that matter for parity work: camera z positions
are jittered so that the neighbour/reference footprint ratio is not sitting
exactly on the 0.5 mip-level threshold of patch_sampler.cc:85-91.

Host-side harness code only (tests / bench); not part of the compute path.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

from .scene_io import Camera, Feature, SceneData

N_WAVES = 24


@dataclass
class SynthParams:
    n_views: int = 5
    width: int = 640
    height: int = 480
    texture_seed: int = 7
    feature_seed: int = 1
    camera_seed: int = 3
    n_features: int = 2000
    distance: float = 10.0
    ring: Tuple[float, float] = (0.6, 1.0)
    z_jitter: float = 0.45
    rot_jitter_deg: float = 1.5
    bump_amp: float = 0.35
    flen: float = 1.0
    fmin: float = 0.3     # texture cycles per world unit
    fmax: float = 12.0


def surface_height(p: SynthParams, x, y):
    return p.bump_amp * np.sin(1.7 * x + 0.3) * np.sin(2.3 * y - 0.2)


def _rodrigues(axis, ang):
    axis = axis / np.linalg.norm(axis)
    k = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * k + (1 - np.cos(ang)) * (k @ k)


def make_cameras(p: SynthParams) -> List[Camera]:
    rng = np.random.RandomState(p.camera_seed)
    cams = []
    for i in range(p.n_views):
        a = 2 * np.pi * (i + 0.25 * rng.uniform(-1, 1)) / max(p.n_views, 1)
        r = rng.uniform(*p.ring)
        c = np.array([r * np.cos(a), r * np.sin(a),
                      -p.distance + p.z_jitter * rng.uniform(-1, 1)])
        rot = _rodrigues(rng.normal(size=3), np.deg2rad(p.rot_jitter_deg) * rng.uniform(-1, 1))
        rot32 = rot.astype(np.float32)
        trans32 = (-(rot32.astype(np.float64) @ c)).astype(np.float32)
        cams.append(Camera(flen=p.flen, paspect=1.0, ppoint=(0.5, 0.5),
                           rot=[float(v) for v in rot32.reshape(-1)],
                           trans=[float(v) for v in trans32]))
    return cams


def _calib(cam: Camera, w: int, h: int):
    """K for a w x h image -- same rule as libs/mve/camera.cc:124-144."""
    aspect = (w / h) * cam.paspect
    if aspect < 1.0:
        ax, ay = cam.flen * h / cam.paspect, cam.flen * h
    else:
        ax, ay = cam.flen * w, cam.flen * w * cam.paspect
    return ax, ay, w * cam.ppoint[0], h * cam.ppoint[1]


def pixel_rays(cam: Camera, w: int, h: int, xs=None, ys=None):
    """World-space (unnormalised, z_cam = 1) rays through pixel centres."""
    ax, ay, cx, cy = _calib(cam, w, h)
    if xs is None:
        ys, xs = np.mgrid[0:h, 0:w]
    dx = (xs + 0.5 - cx) / ax
    dy = (ys + 0.5 - cy) / ay
    d_cam = np.stack([dx, dy, np.ones_like(dx)], axis=-1).astype(np.float64)
    r = np.asarray(cam.rot, np.float64).reshape(3, 3)
    return d_cam @ r          # row-vector times R == R^T d


def intersect(p: SynthParams, cam: Camera, rays: np.ndarray):
    """Ray parameter t (along the unnormalised ray) of the surface hit."""
    c = cam.position()
    t = (0.0 - c[2]) / rays[..., 2]
    for _ in range(12):
        x = c[0] + t * rays[..., 0]
        y = c[1] + t * rays[..., 1]
        t = (surface_height(p, x, y) - c[2]) / rays[..., 2]
    return t


def _texture_basis(p: SynthParams):
    rng = np.random.RandomState(p.texture_seed)
    f = np.exp(rng.uniform(np.log(p.fmin), np.log(p.fmax), N_WAVES))
    th = rng.uniform(0, 2 * np.pi, N_WAVES)
    fx, fy = 2 * np.pi * f * np.cos(th), 2 * np.pi * f * np.sin(th)
    amp = f ** -0.35
    phase = rng.uniform(0, 2 * np.pi, (N_WAVES, 3))
    # sin(a + phi) = sin a cos phi + cos a sin phi  ->  [sin a, cos a] @ mix
    mix = np.concatenate([amp[:, None] * np.cos(phase), amp[:, None] * np.sin(phase)], axis=0)
    norm = np.sqrt(0.5 * np.sum(amp ** 2))
    return fx, fy, mix / (2.2 * norm)


class _SynthParamsC(ctypes.Structure):
    _fields_ = [("cam_pos", ctypes.c_double * 3), ("rot", ctypes.c_double * 9),
                ("ax", ctypes.c_double), ("ay", ctypes.c_double),
                ("cx", ctypes.c_double), ("cy", ctypes.c_double),
                ("bump_amp", ctypes.c_double), ("n_waves", ctypes.c_int32),
                ("width", ctypes.c_int32), ("height", ctypes.c_int32)]


_LIB = None


def _lib():
    """The host-only OpenMP renderer (mve_amd/csrc/synth_render.cc)."""
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libmi_synth.so")
        if not os.path.exists(path):
            raise RuntimeError("%s missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
        _LIB = ctypes.CDLL(path)
        _LIB.mi_synth_render.restype = None
    return _LIB


def _render(p: SynthParams, cam: Camera, w: int, h: int, ppoint, want_rgb: bool, want_depth: bool):
    c2 = Camera(cam.flen, cam.paspect, ppoint if ppoint is not None else cam.ppoint, cam.rot, cam.trans)
    ax, ay, cx, cy = _calib(c2, w, h)
    fx, fy, mix = _texture_basis(p)
    cp = _SynthParamsC()
    cp.cam_pos[:] = list(cam.position())
    cp.rot[:] = [float(v) for v in cam.rot]
    cp.ax, cp.ay, cp.cx, cp.cy = ax, ay, cx, cy
    cp.bump_amp, cp.n_waves, cp.width, cp.height = p.bump_amp, N_WAVES, w, h
    fx = np.ascontiguousarray(fx, np.float64)
    fy = np.ascontiguousarray(fy, np.float64)
    mix = np.ascontiguousarray(mix, np.float64)
    rgb = np.empty((h, w, 3), np.uint8) if want_rgb else None
    dep = np.empty((h, w), np.float32) if want_depth else None
    vp = ctypes.c_void_p
    _lib().mi_synth_render(ctypes.byref(cp), vp(fx.ctypes.data), vp(fy.ctypes.data), vp(mix.ctypes.data),
                           vp(rgb.ctypes.data if want_rgb else None),
                           vp(dep.ctypes.data if want_depth else None))
    return rgb, dep


def render_view(p: SynthParams, cam: Camera) -> np.ndarray:
    return _render(p, cam, p.width, p.height, None, True, False)[0]


def true_depth(p: SynthParams, cam: Camera, w: int, h: int, ppoint=None) -> np.ndarray:
    """Ground-truth radial depth map for a (possibly down-scaled) w x h level."""
    return _render(p, cam, w, h, ppoint, False, True)[1]


def project(cam: Camera, w: int, h: int, pts: np.ndarray):
    r = np.asarray(cam.rot, np.float64).reshape(3, 3)
    pc = pts @ r.T + np.asarray(cam.trans, np.float64)
    ax, ay, cx, cy = _calib(cam, w, h)
    z = pc[:, 2]
    return ax * pc[:, 0] / z + cx - 0.5, ay * pc[:, 1] / z + cy - 0.5, z


def make_features(p: SynthParams, cams: List[Camera]) -> List[Feature]:
    rng = np.random.RandomState(p.feature_seed)
    half_w = 0.5 * p.distance / p.flen * 1.05
    half_h = half_w * p.height / p.width
    xy = np.stack([rng.uniform(-half_w, half_w, p.n_features),
                   rng.uniform(-half_h, half_h, p.n_features)], axis=1)
    pts = np.concatenate([xy, surface_height(p, xy[:, 0], xy[:, 1])[:, None]], axis=1)
    pts32 = pts.astype(np.float32)
    vis = []
    for cam in cams:
        u, v, z = project(cam, p.width, p.height, pts32.astype(np.float64))
        vis.append((z > 0) & (u >= 0) & (u <= p.width - 1) & (v >= 0) & (v <= p.height - 1))
    vis = np.stack(vis, axis=1)
    feats = []
    for i in range(p.n_features):
        ids = [int(v) for v in np.nonzero(vis[i])[0]]
        if len(ids) < 2:
            continue
        feats.append(Feature([float(v) for v in pts32[i]], ids))
    return feats


def make_scene(p: SynthParams) -> SceneData:
    cams = make_cameras(p)
    return SceneData(cams, [render_view(p, c) for c in cams], make_features(p, cams))


# The BASELINE.json configurations (SURVEY.md section 8 header).
CONFIGS = {
    "C1": dict(params=SynthParams(n_views=2, width=640, height=480), scale=0, local_neighbors=1),
    "C2": dict(params=SynthParams(n_views=8, width=1280, height=720), scale=1, local_neighbors=4),
    "C3": dict(params=SynthParams(n_views=20, width=1920, height=1080), scale=2, local_neighbors=4),
    "C5": dict(params=SynthParams(n_views=100, width=4032, height=3024, ring=(0.4, 1.6)), scale=3,
               local_neighbors=4),
}
