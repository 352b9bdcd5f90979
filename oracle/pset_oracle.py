"""TEST INFRASTRUCTURE ONLY: CPU restatement (numpy + plain loops) of the depth-map -> oriented point-set step
of apps/scene2pset (SURVEY 8f "next" row 1) for ONE view:

  mve::geom::depthmap_triangulate      libs/mve/depthmap.cc:161-399 (2x2-block triangulation, depth
                                       discontinuity test with pixel footprints, colours, cam->world)
  TriangleMesh::recalc_normals         libs/mve/mesh.cc:25-160 (angle-weighted pseudo normals)
  MeshInfo::update_vertex              libs/mve/mesh_info.cc:44-157 (fan chaining, vertex classes, adjacency)
  mve::geom::depthmap_mesh_confidences libs/mve/depthmap.cc:497-546 (boundary distance in 4 steps)
  per-vertex scale                     apps/scene2pset/scene2pset.cc:343-356

It keeps the reference's sequential vertex numbering (first use while scanning the 2x2 blocks) and float32
arithmetic.  Pinned against the reference itself (oracle/_ref/ref_pset_driver, oracle/ref_pset_driver.cc) by
tests/test_pointset.py and the fixture tests/golden/pset_*.npz.  The product never imports this module.
"""
from __future__ import annotations

import math

import numpy as np

F = np.float32
TRIS = ((0, 2, 1), (0, 3, 1), (0, 2, 3), (1, 2, 3))     # depthmap.cc:253-255, indices into the 2x2 block
SQRT2 = F(1.4142135623730951)


def inverse_calibration(cam, w, h):
    """CameraInfo::fill_inverse_calibration (libs/mve/camera.cc:179-200), float32."""
    width, height = F(w), F(h)
    aspect = F(F(width / height) * F(cam.paspect))
    if aspect < 1.0:
        ax = F(F(F(cam.flen) * height) / F(cam.paspect)); ay = F(F(cam.flen) * height)
    else:
        ax = F(F(cam.flen) * width); ay = F(F(F(cam.flen) * width) * F(cam.paspect))
    return np.array([[F(1) / ax, 0, F(-width * F(cam.ppoint[0])) / ax],
                     [0, F(1) / ay, F(-height * F(cam.ppoint[1])) / ay],
                     [0, 0, 1]], dtype=np.float32)


def cam_to_world(cam):
    """CameraInfo::fill_cam_to_world (camera.cc:83-93)."""
    r = np.asarray(cam.rot, np.float32).reshape(3, 3)
    t = np.asarray(cam.trans, np.float32)
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] = r.T
    for i in range(3):
        m[i, 3] = -F(F(F(r[0, i] * t[0]) + F(r[1, i] * t[1])) + F(r[2, i] * t[2]))
    return m


def pointset_from_depthmap(depth, color, cam, dd_factor=5.0, scale_factor=2.5, conf_iterations=4):
    """Returns dict(pixel, pos, normal, color, scale, conf) in the reference's vertex order."""
    depth = np.ascontiguousarray(depth, np.float32)
    h, w = depth.shape
    inv = inverse_calibration(cam, w, h)
    ys, xs = np.mgrid[0:h, 0:w]
    fx = (xs.astype(np.float32) + F(0.5)); fy = (ys.astype(np.float32) + F(0.5))
    # invproj * (x+.5, y+.5, 1): ((0 + m0*x) + m1*y) + m2*1
    rx = (F(0) + inv[0, 0] * fx + inv[0, 1] * fy) + inv[0, 2] * F(1)
    ry = (F(0) + inv[1, 0] * fx + inv[1, 1] * fy) + inv[1, 2] * F(1)
    rz = np.ones_like(rx)
    nrm = np.sqrt((rx * rx + ry * ry) + rz * rz).astype(np.float32)
    footprint = (inv[0, 0] * depth / nrm).astype(np.float32)           # pixel_footprint, depthmap.cc:139-146
    pcam = np.stack([rx / nrm * depth, ry / nrm * depth, rz / nrm * depth], -1).astype(np.float32)   # pixel_3dpos
    ctw = cam_to_world(cam)

    def disc(wd, dp, i1, i2, ddf):                                        # dm_is_depthdisc, depthmap.cc:188-206
        imin, imax = (i1, i2) if not (dp[i2] < dp[i1]) else (i2, i1)
        if i1 + i2 == 3:
            ddf = F(ddf * SQRT2)
        return F(dp[imax] - dp[imin]) > F(wd[imin] * ddf)

    faces = []                 # vertex ids, 3 per face
    vidx = -np.ones(h * w, np.int64)
    pix_of = []
    dflat, fflat = depth.reshape(-1), footprint.reshape(-1)
    for y in range(h - 1):
        for x in range(w - 1):
            i = y * w + x
            ids = (i, i + 1, i + w, i + w + 1)
            dp = [dflat[k] for k in ids]
            mask = sum(1 << j for j in range(4) if dp[j] > 0)
            if mask == 7: tri = [1, 0]
            elif mask == 11: tri = [2, 0]
            elif mask == 13: tri = [3, 0]
            elif mask == 14: tri = [4, 0]
            elif mask == 15:
                tri = [2, 3] if abs(F(dp[0] - dp[3])) < abs(F(dp[1] - dp[2])) else [1, 4]
            else:
                continue
            if dd_factor > 0:
                wd = [fflat[k] for k in ids]
                for j in range(2):
                    if tri[j] == 0:
                        break
                    tv = TRIS[tri[j] - 1]
                    if disc(wd, dp, tv[0], tv[1], F(dd_factor)) or disc(wd, dp, tv[1], tv[2], F(dd_factor)) \
                            or disc(wd, dp, tv[2], tv[0], F(dd_factor)):
                        tri[j] = 0
            for j in range(2):
                if tri[j] == 0:
                    continue
                for tvj in TRIS[tri[j] - 1]:
                    k = ids[tvj]
                    if vidx[k] < 0:
                        vidx[k] = len(pix_of)
                        pix_of.append(k)
                    faces.append(vidx[k])
    n = len(pix_of)
    pix = np.asarray(pix_of, np.int64)
    faces = np.asarray(faces, np.int64).reshape(-1, 3)
    pc = pcam.reshape(-1, 3)[pix] if n else np.zeros((0, 3), np.float32)
    # mesh_transform with ctw (Matrix4f::mult(vec, 1)): ((0 + m0 x) + m1 y) + m2 z) + 1*m3
    pos = np.stack([(F(0) + ctw[r, 0] * pc[:, 0] + ctw[r, 1] * pc[:, 1]) + ctw[r, 2] * pc[:, 2] + F(1) * ctw[r, 3]
                    for r in range(3)], -1).astype(np.float32)
    # --- angle weighted vertex normals (mesh.cc:45-120), accumulated in face order
    vn = np.zeros((n, 3), np.float32)
    if len(faces):
        a, b, c = pos[faces[:, 0]], pos[faces[:, 1]], pos[faces[:, 2]]
        ab, bc, ca = (b - a).astype(np.float32), (c - b).astype(np.float32), (a - c).astype(np.float32)
        fn = np.cross(ab, -ca).astype(np.float32)
        fnl = np.sqrt((fn[:, 0] ** 2 + fn[:, 1] ** 2) + fn[:, 2] ** 2).astype(np.float32)
        ok = fnl != 0
        fnn = np.where(ok[:, None], fn / np.where(ok, fnl, 1)[:, None], fn).astype(np.float32)

        def nlen(v):
            return np.sqrt((v[:, 0] ** 2 + v[:, 1] ** 2) + v[:, 2] ** 2).astype(np.float32)

        def dotn(u, v):
            return ((u[:, 0] * v[:, 0] + u[:, 1] * v[:, 1]) + u[:, 2] * v[:, 2]).astype(np.float32)
        abl, bcl, cal = nlen(ab), nlen(bc), nlen(ca)
        with np.errstate(invalid="ignore", divide="ignore"):
            r1 = dotn(ab / abl[:, None], -ca / cal[:, None])
            r2 = dotn(-ab / abl[:, None], bc / bcl[:, None])
            r3 = dotn(ca / cal[:, None], -bc / bcl[:, None])
            ang = [np.arccos(np.clip(r, F(-1), F(1))).astype(np.float32) for r in (r1, r2, r3)]
        for f in range(len(faces)):                     # sequential adds keep the reference's summation order
            if not ok[f]:
                continue
            for k in range(3):
                vn[faces[f, k]] += fnn[f] * ang[k][f]
        vl = nlen(vn)
        nz = vl > 0
        vn[nz] = (vn[nz] / vl[nz, None]).astype(np.float32)
    # --- MeshInfo: classes + adjacency (mesh_info.cc:24-157)
    vfaces = [[] for _ in range(n)]
    for f in range(len(faces)):
        for k in range(3):
            vfaces[faces[f, k]].append(f)
    SIMPLE, BORDER, COMPLEX, UNREF = 0, 1, 2, 3
    vclass = np.full(n, UNREF, np.int32)
    adj = [[] for _ in range(n)]
    for v in range(n):
        temp = []
        for f in vfaces[v]:
            fv = faces[f]
            for j in range(3):
                if fv[j] == v:
                    temp.append((int(fv[(j + 1) % 3]), int(fv[(j + 2) % 3])))
                    break
        if not temp:
            continue
        srt = [temp.pop(0)]
        while temp:
            front, back = srt[0][0], srt[-1][1]
            found = False
            for t_i, t in enumerate(temp):
                if front == t[1]:
                    srt.insert(0, temp.pop(t_i)); found = True; break
                if back == t[0]:
                    srt.append(temp.pop(t_i)); found = True; break
            if not found:
                break
        if temp:
            s = set()
            for t in srt + temp:
                s.add(t[0]); s.add(t[1])
            adj[v] = sorted(s)
            vclass[v] = COMPLEX
            continue
        vclass[v] = SIMPLE if srt[0][0] == srt[-1][1] else BORDER
        adj[v] = [t[0] for t in srt]
        if vclass[v] == BORDER:
            adj[v].append(srt[-1][1])
    # --- confidences (depthmap.cc:497-546)
    conf = np.ones(n, np.float32)
    cur = [v for v in range(n) if vclass[v] == BORDER]
    for it in range(conf_iterations):
        c = F(F(it) / F(conf_iterations))
        for v in cur:
            conf[v] = c
        nxt = []
        for v in cur:
            for u in adj[v]:
                if conf[u] == 1.0:
                    nxt.append(u)
        cur = nxt
    # --- scale (scene2pset.cc:343-356)
    scale = np.zeros(n, np.float32)
    for v in range(n):
        s = F(0)
        for u in adj[v]:
            d = pos[v] - pos[u]
            s = F(s + F(math.sqrt(F(F(F(d[0] * d[0]) + F(d[1] * d[1])) + F(d[2] * d[2])))))
        scale[v] = F(F(s / F(len(adj[v]))) * F(scale_factor)) if adj[v] else F(np.nan)
    col = np.zeros((n, 3), np.float32)
    if color is not None and n:
        cf = np.asarray(color).reshape(h * w, -1)
        if cf.shape[1] >= 3:
            col = (cf[pix][:, :3].astype(np.float32) / F(255)).astype(np.float32)
        else:
            col = np.repeat((cf[pix][:, :1].astype(np.float32) / F(255)), 3, axis=1).astype(np.float32)
    return dict(pixel=pix.astype(np.int32), pos=pos, normal=vn, color=col, scale=scale, conf=conf,
                vclass=vclass)


def read_ref_dump(path):
    """Parse the binary written by oracle/_ref/ref_pset_driver."""
    raw = open(path, "rb").read()
    n = int(np.frombuffer(raw[:4], np.int32)[0])
    rec = np.frombuffer(raw[4:], dtype=np.dtype([("pixel", np.int32), ("v", np.float32, 11)]), count=n)
    v = rec["v"]
    return dict(pixel=rec["pixel"].copy(), pos=v[:, 0:3].copy(), normal=v[:, 3:6].copy(), color=v[:, 6:9].copy(),
                scale=v[:, 9].copy(), conf=v[:, 10].copy())
