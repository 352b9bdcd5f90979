/*
 * pointset_device.hip -- depth map -> oriented point set of ONE view on the GPU: the per-view body of
 * apps/scene2pset (scene2pset.cc:262-356), SURVEY 8f "next" row 1.  One lane per pixel / per 2x2 block.
 *
 *   k_ps_cells   the two triangles of every 2x2 block (mve::geom::depthmap_triangulate, libs/mve/depthmap.cc:210-316:
 *                validity mask, shorter diagonal, depth-discontinuity test against the pixel footprints)
 *   k_ps_vertex  per pixel: is it a mesh vertex, world position (pixel_3dpos + cam_to_world, depthmap.cc:149-156,
 *                :377-399), angle-weighted normal over its incident triangles (mesh.cc:45-160), the fan test of
 *                MeshInfo::update_vertex (mesh_info.cc:44-157: SIMPLE / BORDER / COMPLEX, adjacent vertices) and
 *                the scale = scale_factor * mean edge length (scene2pset.cc:343-356)
 *   k_ps_conf    one step of depthmap_mesh_confidences (depthmap.cc:497-546): hop distance to the mesh border
 *
 * The mesh is never materialised: on a depth-map grid a vertex has at most 8 incident triangles, all inside its
 * 3x3 neighbourhood, so every quantity is a local function of the 2x2-block decisions.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pointset_device.h"

/* the positions are compared bit for bit with the reference's unfused float arithmetic */
#pragma clang fp contract(off)

__device__ __forceinline__ float ps_footprint(const PsParams& p, int x, int y, float depth, float& rx, float& ry, float& rn) {
    /* pixel_footprint / pixel_3dpos (depthmap.cc:139-156) */
    rx = p.inv[0] * ((float)x + 0.5f) + p.inv[2];
    ry = p.inv[4] * ((float)y + 0.5f) + p.inv[5];
    rn = sqrtf(rx * rx + ry * ry + 1.f);
    return p.inv[0] * depth / rn;
}

__device__ __forceinline__ bool ps_disc(const float* wd, const float* dp, float ddf, int i1, int i2) {
    /* dm_is_depthdisc, depthmap.cc:188-206 */
    int imin = i1, imax = i2;
    if (dp[i2] < dp[i1]) { imin = i2; imax = i1; }
    if (i1 + i2 == 3) ddf *= 1.41421356237309504880f;
    return dp[imax] - dp[imin] > wd[imin] * ddf;
}

__constant__ int c_tris[4][3] = {{0, 2, 1}, {0, 3, 1}, {0, 2, 3}, {1, 2, 3}};   /* depthmap.cc:253-255 */

/* cells[(y)*(w-1)+x] = tri0 | tri1 << 3   (0 = none, 1..4 = index+1 into c_tris) */
__global__ __launch_bounds__(256) void k_ps_cells(PsParams p, const float* __restrict__ depth, uint8_t* __restrict__ cells) {
    const int cw = p.w - 1, ch = p.h - 1;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cw * ch) return;
    const int y = c / cw, x = c - y * cw;
    const int i = y * p.w + x;
    float dp[4] = {depth[i], depth[i + 1], depth[i + p.w], depth[i + p.w + 1]};
    int mask = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) if (dp[j] > 0.f) mask |= 1 << j;
    int t0 = 0, t1 = 0;
    switch (mask) {
        case 7: t0 = 1; break;
        case 11: t0 = 2; break;
        case 13: t0 = 3; break;
        case 14: t0 = 4; break;
        case 15:
            if (fabsf(dp[0] - dp[3]) < fabsf(dp[1] - dp[2])) { t0 = 2; t1 = 3; } else { t0 = 1; t1 = 4; }
            break;
        default: break;
    }
    if (t0 && p.dd_factor > 0.f) {
        float wd[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float rx, ry, rn;
            wd[j] = dp[j] == 0.f ? 0.f : ps_footprint(p, x + (j & 1), y + (j >> 1), dp[j], rx, ry, rn);
        }
        int tr[2] = {t0, t1};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (tr[j] == 0) continue;
            const int* tv = c_tris[tr[j] - 1];
            if (ps_disc(wd, dp, p.dd_factor, tv[0], tv[1]) || ps_disc(wd, dp, p.dd_factor, tv[1], tv[2])
                || ps_disc(wd, dp, p.dd_factor, tv[2], tv[0]))
                tr[j] = 0;
        }
        /* (a discarded first triangle does not stop the test of the second: depthmap.cc:286 re-reads tri[j]) */
        t0 = tr[0]; t1 = tr[1];
    }
    cells[c] = (uint8_t)(t0 | (t1 << 3));
}

__device__ __forceinline__ void ps_world(const PsParams& p, const float* depth, int px, int py, float* o) {
    float rx, ry, rn;
    const float d = depth[py * p.w + px];
    ps_footprint(p, px, py, d, rx, ry, rn);
    const float cx = rx / rn * d, cy = ry / rn * d, cz = 1.f / rn * d;
    o[0] = p.ctw[0] * cx + p.ctw[1] * cy + p.ctw[2] * cz + p.ctw[3];
    o[1] = p.ctw[4] * cx + p.ctw[5] * cy + p.ctw[6] * cz + p.ctw[7];
    o[2] = p.ctw[8] * cx + p.ctw[9] * cy + p.ctw[10] * cz + p.ctw[11];
}

__device__ __forceinline__ float ps_len(const float* v) { return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

__global__ __launch_bounds__(128) void k_ps_vertex(PsParams p, const float* __restrict__ depth,
                                                  const uint8_t* __restrict__ cells, PsVertex* __restrict__ out) {
    const int idx = blockIdx.x * 128 + threadIdx.x;
    if (idx >= p.w * p.h) return;
    const int y = idx / p.w, x = idx - y * p.w;
    PsVertex V;
    V.used = 0; V.vclass = 3; V.adj = 0; V.level = -1;
    V.pos[0] = V.pos[1] = V.pos[2] = 0.f; V.nrm[0] = V.nrm[1] = V.nrm[2] = 0.f; V.scale = 0.f;
    /* incident triangles in the reference's face order: blocks (x-1,y-1), (x,y-1), (x-1,y), (x,y); in a block
     * tri[0] then tri[1].  In block (bx,by) this pixel is corner (x-bx) + 2*(y-by). */
    int f_first[8], f_second[8], f_cell[8], f_tri[8], f_k[8];
    int nf = 0;
    const int cw = p.w - 1, ch = p.h - 1;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int bx = x - 1 + (b & 1), by = y - 1 + (b >> 1);
        if (bx < 0 || by < 0 || bx >= cw || by >= ch) continue;
        const int corner = (x - bx) + 2 * (y - by);
        const int cfg = cells[by * cw + bx];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int t = (cfg >> (3 * j)) & 7;
            if (!t) continue;
            const int* tv = c_tris[t - 1];
            for (int k = 0; k < 3; ++k)
                if (tv[k] == corner) {
                    const int a = tv[(k + 1) % 3], c2 = tv[(k + 2) % 3];
                    f_first[nf] = (by + (a >> 1)) * p.w + bx + (a & 1);
                    f_second[nf] = (by + (c2 >> 1)) * p.w + bx + (c2 & 1);
                    f_cell[nf] = by * cw + bx; f_tri[nf] = t - 1; f_k[nf] = k;
                    ++nf;
                }
        }
    }
    if (nf == 0) { out[idx] = V; return; }
    V.used = 1;
    ps_world(p, depth, x, y, V.pos);
    /* angle-weighted pseudo normal (mesh.cc:45-120) */
    float n[3] = {0.f, 0.f, 0.f};
    for (int f = 0; f < nf; ++f) {
        const int bx = f_cell[f] % cw, by = f_cell[f] / cw;
        const int* tv = c_tris[f_tri[f]];
        float P[3][3];
        for (int k = 0; k < 3; ++k) ps_world(p, depth, bx + (tv[k] & 1), by + (tv[k] >> 1), P[k]);
        float ab[3], bc[3], ca[3], fn[3];
        for (int k = 0; k < 3; ++k) { ab[k] = P[1][k] - P[0][k]; bc[k] = P[2][k] - P[1][k]; ca[k] = P[0][k] - P[2][k]; }
        /* fn = ab x (-ca) */
        fn[0] = ab[1] * (-ca[2]) - ab[2] * (-ca[1]);
        fn[1] = ab[2] * (-ca[0]) - ab[0] * (-ca[2]);
        fn[2] = ab[0] * (-ca[1]) - ab[1] * (-ca[0]);
        const float fnl = ps_len(fn);
        if (fnl == 0.f) continue;
        const float abl = ps_len(ab), bcl = ps_len(bc), cal = ps_len(ca);
        float ratio;
        if (f_k[f] == 0) ratio = (ab[0] / abl) * (-ca[0] / cal) + (ab[1] / abl) * (-ca[1] / cal) + (ab[2] / abl) * (-ca[2] / cal);
        else if (f_k[f] == 1) ratio = (-ab[0] / abl) * (bc[0] / bcl) + (-ab[1] / abl) * (bc[1] / bcl) + (-ab[2] / abl) * (bc[2] / bcl);
        else ratio = (ca[0] / cal) * (-bc[0] / bcl) + (ca[1] / cal) * (-bc[1] / bcl) + (ca[2] / cal) * (-bc[2] / bcl);
        const float ang = acosf(fminf(fmaxf(ratio, -1.f), 1.f));
        n[0] += fn[0] / fnl * ang; n[1] += fn[1] / fnl * ang; n[2] += fn[2] / fnl * ang;
    }
    const float nl = ps_len(n);
    if (nl > 0.f) { V.nrm[0] = n[0] / nl; V.nrm[1] = n[1] / nl; V.nrm[2] = n[2] / nl; }
    /* fan chaining (mesh_info.cc:73-120) on the (first, second) neighbour pairs */
    bool taken[8];
    for (int f = 0; f < 8; ++f) taken[f] = f >= nf;
    int front = f_first[0], back = f_second[0], left = nf - 1;
    taken[0] = true;
    bool progress = true;
    while (left > 0 && progress) {
        progress = false;
        for (int f = 0; f < nf; ++f) {
            if (taken[f]) continue;
            if (front == f_second[f]) { front = f_first[f]; taken[f] = true; --left; progress = true; break; }
            if (back == f_first[f]) { back = f_second[f]; taken[f] = true; --left; progress = true; break; }
        }
    }
    V.vclass = left > 0 ? 2 : (front == back ? 0 : 1);        /* COMPLEX : SIMPLE : BORDER */
    /* adjacent vertices: every vertex sharing a triangle with this one, as a mask over the 3x3 neighbourhood */
    unsigned adj = 0;
    for (int f = 0; f < nf; ++f) {
        const int a = f_first[f], b = f_second[f];
        adj |= 1u << (((a / p.w) - y + 1) * 3 + ((a % p.w) - x + 1));
        adj |= 1u << (((b / p.w) - y + 1) * 3 + ((b % p.w) - x + 1));
    }
    V.adj = (uint16_t)adj;
    /* scale (scene2pset.cc:343-356) */
    float s = 0.f; int cnt = 0;
    for (int k = 0; k < 9; ++k) {
        if (!((adj >> k) & 1u)) continue;
        float q[3];
        ps_world(p, depth, x + (k % 3) - 1, y + (k / 3) - 1, q);
        q[0] -= V.pos[0]; q[1] -= V.pos[1]; q[2] -= V.pos[2];
        s += ps_len(q); ++cnt;
    }
    V.scale = s / (float)cnt * p.scale_factor;
    V.level = V.vclass == 1 ? 0 : -1;
    out[idx] = V;
}

/* level[v] = it for unassigned vertices that touch a vertex of level it-1 (depthmap.cc:526-545) */
__global__ __launch_bounds__(256) void k_ps_conf(PsParams p, PsVertex* __restrict__ verts, int it) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.w * p.h) return;
    PsVertex& V = verts[idx];
    if (!V.used || V.level >= 0) return;
    const int y = idx / p.w, x = idx - y * p.w;
    for (int k = 0; k < 9; ++k) {
        if (!((V.adj >> k) & 1u)) continue;
        const int q = (y + (k / 3) - 1) * p.w + x + (k % 3) - 1;
        if (verts[q].level == it - 1) { V.level = (int8_t)it; return; }
    }
}

void mi_ps_launch(hipStream_t s, const PsParams& p, const float* depth, uint8_t* cells, PsVertex* verts) {
    const int ncell = (p.w - 1) * (p.h - 1), npix = p.w * p.h;
    hipLaunchKernelGGL(k_ps_cells, dim3((ncell + 255) / 256), dim3(256), 0, s, p, depth, cells);
    hipLaunchKernelGGL(k_ps_vertex, dim3((npix + 127) / 128), dim3(128), 0, s, p, depth, cells, verts);
    for (int it = 1; it < p.conf_iterations; ++it)
        hipLaunchKernelGGL(k_ps_conf, dim3((npix + 255) / 256), dim3(256), 0, s, p, verts, it);
}
