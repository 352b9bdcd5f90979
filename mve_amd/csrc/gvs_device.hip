/*
 * gvs_device.hip -- DMRecon::analyzeFeatures + GlobalViewSelection::performVS on the device, for all reference
 * views of a call at once (libs/dmrecon/dmrecon.cc:178-208, global_view_selection.cc:33-101).
 *
 * Input: the scene tables the host builds once per scene (SceneGeom, dmrecon_host.cpp: which view sees which
 * feature, the feature's depth in every view, the parallax between two views at a feature).  One workgroup per
 * reference view runs the reference's greedy loop.  The selection must be IDENTICAL to the reference's, so every
 * float operation is the reference's own, correctly rounded and uncontracted (__f*_rn), in the reference's order:
 *   - the per-feature score of a candidate is base x pen(s_0) x pen(s_1) ... over the selected views in ascending id
 *     order (benefitFromView iterates a std::set) -- independent per feature, computed by all threads;
 *   - a candidate's benefit is the sum of its scores in ascending feature order -- ONE thread per candidate adds
 *     them one after the other (a parallel reduction would round differently and could flip a near tie); features
 *     the candidate does not see contribute +0.f, which leaves every partial sum as it is;
 *   - arg-max with strict '>' in ascending view order (global_view_selection.cc:45-52).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gvs_device.h"

#define GVS_THREADS 1024
#define GVS_TILE_FLOATS 8192                /* LDS tile of scores: 32 KB */

__device__ __forceinline__ float gvs_penalty(float plx) {                /* (plx / 10)^2, global_view_selection.cc:78,97 */
    const float q = __fdiv_rn(plx, 10.f);
    return __fmul_rn(q, q);
}

__global__ __launch_bounds__(GVS_THREADS) void k_gvs(GvsArgs a) {
    __shared__ int s_wave_cnt[GVS_THREADS / 64];
    __shared__ int s_nfeat, s_nsel, s_found, s_best;
    __shared__ int s_sel[MI_GVS_MAX_OUT];
    __shared__ uint8_t s_avail[1024];
    __shared__ float s_tile[GVS_TILE_FLOATS + 64];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const GvsScene& S = a.sc;
    const int nv = S.nv, nf = S.nf, ref = a.refs[r].ref;
    int32_t* feat = a.feat + (size_t)r * nf;
    float* base = a.base + (size_t)r * nv * nf;
    float* benefit = a.benefit + (size_t)r * nv;
    const uint8_t* sees_ref = S.sees + (size_t)ref * nf;

    /* ---- features attached to the reference view (dmrecon.cc:185-196): ordered compaction */
    if (tid == 0) { s_nfeat = 0; s_nsel = 0; }
    __syncthreads();
    for (int f0 = 0; f0 < nf; f0 += GVS_THREADS) {
        const int f = f0 + tid;
        bool keep = f < nf && sees_ref[f] != 0;
        if (keep && a.use_box) {
            const float* p = S.fpos + 3 * (size_t)f;
            for (int k = 0; k < 3; ++k) if (p[k] < a.aabb_min[k] || p[k] > a.aabb_max[k]) keep = false;
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = s_nfeat;
        for (int w = 0; w < wave; ++w) off += s_wave_cnt[w];
        if (keep) feat[off + __popcll(m & ((1ull << lane) - 1ull))] = f;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < GVS_THREADS / 64; ++w) t += s_wave_cnt[w]; s_nfeat += t; }
        __syncthreads();
    }
    const int nfeat = s_nfeat;
    for (int i = tid; i < nv; i += GVS_THREADS) s_avail[i] = (i != ref && S.valid[i]) ? 1 : 0;   /* global_view_selection.cc:23-30 */
    __threadfence_block();
    __syncthreads();

    /* ---- the part of benefitFromView's score that does not depend on the selected set (:76-89) */
    const float inv_m = a.refs[r].inv_m;
    const float* plx_ref = S.plx + (size_t)ref * nv * nf;
    const float* z_ref = S.zcam + (size_t)ref * nf;
    for (size_t idx = tid; idx < (size_t)nv * nfeat; idx += GVS_THREADS) {
        const int i = (int)(idx / nfeat), l = (int)(idx - (size_t)i * nfeat);
        if (!s_avail[i]) continue;
        const int f = feat[l];
        if (!S.sees[(size_t)i * nf + f]) continue;
        float sc = 1.f;
        const float plx = plx_ref[(size_t)i * nf + f];
        if (plx < a.minParallax) sc = __fmul_rn(sc, gvs_penalty(plx));
        const float mfp = __fmul_rn(z_ref[f], inv_m);                                  /* footPrintScaled */
        const float nfp = __fmul_rn(S.zcam[(size_t)i * nf + f], S.inv0[i]);             /* footPrint */
        float ratio = __fdiv_rn(mfp, nfp);
        if ((double)ratio > 2.) ratio = (float)__ddiv_rn(2., (double)ratio);
        else if ((double)ratio > 1.) ratio = 1.f;
        base[(size_t)i * nf + l] = __fmul_rn(sc, ratio);
    }
    __syncthreads();

    /* ---- greedy selection (global_view_selection.cc:33-60) */
    const int nvp = nv | 1;                                              /* odd row length: the transposed tile writes spread over the banks */
    const int TL = max(1, min(64, GVS_TILE_FLOATS / nvp));               /* features per tile */
    for (;;) {
        const int nsel = s_nsel;
        if (nsel >= a.globalVSMax || nsel >= MI_GVS_MAX_OUT) break;
        /* The scores of a tile of features for every candidate, by all threads, into LDS; then thread i adds candidate
         * i's column to its running benefit, feature after feature.  Nothing but the tile leaves the registers. */
        float b = 0.f;                                                   /* benefit of candidate `tid` */
        for (int l0 = 0; l0 < nfeat; l0 += TL) {
            const int tl = min(TL, nfeat - l0);
            for (int idx = tid; idx < nv * tl; idx += GVS_THREADS) {
                const int i = idx / tl, lt = idx - i * tl, l = l0 + lt;
                if (!s_avail[i]) continue;
                const int f = feat[l];
                float sc = 0.f;                                          /* a feature the candidate does not see adds +0: x + 0 = x */
                if (S.sees[(size_t)i * nf + f]) {
                    sc = base[(size_t)i * nf + l];
                    for (int q = 0; q < nsel; ++q) {
                        const int sv = s_sel[q];
                        if (!S.sees[(size_t)sv * nf + f]) continue;                     /* :93 */
                        const float plx = S.plx[((size_t)sv * nv + i) * nf + f];
                        if (plx < a.minParallax) sc = __fmul_rn(sc, gvs_penalty(plx)); /* :96-98; otherwise x 1 */
                    }
                }
                s_tile[lt * nvp + i] = sc;
            }
            __syncthreads();
            if (tid < nv && s_avail[tid]) {
#pragma unroll 8
                for (int lt = 0; lt < tl; ++lt) b = __fadd_rn(b, s_tile[lt * nvp + tid]);
            }
            __syncthreads();
        }
        if (tid < nv) benefit[tid] = b;
        __threadfence_block();
        __syncthreads();
        if (tid == 0) {
            float maxBenefit = 0.f; int best = -1;
            for (int i = 0; i < nv; ++i)
                if (s_avail[i] && benefit[i] > maxBenefit) { maxBenefit = benefit[i]; best = i; }
            s_found = best >= 0; s_best = best;
            if (best >= 0) {
                int p = nsel;                                            /* keep the list ascending (std::set) */
                while (p > 0 && s_sel[p - 1] > best) { s_sel[p] = s_sel[p - 1]; --p; }
                s_sel[p] = best;
                s_avail[best] = 0;
                s_nsel = nsel + 1;
            }
        }
        __syncthreads();
        if (!s_found) break;
    }
    __syncthreads();
    if (tid < MI_GVS_MAX_OUT) a.out_ids[(size_t)r * MI_GVS_MAX_OUT + tid] = tid < s_nsel ? s_sel[tid] : -1;
    if (tid == 0) a.out_n[r] = s_nsel;
}

void mi_gvs_launch(hipStream_t s, const GvsArgs& a, int n_refs) {
    if (n_refs <= 0) return;
    hipLaunchKernelGGL(k_gvs, dim3(n_refs), dim3(GVS_THREADS), 0, s, a);
}
