/*
 * dmrecon_device.hip -- gfx950 kernels of the MI355X-native dmrecon hot path.
 *
 * What the reference does one pixel at a time on one CPU thread
 * (libs/dmrecon: DMRecon::processQueue -> PatchOptimization -> PatchSampler),
 * this file does as data-parallel sweeps:
 *
 *   k_generate  one lane per reference-image pixel: "did a 4-neighbour get a
 *               better hypothesis last round?" (the push rule of
 *               dmrecon.cc:400-431 turned into a pull), ballot-compacted into a
 *               dense work list.
 *   k_optimize  the hot kernel.  A pixel's patch optimisation
 *               (patch_optimization.cc:170-242) is run by a QUAD of lanes, one
 *               lane per local neighbour view (nrReconNeighbors = 4): each lane
 *               projects the 25 patch points into its view, gathers the
 *               bilinear RGBA8 texels, and the Gauss-Newton / NCC sums are
 *               combined with DPP quad permutes -- no LDS round trip, no
 *               per-lane view loop.  A 64-wide wavefront therefore advances 16
 *               pixels x 4 views in lock step.  Per-patch data shared by the quad
 *               (25 master colours, 25 view rays, NCC scratch for view
 *               selection) and the sRGB->linear table live in LDS.
 *               (A second layout, Lay<16>, spends a whole wavefront on one
 *               pixel for the latency-bound tail.)  A bulk round is two
 *               launches: first attempts, then the follow-up list.
 *   k_apply     writes accepted results back to the state maps (Jacobi sweep:
 *               all of a round's optimisations read the previous round's state).
 *   k_tail      one fused round of the propagation tail: candidates from the
 *               previous round's accepted pixels, claim, optimisation (Lay<16>),
 *               write into the pixel's other state slot; k_flatten folds the
 *               slots at the end.
 *   k_pyramid   byte-exact 4x4 Gaussian half-size pyramid (image_tools.h:619-690).
 *
 * No MFMA: every reduction here is a 75-term dot product per lane.
 * Reference citations are relative to /root/reference/libs/dmrecon unless noted.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include <type_traits>

#include "dmrecon_types.h"
#include "dmrecon_device.h"

#define WAVE 64

/* This file is compiled once per supported filter width (mvs::Settings::filterWidth, apps/dmrecon --filter-width):
 * -DMI_FW=3 / 5 / 7 / 9 / 11.  Everything that depends on it lives in a namespace of its own; the host picks the table of
 * launchers (MiDeviceApi, dmrecon_device.h) that belongs to the call's settings. */
#ifndef MI_FW
#define MI_FW 5
#endif
#define MI_HALF (MI_FW / 2)             /* PatchSampler::offset */
#define MI_NS (MI_FW * MI_FW)           /* PatchSampler::nrSamples */
#define MI_MID (MI_NS / 2)              /* the centre sample, patchPoints[nrSamples / 2] */
/* Q3: fastColAndDeriv measures its derivative step at patchPoints[12] whatever the filter width (patch_sampler.cc:96);
 * that is the centre only for width 5.  Width 7: sample 12 = row 1, column 5, kept.  Width 3 has no sample 12 -- the
 * reference reads past the end of the vector there (undefined behaviour); the centre is used instead. */
#define MI_STEP_SAMPLE (MI_NS > 12 ? 12 : MI_MID)
#define MI_CAT2(a, b) a##b
#define MI_CAT(a, b) MI_CAT2(a, b)
#define MI_FWNS MI_CAT(mi_fw, MI_FW)
/* wavefronts per SIMD the bulk kernels aim for: width 7 needs 18.8 KB of rays / master colours per wavefront in LDS,
 * which caps the occupancy anyway -- let the compiler use the registers */
#define MI_BULK_WAVES (MI_FW >= 7 ? 1 : MI_WAVES_PER_SIMD)
/* Words per footprint record.  -DMI_EMU_LIN48 (an experiment, `make variant`): 12 -- the MEMORY side of a record that would hold
 * the footprint's bilinear coefficients as twelve f32 (c00, d1, d2, d3 per channel: no table look-up, no differences in the
 * sampler) instead of four RGBA8 texels: records of 48 bytes, of which the sampler still computes with the first 16 but
 * gathers all 48 (three 16-byte loads kept alive), so that the arithmetic is unchanged and the run shows what the three-fold
 * gather costs by itself (profiles/r6_ab_experiments.txt). */
#ifdef MI_EMU_LIN48
#define MI_QUAD_WORDS 12
#elif defined(MI_QUAD_RECORDS)
/* the layout of rounds 2-5 (an experiment build now): one 16-byte record per texel position = its 2 x 2 footprint, every texel
 * four times */
#define MI_QUAD_WORDS 4
#else
/* Column pairs (the default since round 6): element (x, y) = texels (x, y), (x, y + 1) -- 8 bytes; a sample's 2 x 2 footprint =
 * elements x and x + 1 of row y = ONE 16-byte gather from an 8-byte aligned address.  Half the bytes per texel position of the
 * 16-byte footprint records (every texel twice instead of four times): the footprints of a wavefront's 16 neighbouring patches cover
 * half as many cache lines, and that -- the lines a wavefront's gathers touch in the vector L1 -- is what the throughput layout's
 * time goes with (profiles/r6_ab_experiments.txt D, I, J: bulk kernels - 4 %, a view's images 12 instead of 20 bytes per texel). */
#define MI_PAIR_RECORDS 1
#define MI_QUAD_WORDS 2
#endif

namespace MI_FWNS {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef u32x2 u32x2_a4 __attribute__((aligned(4)));      /* gfx950 global loads only need dword alignment */
/* Pointers that are themselves loaded from memory (DevView::img, DevJob maps) have no provable address
 * space, so hipcc emits flat_load for them -- which also ticks the LDS counter and serialises with the
 * table lookups.  These typedefs pin them to the global address space (global_load). */
typedef const __attribute__((address_space(1))) u32x2_a4* gtex2_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u32x4* gtex4_t;
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
typedef const __attribute__((address_space(1))) u32x4_a4* gtex4u_t;   /* dword-aligned 16-byte load */
typedef const __attribute__((address_space(1))) uint32_t* gtex_t;
typedef const __attribute__((address_space(1))) float* gf32_t;
typedef const __attribute__((address_space(1))) int32_t* gi32_t;
#define GF(p) (*(gf32_t)(p))          /* global-address-space loads of map elements */
#define GI(p) (*(gi32_t)(p))
#define GU(p) (*(gtex_t)(p))

/* Wavefronts per SIMD the optimise kernels are compiled for: 3 = 168 VGPRs (a few cold-path spills); 2 (256 VGPRs,
 * no spills) measured ~10 % slower, 4 spills 88 registers. */
#ifndef MI_WAVES_PER_SIMD
#define MI_WAVES_PER_SIMD 3
#endif

#ifdef MI_PROBE
/* development aid (tools/patch_probe.py, `make -C mve_amd/csrc probe`): the first lane of a wavefront logs (id, shader
 * clock) pairs of the patch it is working on into LDS; k_front copies the log of an attempt into the debug buffer */
#define MI_PROBE_LOG 96
__shared__ unsigned long long g_plog[8][MI_PROBE_LOG];
__shared__ unsigned g_pidx[8];
__device__ __forceinline__ void probe_stamp(unsigned id) {
    if ((threadIdx.x & 63u) == 0) {
        const unsigned w = (threadIdx.x >> 6) & 7u, k = g_pidx[w];
        if (k < MI_PROBE_LOG) { g_plog[w][k] = ((unsigned long long)id << 56) | (__builtin_readcyclecounter() & 0x00FFFFFFFFFFFFFFull); g_pidx[w] = k + 1; }
    }
}
#define TSTAMP(id) probe_stamp(id)
#else
#define TSTAMP(id) do { } while (0)
#endif

/* LDS of one 64-lane workgroup = 16 patches (file scope so that the one non-inlined
 * device function below addresses it with ds_* instructions, not flat ones). */
__shared__ float g_lut[256];                                   /* sRGB -> linear, mvs_tools.cc:22-93 */
__shared__ float g_geo[MI_PATCHES_PER_WAVE][MI_NS];            /* 1 / |K^-1 (pixel of sample i)|: the unit-ray scale of PatchSampler::masterViewDirs */
__shared__ float g_mcol[MI_PATCHES_PER_WAVE][3 * MI_NS];       /* PatchSampler::masterColorSamples */
#ifdef MI_LDS_WINDOW
/*
 * LDS windows (north_star: "neighbour-view pixels staged through LDS"; an experiment build, `make variant VFLAGS=-DMI_LDS_WINDOW`).
 * A wavefront of the throughput layout holds 16 neighbouring patches x 4 view slots; the 25 x 16 footprints a pass gathers from one
 * neighbour image lie in a box of a few hundred texels.  Per pass and view slot the wavefront's active lanes find that box (the four
 * window corners of every patch, reduced with LDS atomics -- the lanes of a pass are an arbitrary subset of the wavefront), copy it
 * from the level's plain RGBA8 plane into a tile of MI_WIN_TP x MI_WIN_TH texels with coalesced 16-byte loads, and the samples read
 * their 2 x 2 footprints from the tile (two ds_read2_b32) instead of gathering them from memory.  Whenever the four slots' patches do
 * not share one image and level each, a box does not fit, or a sample falls outside its box (the box comes from the corners), the
 * pass runs on the global gathers -- decided per wavefront and pass; same texels, same arithmetic, same bits either way.
 */
#ifndef MI_WIN_TP
#define MI_WIN_TP 28          /* tile pitch = largest box width, texels (a multiple of 4: rows are filled in 16-byte pieces) */
#endif
#ifndef MI_WIN_TH
#define MI_WIN_TH 24          /* largest box height */
#endif
__shared__ __attribute__((aligned(16))) uint32_t g_win[4][MI_WIN_TH * MI_WIN_TP];
__shared__ __attribute__((aligned(16))) int g_wbox[4][4];          /* per view slot: min x, min y, max x, max y (texel indices, inclusive) */
__shared__ __attribute__((aligned(16))) unsigned g_wimg[4][4];     /* per view slot: the level's plane (address lo, hi), its width, unused */
__shared__ unsigned g_wstat[4];                                    /* wavefront-passes of this workgroup: on windows, redone after a miss, other images, box too large */
#endif
/* LocalViewSelection ncc[] of the throughput layouts: L::PATCHES x DevSettings::ncc_stride floats of DYNAMIC shared memory, sized
 * by the launcher -- 64 per patch unless globalVSMax asks for more (MI_MAX_GLOBAL = 128): the 4 KB a wavefront of 16 patches has
 * always had; 8 KB per wavefront would cost the general kernels a wavefront per SIMD (160 KB per CU, 12 wavefronts) */
extern __shared__ float g_ncc_dyn[];
/* the same for the latency layout: one patch per wavefront, at most MI_LAT_SLOTS wavefronts per workgroup.  Separate
 * (smaller) arrays because a kernel's LDS is what it references: the tail kernels then ask for 4 KB instead of 12.8 KB,
 * and a CU filled with bulk workgroups of another call (12 x 12.6 KB of 160 KB) has that much to spare */
#ifndef MI_LAT_SLOTS
#define MI_LAT_SLOTS 8
#endif
__shared__ float g_geo_lat[MI_LAT_SLOTS][MI_NS];
__shared__ float g_mcol_lat[MI_LAT_SLOTS][3 * MI_NS];
__shared__ float g_ncc_lat[MI_LAT_SLOTS][MI_MAX_GLOBAL];
template <class L> __device__ __forceinline__ float* lds_geo(int patch) { return L::LAT ? g_geo_lat[patch] : g_geo[patch]; }
template <class L> __device__ __forceinline__ float* lds_mcol(int patch) { return L::LAT ? g_mcol_lat[patch] : g_mcol[patch]; }
template <class L> __device__ __forceinline__ float* lds_ncc(int patch, int stride) { return L::LAT ? g_ncc_lat[patch] : g_ncc_dyn + patch * stride; }


/* ------------------------------------------------------------------------- */
/* Lane layouts.  A patch is optimised by 4 "view slots" (one per local neighbour view);
 * each view slot is LPV lanes wide and its lanes split the 25 samples of a pass.
 *   Lay<1>  : view slot = 1 lane, patch = a quad, 16 patches per wavefront.  Throughput layout
 *             (every lane busy on its own view) used while the work list is large.
 *   Lay<16> : view slot = a 16-lane DPP row, patch = the whole wavefront.  Latency layout for
 *             the long tail of small propagation rounds: a pass is 2 samples deep instead of 25.
 * All cross-lane traffic is DPP (quad_perm / row mirrors) plus a few readlanes; no LDS. */

__device__ __forceinline__ int dpp_xor1(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ int dpp_xor2(int v) { return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true); }
__device__ __forceinline__ int dpp_half_mirror(int v) { return __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true); }
__device__ __forceinline__ int dpp_mirror(int v) { return __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true); }
template <int K> __device__ __forceinline__ int dpp_bcast(int v) { return __builtin_amdgcn_mov_dpp(v, K * 0x55, 0xF, 0xF, true); }

__device__ __forceinline__ float fadd_i(float a, int b) { return a + __int_as_float(b); }
__device__ __forceinline__ double dmov(double v, int (*f)(int)) {
    return __hiloint2double(f(__double2hiint(v)), f(__double2loint(v)));
}
__device__ __forceinline__ unsigned long long dmov_u(unsigned long long v, int (*f)(int)) {
    return ((unsigned long long)(unsigned)f((int)(unsigned)(v >> 32)) << 32) | (unsigned)f((int)(unsigned)v);
}
__device__ __forceinline__ double shfl_xor_d(double v, int m) {
    return __hiloint2double(__shfl_xor(__double2hiint(v), m), __shfl_xor(__double2loint(v), m));
}

/* 1-ulp hardware reciprocal / square root / rsqrt: the parity tolerances (1e-5 on colours, 1e-3 on
 * depth) leave six orders of magnitude of room, and the IEEE sequences cost ~10 instructions each */
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }

/* two floats that travel together through the packed-f32 instructions of gfx950 (v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32):
 * an even-aligned register pair; a value that is the same for both halves is read twice from one register (op_sel) */
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 mk2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ f2 sp2(float a) { return mk2(a, a); }

/* view sets are 8-bit indices into the job's global list, MI_VIEW_NONE padded, ascending (std::set order): NV of them */
template <int NV> struct ViewPack;
template <> struct ViewPack<4> { typedef uint32_t type; static constexpr uint32_t NONE = 0xFFFFFFFFu; };
template <> struct ViewPack<8> { typedef unsigned long long type; static constexpr unsigned long long NONE = ~0ull; };

template <int LPV, int NV> struct Lay;

/* The value of the lane 16 / 32 lanes away (lane ^ 16, lane ^ 32): exchanges ACROSS the DPP rows of a wavefront, which DPP itself
 * cannot do.  gfx950 has them as VALU instructions -- v_permlane16_swap_b32 vdst, src: the odd rows of vdst and the even rows
 * of src change places (with both = v the partner row's value arrives in the second result for the even rows 0, 2 and in the
 * first for the odd rows 1, 3); v_permlane32_swap_b32: the upper half of vdst and the lower half of src change places -- no
 * trip through the scalar unit (v_readlane + its hazards) or the LDS crossbar (-DMI_T_SHFL: ds_bpermute). */
#ifdef MI_T_SHFL
__device__ __forceinline__ int x16(int v) { return __shfl_xor(v, 16); }
__device__ __forceinline__ int x32(int v) { return __shfl_xor(v, 32); }
#else
__device__ __forceinline__ int x16(int v) {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    return (int)((threadIdx.x & 16u) ? r[0] : r[1]);
}
__device__ __forceinline__ int x32(int v) {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return (int)((threadIdx.x & 32u) ? r[0] : r[1]);
}
#endif

/* ---- four view slots (nrReconNeighbors <= 4, the reference's default) */
#ifdef MI_TRANSPOSED
/*
 * The throughput layout with the lanes TRANSPOSED: lane = view slot * 16 + patch instead of patch * 4 + view slot.  The same
 * 16 patches x 4 view slots, the same arithmetic in the same order -- the sums over the view slots run (v0 + v1) + (v2 + v3)
 * in both, so the maps are the same bits -- but the 16 lanes of a DPP row now gather from ONE neighbour image, at the
 * positions of 16 patches that lie within a few pixels of each other (k_generate emits a wavefront's entries sub-tile by
 * sub-tile): consecutive lanes ask for records that are neighbours in memory, where the quad layout put four different
 * images side by side (no two consecutive lanes ever shared a cache line, and the texture path looked 64 lines up per
 * gather).  What the quad layout did with DPP quad permutes -- the exchanges across the view slots of a patch: two to nine
 * sums per pass, a few ballots per turn -- crosses DPP rows here: v_permlane16_swap / v_permlane32_swap (gfx950), or an LDS
 * permute (-DMI_T_SHFL).
 */
template <> struct Lay<1, 4> {
    static constexpr int LPV = 1, NV = 4, PATCHES = 16;
    static constexpr bool LAT = false;
    __device__ static __forceinline__ int vslot(int lane) { return lane >> 4; }
    __device__ static __forceinline__ int sub(int) { return 0; }
    __device__ static __forceinline__ int patch(int lane) { return lane & 15; }
    __device__ static __forceinline__ float view_sum(float v) { return v; }
    __device__ static __forceinline__ double view_sum(double v) { return v; }
    __device__ static __forceinline__ bool view_all(bool p) { return p; }
    __device__ static __forceinline__ float patch_sum(float v) {
        v = fadd_i(v, x16(__float_as_int(v)));              /* v0 + v1 | v2 + v3 */
        v = fadd_i(v, x32(__float_as_int(v)));              /* (v0 + v1) + (v2 + v3) */
        return v;
    }
    __device__ static __forceinline__ double patch_sum(double v) {
        v += dmov(v, x16);
        v += dmov(v, x32);
        return v;
    }
    __device__ static __forceinline__ unsigned long long patch_or(unsigned long long v) {
        v |= dmov_u(v, x16); v |= dmov_u(v, x32);
        return v;
    }
    __device__ static __forceinline__ float wave_sum(float v) { return v; }      /* unused in this layout */
    /* out[k] = v of view slot k of my patch: mine, and my partners' across 16, 32 and 48 lanes */
    __device__ static __forceinline__ void from_views(int v, int lane, int* out) {
        const int a = x16(v), b = x32(v), c = x32(a);
        const int s = lane >> 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = (k == s) ? v : (k == (s ^ 1)) ? a : (k == (s ^ 2)) ? b : c;
    }
    template <int S> __device__ static __forceinline__ int view_xor(int v) { return S == 0 ? x16(v) : x32(v); }
    /* bit k = predicate of view slot k of my patch */
    __device__ static __forceinline__ unsigned view_ballot(bool p, int lane) {
        const unsigned long long r = __ballot(p) >> (lane & 15);
        return (unsigned)((r & 1ull) | ((r >> 15) & 2ull) | ((r >> 30) & 4ull) | ((r >> 45) & 8ull));
    }
    __device__ static __forceinline__ unsigned rows_to_lane0(unsigned v) { return v; }
};
#else
template <> struct Lay<1, 4> {
    static constexpr int LPV = 1, NV = 4, PATCHES = 16;
    static constexpr bool LAT = false;
    __device__ static __forceinline__ int vslot(int lane) { return lane & 3; }
    __device__ static __forceinline__ int sub(int) { return 0; }
    __device__ static __forceinline__ int patch(int lane) { return lane >> 2; }
    __device__ static __forceinline__ float view_sum(float v) { return v; }
    __device__ static __forceinline__ double view_sum(double v) { return v; }
    __device__ static __forceinline__ bool view_all(bool p) { return p; }
    __device__ static __forceinline__ float patch_sum(float v) {
        v = fadd_i(v, dpp_xor1(__float_as_int(v)));
        v = fadd_i(v, dpp_xor2(__float_as_int(v)));
        return v;
    }
    __device__ static __forceinline__ double patch_sum(double v) {
        v += dmov(v, dpp_xor1);
        v += dmov(v, dpp_xor2);
        return v;
    }
    __device__ static __forceinline__ unsigned long long patch_or(unsigned long long v) {
        v |= dmov_u(v, dpp_xor1); v |= dmov_u(v, dpp_xor2);
        return v;
    }
    __device__ static __forceinline__ float wave_sum(float v) { return v; }      /* unused in this layout */
    /* out[k] = v of view slot k of my patch */
    __device__ static __forceinline__ void from_views(int v, int, int* out) {
        out[0] = dpp_bcast<0>(v); out[1] = dpp_bcast<1>(v); out[2] = dpp_bcast<2>(v); out[3] = dpp_bcast<3>(v);
    }
    /* butterfly partner exchange over the view slots: step 0, 1 (, 2) */
    template <int S> __device__ static __forceinline__ int view_xor(int v) { return S == 0 ? dpp_xor1(v) : dpp_xor2(v); }
    /* bit k = predicate of view slot k of my patch */
    __device__ static __forceinline__ unsigned view_ballot(bool p, int lane) {
        const unsigned long long b = __ballot(p);
        return (unsigned)(b >> (lane & ~3)) & 0xFu;
    }
    /* per-view counters of the wavefront's one patch summed into lane 0 (latency layouts only) */
    __device__ static __forceinline__ unsigned rows_to_lane0(unsigned v) { return v; }
};
#endif

template <> struct Lay<16, 4> {
    static constexpr int LPV = 16, NV = 4, PATCHES = 1;
    static constexpr bool LAT = true;
    __device__ static __forceinline__ int vslot(int lane) { return lane >> 4; }
    __device__ static __forceinline__ int sub(int lane) { return lane & 15; }
    /* LDS slot of the patch = the wavefront's index in its workgroup (k_tail runs four wavefronts per workgroup) */
    __device__ static __forceinline__ int patch(int) { return (int)(threadIdx.x >> 6); }
    __device__ static __forceinline__ float view_sum(float v) {      /* all 16 lanes of the row get the sum */
        v = fadd_i(v, dpp_xor1(__float_as_int(v)));
        v = fadd_i(v, dpp_xor2(__float_as_int(v)));
        v = fadd_i(v, dpp_half_mirror(__float_as_int(v)));
        v = fadd_i(v, dpp_mirror(__float_as_int(v)));
        return v;
    }
    __device__ static __forceinline__ double view_sum(double v) {
        v += dmov(v, dpp_xor1);
        v += dmov(v, dpp_xor2);
        v += dmov(v, dpp_half_mirror);
        v += dmov(v, dpp_mirror);
        return v;
    }
    __device__ static __forceinline__ bool view_all(bool p) {
        int v = p ? 1 : 0;
        v &= dpp_xor1(v); v &= dpp_xor2(v); v &= dpp_half_mirror(v); v &= dpp_mirror(v);
        return v != 0;
    }
    /* inputs are already uniform within each row: four readlanes instead of LDS permutes.  (-DMI_LAT_PERMLANE: the same sums,
     * (v0 + v1) + (v2 + v3), by two cross-row exchanges x16 / x32 -- v_permlane16/32_swap -- instead of the trip through the
     * scalar unit.  Measured, same lease: the front of a lone 20-view call 14.8-15.1 ms against 13.4 with the readlanes, the
     * front of a 400-view batch 2.09 against 1.85 ms per step: the swaps sit in the turn's dependent chain and are slower than
     * v_readlane + s_nop.) */
#ifndef MI_LAT_PERMLANE
    __device__ static __forceinline__ float patch_sum(float v) {
        const int i = __float_as_int(v);
        return (__int_as_float(__builtin_amdgcn_readlane(i, 0)) + __int_as_float(__builtin_amdgcn_readlane(i, 16)))
             + (__int_as_float(__builtin_amdgcn_readlane(i, 32)) + __int_as_float(__builtin_amdgcn_readlane(i, 48)));
    }
    __device__ static __forceinline__ double patch_sum(double v) {
        const int hi = __double2hiint(v), lo = __double2loint(v);
        const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
        const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
        const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
        const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
        return (r0 + r1) + (r2 + r3);
    }
#else
    __device__ static __forceinline__ float patch_sum(float v) {
        v = fadd_i(v, x16(__float_as_int(v)));
        v = fadd_i(v, x32(__float_as_int(v)));
        return v;
    }
    __device__ static __forceinline__ double patch_sum(double v) {
        v += dmov(v, x16);
        v += dmov(v, x32);
        return v;
    }
#endif
    /* sum over all 64 lanes (inputs arbitrary) */
    __device__ static __forceinline__ float wave_sum(float v) { return patch_sum(view_sum(v)); }
    __device__ static __forceinline__ unsigned long long patch_or(unsigned long long v) {
        int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
        lo |= __shfl_xor(lo, 16); lo |= __shfl_xor(lo, 32); hi |= __shfl_xor(hi, 16); hi |= __shfl_xor(hi, 32);
        return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
    }
    __device__ static __forceinline__ void from_views(int v, int, int* out) {
        out[0] = __builtin_amdgcn_readlane(v, 0); out[1] = __builtin_amdgcn_readlane(v, 16);
        out[2] = __builtin_amdgcn_readlane(v, 32); out[3] = __builtin_amdgcn_readlane(v, 48);
    }
    template <int S> __device__ static __forceinline__ int view_xor(int v) { return __shfl_xor(v, S == 0 ? 16 : 32); }
    __device__ static __forceinline__ unsigned view_ballot(bool p, int) {
        const unsigned long long b = __ballot(p);
        return (unsigned)((b & 1ull) | ((b >> 15) & 2ull) | ((b >> 30) & 4ull) | ((b >> 45) & 8ull));
    }
    __device__ static __forceinline__ unsigned rows_to_lane0(unsigned v) {
        return (unsigned)(__builtin_amdgcn_readlane((int)v, 0) + __builtin_amdgcn_readlane((int)v, 16)
                        + __builtin_amdgcn_readlane((int)v, 32) + __builtin_amdgcn_readlane((int)v, 48));
    }
};

/* ---- eight view slots (nrReconNeighbors 5..8, apps/dmrecon --local-neighbors): a patch is an OCTET of lanes in the
 * throughput layout (8 patches per wavefront), eight 8-lane half rows in the latency layout.  Sums run in the order
 * ((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7)) in both. */
template <> struct Lay<1, 8> {
    static constexpr int LPV = 1, NV = 8, PATCHES = 8;
    static constexpr bool LAT = false;
    __device__ static __forceinline__ int vslot(int lane) { return lane & 7; }
    __device__ static __forceinline__ int sub(int) { return 0; }
    __device__ static __forceinline__ int patch(int lane) { return lane >> 3; }
    __device__ static __forceinline__ float view_sum(float v) { return v; }
    __device__ static __forceinline__ double view_sum(double v) { return v; }
    __device__ static __forceinline__ bool view_all(bool p) { return p; }
    __device__ static __forceinline__ float patch_sum(float v) {
        v = fadd_i(v, dpp_xor1(__float_as_int(v)));
        v = fadd_i(v, dpp_xor2(__float_as_int(v)));
        v = fadd_i(v, dpp_half_mirror(__float_as_int(v)));         /* the quads of an octet are uniform by now */
        return v;
    }
    __device__ static __forceinline__ double patch_sum(double v) {
        v += dmov(v, dpp_xor1);
        v += dmov(v, dpp_xor2);
        v += dmov(v, dpp_half_mirror);
        return v;
    }
    __device__ static __forceinline__ unsigned long long patch_or(unsigned long long v) {
        v |= dmov_u(v, dpp_xor1); v |= dmov_u(v, dpp_xor2); v |= dmov_u(v, dpp_half_mirror);
        return v;
    }
    __device__ static __forceinline__ float wave_sum(float v) { return v; }      /* unused in this layout */
    __device__ static __forceinline__ void from_views(int v, int lane, int* out) {
        /* b[k]: lane k of my own quad; m[k]: lane k of the octet's other quad (half_mirror of a quad-uniform value) */
        const int b0 = dpp_bcast<0>(v), b1 = dpp_bcast<1>(v), b2 = dpp_bcast<2>(v), b3 = dpp_bcast<3>(v);
        const int m0 = dpp_half_mirror(b0), m1 = dpp_half_mirror(b1), m2 = dpp_half_mirror(b2), m3 = dpp_half_mirror(b3);
        const bool upper = (lane & 4) != 0;
        out[0] = upper ? m0 : b0; out[1] = upper ? m1 : b1; out[2] = upper ? m2 : b2; out[3] = upper ? m3 : b3;
        out[4] = upper ? b0 : m0; out[5] = upper ? b1 : m1; out[6] = upper ? b2 : m2; out[7] = upper ? b3 : m3;
    }
    template <int S> __device__ static __forceinline__ int view_xor(int v) { return S == 0 ? dpp_xor1(v) : S == 1 ? dpp_xor2(v) : dpp_half_mirror(v); }
    __device__ static __forceinline__ unsigned view_ballot(bool p, int lane) {
        const unsigned long long b = __ballot(p);
        return (unsigned)(b >> (lane & ~7)) & 0xFFu;
    }
    __device__ static __forceinline__ unsigned rows_to_lane0(unsigned v) { return v; }
};

template <> struct Lay<8, 8> {
    static constexpr int LPV = 8, NV = 8, PATCHES = 1;
    static constexpr bool LAT = true;
    __device__ static __forceinline__ int vslot(int lane) { return lane >> 3; }
    __device__ static __forceinline__ int sub(int lane) { return lane & 7; }
    __device__ static __forceinline__ int patch(int) { return (int)(threadIdx.x >> 6); }
    __device__ static __forceinline__ float view_sum(float v) {      /* all 8 lanes of the half row get the sum */
        v = fadd_i(v, dpp_xor1(__float_as_int(v)));
        v = fadd_i(v, dpp_xor2(__float_as_int(v)));
        v = fadd_i(v, dpp_half_mirror(__float_as_int(v)));
        return v;
    }
    __device__ static __forceinline__ double view_sum(double v) {
        v += dmov(v, dpp_xor1);
        v += dmov(v, dpp_xor2);
        v += dmov(v, dpp_half_mirror);
        return v;
    }
    __device__ static __forceinline__ bool view_all(bool p) {
        int v = p ? 1 : 0;
        v &= dpp_xor1(v); v &= dpp_xor2(v); v &= dpp_half_mirror(v);
        return v != 0;
    }
    __device__ static __forceinline__ float rl(int i, int lane) { return __int_as_float(__builtin_amdgcn_readlane(i, lane)); }
    __device__ static __forceinline__ float patch_sum(float v) {
        const int i = __float_as_int(v);
        return ((rl(i, 0) + rl(i, 8)) + (rl(i, 16) + rl(i, 24))) + ((rl(i, 32) + rl(i, 40)) + (rl(i, 48) + rl(i, 56)));
    }
    __device__ static __forceinline__ double rld(int hi, int lo, int lane) {
        return __hiloint2double(__builtin_amdgcn_readlane(hi, lane), __builtin_amdgcn_readlane(lo, lane));
    }
    __device__ static __forceinline__ double patch_sum(double v) {
        const int hi = __double2hiint(v), lo = __double2loint(v);
        return ((rld(hi, lo, 0) + rld(hi, lo, 8)) + (rld(hi, lo, 16) + rld(hi, lo, 24)))
             + ((rld(hi, lo, 32) + rld(hi, lo, 40)) + (rld(hi, lo, 48) + rld(hi, lo, 56)));
    }
    __device__ static __forceinline__ float wave_sum(float v) { return patch_sum(view_sum(v)); }
    __device__ static __forceinline__ unsigned long long patch_or(unsigned long long v) {
        int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
        lo |= __shfl_xor(lo, 8); lo |= __shfl_xor(lo, 16); lo |= __shfl_xor(lo, 32);
        hi |= __shfl_xor(hi, 8); hi |= __shfl_xor(hi, 16); hi |= __shfl_xor(hi, 32);
        return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
    }
    __device__ static __forceinline__ void from_views(int v, int, int* out) {
#pragma unroll
        for (int k = 0; k < 8; ++k) out[k] = __builtin_amdgcn_readlane(v, 8 * k);
    }
    template <int S> __device__ static __forceinline__ int view_xor(int v) { return __shfl_xor(v, S == 0 ? 8 : S == 1 ? 16 : 32); }
    __device__ static __forceinline__ unsigned view_ballot(bool p, int) {
        const unsigned long long b = __ballot(p);
        unsigned r = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) r |= (unsigned)((b >> (8 * k)) & 1ull) << k;
        return r;
    }
    __device__ static __forceinline__ unsigned rows_to_lane0(unsigned v) {
        unsigned r = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) r += (unsigned)__builtin_amdgcn_readlane((int)v, 8 * k);
        return r;
    }
};

/* ---- sixteen view slots (nrReconNeighbors 9..16): a patch is a DPP ROW of lanes, four patches per wavefront, in the throughput
 * layout -- the only layout such a patch has: its views never hand over to the fused rounds (the host keeps every round of
 * such a call host-visible), there is no FAST and no speculative kernel for it.  A rarely used setting (the reference's
 * default is 4; apps/dmrecon --local-neighbors): built for the reference's semantics, not for speed.  Sums run in the order
 * (((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7))) + (the same of v8..v15). */
template <> struct Lay<1, 16> {
    static constexpr int LPV = 1, NV = 16, PATCHES = 4;
    static constexpr bool LAT = false;
    __device__ static __forceinline__ int vslot(int lane) { return lane & 15; }
    __device__ static __forceinline__ int sub(int) { return 0; }
    __device__ static __forceinline__ int patch(int lane) { return lane >> 4; }
    __device__ static __forceinline__ float view_sum(float v) { return v; }
    __device__ static __forceinline__ double view_sum(double v) { return v; }
    __device__ static __forceinline__ bool view_all(bool p) { return p; }
    __device__ static __forceinline__ float patch_sum(float v) {
        v = fadd_i(v, dpp_xor1(__float_as_int(v)));
        v = fadd_i(v, dpp_xor2(__float_as_int(v)));
        v = fadd_i(v, dpp_half_mirror(__float_as_int(v)));         /* (the quads of an octet are uniform by now) */
        v = fadd_i(v, dpp_mirror(__float_as_int(v)));              /* (... and the octets of the row) */
        return v;
    }
    __device__ static __forceinline__ double patch_sum(double v) {
        v += dmov(v, dpp_xor1);
        v += dmov(v, dpp_xor2);
        v += dmov(v, dpp_half_mirror);
        v += dmov(v, dpp_mirror);
        return v;
    }
    __device__ static __forceinline__ unsigned long long patch_or(unsigned long long v) {
        v |= dmov_u(v, dpp_xor1); v |= dmov_u(v, dpp_xor2); v |= dmov_u(v, dpp_half_mirror); v |= dmov_u(v, dpp_mirror);
        return v;
    }
    __device__ static __forceinline__ float wave_sum(float v) { return v; }      /* unused in this layout */
    /* out[k] = v of view slot k of my patch (an LDS permute per slot: this layout is not the fast path) */
    __device__ static __forceinline__ void from_views(int v, int lane, int* out) {
#pragma unroll
        for (int k = 0; k < 16; ++k) out[k] = __shfl(v, (lane & ~15) | k);
    }
    /* butterfly partner exchange over the view slots: steps 0..3 (the mirrors stand for xor 4 / xor 8 on values that are
     * uniform over the quads / octets after the steps before them) */
    template <int S> __device__ static __forceinline__ int view_xor(int v) {
        return S == 0 ? dpp_xor1(v) : S == 1 ? dpp_xor2(v) : S == 2 ? dpp_half_mirror(v) : dpp_mirror(v);
    }
    __device__ static __forceinline__ unsigned view_ballot(bool p, int lane) {
        const unsigned long long b = __ballot(p);
        return (unsigned)(b >> (lane & ~15)) & 0xFFFFu;
    }
    __device__ static __forceinline__ unsigned rows_to_lane0(unsigned v) { return v; }
};

/* the latency layout that goes with a number of view slots */
template <int NV> struct LatLay;
template <> struct LatLay<4> { typedef Lay<16, 4> type; };
template <> struct LatLay<8> { typedef Lay<8, 8> type; };

/* ------------------------------------------------------------------------- */

#ifdef MI_ACTIVITY
__shared__ unsigned g_act[2];
#endif

/* LocalViewSelection::available over the job's global list: bit g % 64 of word g / 64 (MI_MAX_GLOBAL bits) */
struct Avail {
    unsigned long long w[MI_AVAIL_WORDS];
    __device__ __forceinline__ bool test(int g) const {
        unsigned long long v = w[0];
#pragma unroll
        for (int k = 1; k < MI_AVAIL_WORDS; ++k) v = (g >> 6) == k ? w[k] : v;
        return ((v >> (g & 63)) & 1ull) != 0;
    }
    __device__ __forceinline__ void clear(int g) {
        const unsigned long long m = ~(1ull << (g & 63));
#pragma unroll
        for (int k = 0; k < MI_AVAIL_WORDS; ++k) w[k] &= (g >> 6) == k ? m : ~0ull;
    }
    __device__ __forceinline__ void set(int g) {
        const unsigned long long m = 1ull << (g & 63);
#pragma unroll
        for (int k = 0; k < MI_AVAIL_WORDS; ++k) w[k] |= (g >> 6) == k ? m : 0ull;
    }
    /* the first n views available */
    __device__ __forceinline__ void first(int n) {
#pragma unroll
        for (int k = 0; k < MI_AVAIL_WORDS; ++k) {
            const int r = n - 64 * k;
            w[k] = r >= 64 ? ~0ull : (r <= 0 ? 0ull : ((1ull << r) - 1ull));
        }
    }
};

struct PatchState {
    /* uniform over the lanes of a patch */
    const DevJob* job;
    int x, y;
    float depth, dzI, dzJ;
    float xbar0, xbar1, xbar2;   /* PatchSampler::meanX */
    float sqrDevX, mmean;        /* PatchSampler::sqrDevX, masterMeanCol */
    float mfp;                   /* footPrintScaled(centre point) at the current state */
    float inrm_c;                /* geo[MI_MID]: 1 / |K_s^-1 (x + .5, y + .5, 1)| of the centre pixel */
    float jinv0;                 /* invproj[0] of the reference level */
    Avail avail;                 /* LocalViewSelection::available over global indices -- once a view selection has asked for it
                                  * (avail_ready); until then the words hold the propagated view set it will be made from */
    bool avail_ready;
    /* per view slot */
    int sel;                     /* my view: index into job->global_ids, or -1 */
    float cs0, cs1, cs2;         /* PatchOptimization::colorScale[my view] */
    float ncc;                   /* getFastNCC(my view) at the current state */
    /* counters (flushed at kernel end) */
    unsigned n_eval, n_pass;
    DevCounters* counters;       /* rare-event diagnostics are added directly (one atomic per event) */
};

/*
 * Geometry of a sampling pass.  The reference builds sample (di, dj) of a patch as the 3-D point
 *     P = C + t r,  r = R^T K_s^-1 (x + di + .5, y + dj + .5, 1) / |.|,  t = depth + di dzI + dj dzJ
 * (computePatchPoints, patch_sampler.cc:273-295; view rays single_view.cc:106-114) and projects it with the neighbour's
 * K [R|t] (worldToScreen, single_view.h:187-195).  C and the pixel grid are the same for all samples, so
 *     K_n (R_n P + t_n) = s_C + (t g) (A + di B + dj D)
 * with s_C = K_n (R_n C + t_n) (the reference camera centre seen from the neighbour), H = K_n R_n R^T K_s^-1 (a per
 * (reference view, neighbour) matrix, DevJobView::H), A = H (x + .5, y + .5, 1), B = H e_x, D = H e_y and
 * g = 1 / |K_s^-1 (x + di + .5, y + dj + .5, 1)| (per patch and sample, in LDS): three FMAs for the direction, one
 * multiply for the scale, three FMAs for the point -- instead of a 3 x 4 matrix product per sample -- and the
 * ray-advanced twin of fastColAndDeriv (patch_sampler.cc:101-113) is s + (step g) (A + di B + dj D).
 */
struct NView {                   /* my neighbour view at the selected mip level (rows 0, 1 carry the level's K) */
    float sx, sy, sz;            /* s_C */
    float ax, ay, az;            /* A: the centre pixel */
    float bx, by, bz;            /* B: one pixel to the right */
    float dx, dy, dz;            /* D: one pixel down */
    int w, h;
    const uint32_t* img;         /* footprint elements of the level (DevView::quad) */
#ifdef MI_LDS_WINDOW
    const uint32_t* raw;         /* the level's plain RGBA8 texels (DevView::img): what a wavefront's LDS windows are filled from */
#endif
};

/* fold the level's calibration into rows 0, 1: x' = ax.x + cx.z, y' = ay.y + cy.z */
__device__ __forceinline__ void premultiply(NView& nv, float ax, float ay, float cx, float cy) {
    nv.sx = ax * nv.sx + cx * nv.sz; nv.ax = ax * nv.ax + cx * nv.az; nv.bx = ax * nv.bx + cx * nv.bz; nv.dx = ax * nv.dx + cx * nv.dz;
    nv.sy = ay * nv.sy + cy * nv.sz; nv.ay = ay * nv.ay + cy * nv.az; nv.by = ay * nv.by + cy * nv.bz; nv.dy = ay * nv.dy + cy * nv.dz;
}
/* rows 0, 1 (no calibration yet) and row 2 of the pass geometry of view J for the patch at pixel (x, y) */
__device__ __forceinline__ void view_rows01(NView& nv, const DevJobView& J, float fx, float fy) {
    nv.sx = J.sc[0]; nv.sy = J.sc[1];
    nv.bx = J.H[0]; nv.dx = J.H[1]; nv.ax = J.H[0] * fx + J.H[1] * fy + J.H[2];
    nv.by = J.H[3]; nv.dy = J.H[4]; nv.ay = J.H[3] * fx + J.H[4] * fy + J.H[5];
}
__device__ __forceinline__ void view_row2(NView& nv, const DevJobView& J, float fx, float fy) {
    nv.sz = J.sc[2];
    nv.bz = J.H[6]; nv.dz = J.H[7]; nv.az = J.H[6] * fx + J.H[7] * fy + J.H[8];
}

/* mip level rule of patch_sampler.cc:72-91 / :353-373 from the view-space depth z of the centre patch point.
 * Returns -1 if nfp <= 0. */
__device__ __forceinline__ int mip_level(float z, float inv0, float mfp, int maxl) {
    const float nfp = z * inv0;                  /* SingleView::footPrint */
    if (!(nfp > 0.f)) return -1;
    const float ratio = nfp / mfp;               /* (an IEEE division: the synthetic scenes sit exactly on the rule's threshold, Q4) */
    /* the reference doubles the ratio until it reaches 0.5, at most MI_MAX_LEVELS times (patch_sampler.cc:85-91): the
     * doublings are exact, so their number is the negated exponent of ratio = f 2^e, f in [0.5, 1) -- one v_frexp_exp_i32_f32
     * instead of a loop with a branch per level (a dependent chain of ~40 cycles per level in the latency layout) */
    int mm = 0;
    if (ratio < 0.5f) {
        const int e = -__builtin_amdgcn_frexp_expf(ratio);
        mm = (ratio > 0.f && e < MI_MAX_LEVELS) ? e : MI_MAX_LEVELS;
    }
    return mm > maxl ? maxl : mm;                /* clampLevel with minLevel 0 (dmrecon.cc:240) */
}

/* One-shot set-up of a view (view selection candidates, parity hook): two dependent loads (the job's record of the
 * view, then the level). */
__device__ __forceinline__ bool setup_view(const DevView* __restrict__ views, const DevJobView& J, const PatchState& ps,
                                           NView& nv, int& level) {
    const float fx = (float)ps.x + 0.5f, fy = (float)ps.y + 0.5f;
    view_row2(nv, J, fx, fy);
    const float z = nv.sz + (ps.depth * ps.inrm_c) * nv.az;          /* (worldToCam . centre point).z */
    const int mm = mip_level(z, J.inv0, ps.mfp, J.maxl);
    if (mm < 0) return false;
    level = mm;
    view_rows01(nv, J, fx, fy);
    const DevView* V = views + J.view;
    const DevLevel& L = V->lv[mm];
    premultiply(nv, L.ax, L.ay, L.cx, L.cy);
    nv.w = L.w; nv.h = L.h;
    nv.img = V->quad + MI_QUAD_WORDS * (size_t)L.tex_off;      /* footprint elements of this level */
#ifdef MI_LDS_WINDOW
    nv.raw = V->img + L.tex_off;
#endif
    return true;
}

/*
 * A view slot's selected view, kept across the passes of a patch (latency layout): the per-pass set-up of the
 * reference (worldToScreen matrices, mip level, patch_sampler.cc:72-91) costs three dependent memory accesses
 * (job record -> DevView -> DevLevel) when done from scratch, which is most of a pass's latency in the tail.
 * Cached here, a pass re-evaluates the level rule in registers and touches memory only when the view or its
 * level changed.
 */
struct ViewC {
    int sel;                     /* PatchState::sel this cache belongs to (-2 = empty) */
    int lvl;                     /* mip level rows 0, 1 of nv belong to (-1 = none) */
    const DevJobView* V;
    float inv0; int maxl;
    NView nv;                    /* row 2 valid once sel is set; rows 0, 1, w, h, img per level */
};

__device__ __forceinline__ void viewc_reset(ViewC& vc) {
    vc.sel = -2; vc.lvl = -1; vc.V = nullptr; vc.inv0 = 0.f; vc.maxl = 0;
    vc.nv.w = 0; vc.nv.h = 0; vc.nv.img = nullptr;
#ifdef MI_LDS_WINDOW
    vc.nv.raw = nullptr;
#endif
    vc.nv.sx = vc.nv.sy = vc.nv.sz = vc.nv.ax = vc.nv.ay = vc.nv.az = 0.f;
    vc.nv.bx = vc.nv.by = vc.nv.bz = vc.nv.dx = vc.nv.dy = vc.nv.dz = 0.f;
}


struct ColorSums {               /* shifted one-pass sums of the colours of one view at one state */
    float s0, s1, s2;            /* shift (any value near the mean colour; only conditions the sums) */
    float a0, a1, a2;            /* sum (n - s) */
    float aa0, aa1, aa2;         /* sum (n - s)^2 */
    float ba0, ba1, ba2;         /* sum (m - xbar)(n - s) */
};

enum { PASS_COLOR = 0, PASS_DEPTH = 1, PASS_NORMAL = 2, PASS_DUMP = 3, PASS_DEPTH_FIXED = 4,
       PASS_DEPTH_FIXED_NC = 5 /* PASS_DEPTH_FIXED without the colour sums: the passes of iterations 1..3, whose NCC nobody asks for */ };

template <int LPV> struct NormalAcc { typedef float type; };     /* 2 terms per lane: float partial sums */
template <> struct NormalAcc<1> { typedef double type; };          /* 25 terms per lane: accumulate in double as the reference */

/* Gauss-Newton sums of one view at one state.
 * Depth-only form, per colour channel c and independent of later colour-scale changes:
 *   dr_c = sum d_c (m_c - cs0_c n_c)   (cs0 = the colour scale at pass time)
 *   dn_c = sum d_c n_c,  dd_c = sum d_c^2
 * so that for any colour scale cs:  num = sum_c cs_c (dr_c - (cs_c - cs0_c) dn_c),  den = sum_c cs_c^2 dd_c
 * which is optimizeDepthOnly's  sum (cs d)(m - cs n) / sum (cs d)^2  (patch_optimization.cc:283-290). */
struct GNSums {
    float num, den;                                    /* PASS_DEPTH_FIXED: optimizeDepthOnly's sums with the current colour scale */
    float dr0, dr1, dr2, dn0, dn1, dn2, dd0, dd1, dd2;
    float c00, c01, c02;                               /* cs0 */
    double A00, A01, A02, A11, A12, A22, B0, B1, B2;   /* optimizeDepthAndNormal (:312-343), colour scale baked in */
};

#ifdef MI_LDS_WINDOW
struct Win { int x0, y0; unsigned limx, limy; const uint32_t* tile; };
/* Builds the LDS windows of the pass the calling lanes are about to run (see g_win).  Called by an arbitrary subset of a
 * wavefront's lanes; returns the same value in all of them: true = every caller's samples can be read from its slot's tile. */
__device__ __forceinline__ bool window_build(const PatchState& ps, const NView& nv, const float* __restrict__ geo, int lane, Win& W) {
    const int slot = lane & 3;
    const unsigned long long act = __ballot(true);
    const float wlim = __uint_as_float(__float_as_uint((float)(nv.w - 1)) - 1u), hlim = __uint_as_float(__float_as_uint((float)(nv.h - 1)) - 1u);
    constexpr float kLo = 1.17549435e-38f;
    float umin = 3.0e38f, vmin = 3.0e38f, umax = 0.f, vmax = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int di = (c & 1) ? MI_HALF : -MI_HALF, dj = (c & 2) ? MI_HALF : -MI_HALF;
        const int i = (dj + MI_HALF) * MI_FW + (di + MI_HALF);
        const float fi = (float)di, fj = (float)dj;
        const float lam = (ps.depth + fi * ps.dzI + fj * ps.dzJ) * geo[i];
        const float vx = nv.ax + fi * nv.bx + fj * nv.dx, vy = nv.ay + fi * nv.by + fj * nv.dy, vz = nv.az + fi * nv.bz + fj * nv.dz;
        const float iz = fast_rcp(nv.sz + lam * vz);
        const float u = __builtin_amdgcn_fmed3f((nv.sx + lam * vx) * iz - 0.5f, kLo, wlim), v = __builtin_amdgcn_fmed3f((nv.sy + lam * vy) * iz - 0.5f, kLo, hlim);
        umin = fminf(umin, u); umax = fmaxf(umax, u); vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
    }
    /* texel indices, inclusive, one texel of margin (the box of the corners is the box of the window only as far as the patch
     * projects as a plane; the unit-ray scales bend it by a fraction of a texel); x + 1 / y + 1 = the footprint's far side */
    int x0 = (int)umin - 1, y0 = (int)vmin - 1, x1 = (int)umax + 2, y1 = (int)vmax + 2;
    x0 = x0 < 0 ? 0 : x0; y0 = y0 < 0 ? 0 : y0; x1 = x1 > nv.w - 1 ? nv.w - 1 : x1; y1 = y1 > nv.h - 1 ? nv.h - 1 : y1;
    /* the slot's box and image over the calling lanes */
    const int4 init = make_int4(0x7FFFFFFF, 0x7FFFFFFF, (int)0x80000000, (int)0x80000000);
    *reinterpret_cast<int4*>(g_wbox[slot]) = init;
    const unsigned rlo = (unsigned)(uintptr_t)nv.raw, rhi = (unsigned)((uintptr_t)nv.raw >> 32);
    *reinterpret_cast<uint4*>(g_wimg[slot]) = make_uint4(rlo, rhi, (unsigned)nv.w, 0u);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    atomicMin(&g_wbox[slot][0], x0); atomicMin(&g_wbox[slot][1], y0); atomicMax(&g_wbox[slot][2], x1); atomicMax(&g_wbox[slot][3], y1);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const int4 box = *reinterpret_cast<const int4*>(g_wbox[slot]);
    const uint4 im = *reinterpret_cast<const uint4*>(g_wimg[slot]);
    const int bw = box.z - box.x + 1, bh = box.w - box.y + 1;
    const bool other = im.x != rlo || im.y != rhi || im.z != (unsigned)nv.w;
    const bool large = bw > MI_WIN_TP || bh > MI_WIN_TH;
    const unsigned long long m_other = __ballot(other), m_large = __ballot(large);
    if ((m_other | m_large) != 0ull) {
        /* (why not, per wavefront-pass: g_wstat[2] the slots' patches sample different images or levels, [3] a box does not fit) */
        if ((int)lane == __ffsll((long long)act) - 1) atomicAdd(&g_wstat[m_other ? 2 : 3], 1u);
        return false;
    }
    /* fill: the boxes' rows in 16-byte pieces, eight pieces per tile row, dealt over the calling lanes (a row's last piece may reach
     * past the box: it stays inside the pitch, and inside the view's allocation -- the footprint elements follow the planes) */
    const unsigned nact = (unsigned)__popcll(act);
    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)act, 0u));
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
        if (!(act & (0x1111111111111111ull << sl))) continue;               /* no caller has a view in this slot */
        const int4 b = *reinterpret_cast<const int4*>(g_wbox[sl]);
        const uint4 m = *reinterpret_cast<const uint4*>(g_wimg[sl]);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(((uintptr_t)m.y << 32) | (uintptr_t)m.x);
        const int sw = (int)m.z, sbw = b.z - b.x + 1, sbh = b.w - b.y + 1;
        for (unsigned idx = rank; idx < (unsigned)sbh * (MI_WIN_TP / 4u); idx += nact) {
            const unsigned r = idx / (MI_WIN_TP / 4u), c4 = (idx - r * (MI_WIN_TP / 4u)) * 4u;
            if ((int)c4 < sbw) {
                const u32x4 t = *(gtex4u_t)(src + (size_t)(b.y + (int)r) * (size_t)sw + (size_t)(b.x + (int)c4));
                *reinterpret_cast<u32x4*>(&g_win[sl][r * MI_WIN_TP + c4]) = t;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    W.x0 = box.x; W.y0 = box.y; W.limx = (unsigned)(bw - 2); W.limy = (unsigned)(bh - 2);
    W.tile = g_win[slot];
    return true;
}
#endif

/*
 * One pass over the 25 samples of my view (split over the L::LPV lanes of my view slot) at the current
 * patch state.  It always yields the colour sums that getFastNCC / computeColorScale need
 * (patch_sampler.cc:347-393,135-163; patch_optimization.cc:81-111), and in addition
 *   PASS_DEPTH  the colour-scale independent sums of optimizeDepthOnly (used when computeColorScale may
 *               still change the scale between this pass and the step: ctor, after a normal step),
 *   PASS_DEPTH_FIXED  optimizeDepthOnly's numerator / denominator directly (colour scale fixed until the step),
 *   PASS_DEPTH_FIXED_NC  the same WITHOUT the colour sums: the reference first asks for NCCs when its main loop starts
 *               (patch_optimization.cc:184-192) -- the passes that lead to the depth-only steps of iterations 1, 2, 3 feed no NCC,
 *               no colour scale and no convergence test (12 of a sample's ~107 VALU instructions, three of a patch's six passes),
 *   PASS_NORMAL the normal equations of optimizeDepthAndNormal (with the current colour scale),
 *   PASS_DUMP   the raw samples (parity hook, L::LPV = 1).
 * The reference samples the same texels twice per Gauss-Newton iteration -- computeNeighColorSamples
 * on the new state, then fastColAndDeriv on that same state in the next iteration (patch_sampler.cc:64-133,
 * mvs_tools.cc:97-145); one fused pass here gathers them once.
 * Returns PatchSampler::success[v]; sums are complete (reduced over the view slot) on return.
 */
template <int MODE, class L, bool WIN = false>
__device__ __forceinline__ bool sample_pass(const PatchState& ps, const NView& nv, const float* __restrict__ s_lut,
                                            const float* __restrict__ geo, const float* __restrict__ mcol,
                                            ColorSums& cs_out, GNSums& gn, float* dump_col, float* dump_der, int sub) {
    float step = 0.f, dnorm = 0.f;
    bool ok = true;
    if (MODE != PASS_COLOR) {
        /* derivative step size (patch_sampler.cc:93-100): the centre point advanced by its unit ray against
         * patchPoints[12] -- hard-coded there; the centre itself only for filter width 5 (MI_STEP_SAMPLE) */
        const float lc = ps.depth * ps.inrm_c;
        float l0 = lc, vx = nv.ax, vy = nv.ay, vz = nv.az;
        if (MI_STEP_SAMPLE != MI_MID) {
            constexpr int sj = MI_STEP_SAMPLE / MI_FW - MI_HALF, si = MI_STEP_SAMPLE - (MI_STEP_SAMPLE / MI_FW) * MI_FW - MI_HALF;
            l0 = (ps.depth + (float)si * ps.dzI + (float)sj * ps.dzJ) * geo[MI_STEP_SAMPLE];
            vx = nv.ax + (float)si * nv.bx + (float)sj * nv.dx; vy = nv.ay + (float)si * nv.by + (float)sj * nv.dy;
            vz = nv.az + (float)si * nv.bz + (float)sj * nv.dz;
        }
        const float iz0 = fast_rcp(nv.sz + l0 * vz);
        const float u0 = (nv.sx + l0 * vx) * iz0, v0 = (nv.sy + l0 * vy) * iz0;
        const float l1 = lc + ps.inrm_c;
        const float iz1 = fast_rcp(nv.sz + l1 * nv.az);
        const float u1 = (nv.sx + l1 * nv.ax) * iz1, v1 = (nv.sy + l1 * nv.ay) * iz1;
        const float du = u1 - u0, dv = v1 - v0;
        dnorm = fast_sqrt(du * du + dv * dv);          /* deriv /= stepSize  ==  deriv * dnorm */
        ok = dnorm > 0.f;
        step = ok ? fast_rcp(dnorm) : 0.f;
    }
    TSTAMP(50);                                        /* (pass set-up done: the derivative step) */
    /* The strict interior test 0 < u < w - 1, 0 < v < h - 1 (patch_sampler.cc:116-119, :386-389) and the memory-safe clamp in one:
     * the coordinates are clamped into [FLT_MIN, the float below w - 1] -- a sample inside the image is not moved by that, one
     * outside (or on the edge, or NaN: v_med3 then returns the lower bound) is --, so "the clamp changed nothing" IS the test:
     * one v_med3 and one v_cmp_eq per coordinate instead of one v_med3 and two compares.  (A coordinate in (0, 1.2e-38) would
     * be counted as outside.) */
    const float wlim = __uint_as_float(__float_as_uint((float)(nv.w - 1)) - 1u), hlim = __uint_as_float(__float_as_uint((float)(nv.h - 1)) - 1u);
    constexpr float kLo = 1.17549435e-38f;
    ColorSums S;
    S.s0 = ps.xbar0 * ps.mmean; S.s1 = ps.xbar1 * ps.mmean; S.s2 = ps.xbar2 * ps.mmean;
    S.a0 = S.a1 = S.a2 = 0.f; S.aa0 = S.aa1 = S.aa2 = 0.f; S.ba0 = S.ba1 = S.ba2 = 0.f;
    /* Float accumulators are PAIRS: the latency layouts run the samples of a lane two at a time as packed f32 (v_pk_fma_f32 /
     * v_pk_mul_f32 / v_pk_add_f32: two samples' arithmetic per instruction -- a third fewer instructions in a pass whose
     * length is a dependent chain; front of a lone call 13.3-13.6 -> 12.7 ms) -- half x takes the first sample of a pair and
     * a sample handled alone, half y the second; the halves are added after the loop.  The throughput layout handles its 25
     * samples one by one (below) and only ever touches half x. */
    f2 Pa0 = sp2(0.f), Pa1 = sp2(0.f), Pa2 = sp2(0.f), Paa0 = sp2(0.f), Paa1 = sp2(0.f), Paa2 = sp2(0.f);
    f2 Pba0 = sp2(0.f), Pba1 = sp2(0.f), Pba2 = sp2(0.f);
    f2 Pdr0 = sp2(0.f), Pdr1 = sp2(0.f), Pdr2 = sp2(0.f), Pdn0 = sp2(0.f), Pdn1 = sp2(0.f), Pdn2 = sp2(0.f);
    f2 Pdd0 = sp2(0.f), Pdd1 = sp2(0.f), Pdd2 = sp2(0.f);
    f2 Pnum = sp2(0.f), Pden = sp2(0.f);
    typedef typename NormalAcc<L::LPV>::type acc_t;
    /* the normal equations: double in the throughput layout (25 terms per lane, as the reference accumulates), there the two
     * halves of a pair are added one after the other; float pairs in the latency layouts (2-4 terms per lane) */
    acc_t A00 = 0, A01 = 0, A02 = 0, A11 = 0, A12 = 0, A22 = 0, B0 = 0, B1 = 0, B2 = 0;
    f2 PA00 = sp2(0.f), PA01 = sp2(0.f), PA02 = sp2(0.f), PA11 = sp2(0.f), PA12 = sp2(0.f), PA22 = sp2(0.f);
    f2 PB0 = sp2(0.f), PB1 = sp2(0.f), PB2 = sp2(0.f);
    /* per-channel colour sums are only needed when computeColorScale may follow this pass */
    constexpr bool PER_CHANNEL = (MODE == PASS_COLOR || MODE == PASS_DEPTH);
    constexpr bool WANT_S = MODE != PASS_DEPTH_FIXED_NC;                         /* the colour sums (NCC, colour scale) */
    constexpr bool FIXED = MODE == PASS_DEPTH_FIXED || MODE == PASS_DEPTH_FIXED_NC;

    constexpr int NITER = (MI_NS + L::LPV - 1) / L::LPV;
    const f2 Ss0 = sp2(S.s0), Ss1 = sp2(S.s1), Ss2 = sp2(S.s2);
    const f2 Cs0 = sp2(ps.cs0), Cs1 = sp2(ps.cs1), Cs2 = sp2(ps.cs2);
    /* A sample is handled in two steps so that texel gathers can be in flight while other samples are consumed:
     * geom() = geometry + the footprint gather, consume() = table look-ups, interpolation and the sums.
     *   latency layout (one wavefront per patch, nothing else to hide a gather behind): all samples of a lane are
     *   fetched before the first is consumed -- one exposed memory latency per pass instead of two;
     *   throughput layout: a whole row of the 5 x 5 window per gather round (see below).
     * geom2() / consume2() are the same for two samples at once, in packed arithmetic. */
#ifdef MI_EMU_LIN48
    struct Pre { int i; bool live; float fx, fy, gu, gv; u32x4 t, e1, e2; };
    struct Pre2 { int i0, i1; f2 wgt, fx, fy, gu, gv; u32x4 tA, tB, eA1, eA2, eB1, eB2; };
#else
    struct Pre { int i; bool live; float fx, fy, gu, gv; u32x4 t; };
    struct Pre2 { int i0, i1; f2 wgt, fx, fy, gu, gv; u32x4 tA, tB; };
#endif
    auto record = [&](float uc, float vc) -> u32x4 {
        /* one aligned 16-byte gather = the sample's 2 x 2 texel footprint (DevView::quad).  The L1 processes a
         * gather lane by lane when the lanes' addresses do not form one contiguous run, and that access rate is what
         * bounds the throughput layout: two 8-byte row gathers cost 2.45 L1 accesses per lane and sample, this 1.
         * The record index top * w + left as a 24-bit multiply-add in 32 bits (rows and widths are below 2^24, a level
         * below 2^32 texels) instead of a 64-bit multiply-add (6.3 issue cycles against 5.6, and no sign extension);
         * the texel indices by truncation (the coordinates are not negative). */
#ifdef MI_TILED_QUADS
        /* experiment: the records block-linear, 8 x 8 per tile (k_quadify), so that records that are neighbours in the image in
         * BOTH directions are neighbours in memory -- six more VALU instructions per sample in an instruction-bound kernel */
        const unsigned ux = (unsigned)uc, uy = (unsigned)vc;
        const unsigned rec = ((__umul24(uy >> 3, (unsigned)(nv.w + 7) >> 3) + (ux >> 3)) << 6) | ((uy & 7u) << 3) | (ux & 7u);
#else
        const unsigned rec = __umul24((unsigned)vc, (unsigned)nv.w) + (unsigned)uc;
#endif
#ifdef MI_PAIR_RECORDS
        {
            /* (x, y) (x, y+1) (x+1, y) (x+1, y+1) as they lie in memory -> the order the sampler names them in: x = (x, y), y = (x+1, y),
             * z = (x, y+1), w = (x+1, y+1) */
            const u32x4 r = *(gtex4u_t)(nv.img + MI_QUAD_WORDS * (size_t)rec);
            u32x4 o; o.x = r.x; o.y = r.z; o.z = r.y; o.w = r.w;
            return o;
        }
#else
        return *(gtex4_t)(nv.img + MI_QUAD_WORDS * (size_t)rec);
#endif
    };
#ifdef MI_EMU_LIN48
    /* (the other 32 bytes of the 48-byte record: gathered with the first 16, carried to where the sample is consumed -- as twelve
     * coefficients would be -- and touched there, so that the loads stay in flight as long as the real ones would) */
    auto record_more = [&](float uc, float vc, int part) -> u32x4 {
        const unsigned rec = __umul24((unsigned)vc, (unsigned)nv.w) + (unsigned)uc;
        return *(gtex4_t)(nv.img + MI_QUAD_WORDS * (size_t)rec + 4 * part);
    };
#endif
#ifdef MI_LDS_WINDOW
    Win win; win.x0 = win.y0 = 0; win.limx = win.limy = 0u; win.tile = nullptr;
    unsigned win_mx = 0u, win_my = 0u;            /* the largest tile coordinates a sample asked for (unsigned: below the box = huge) */
    auto record_win = [&](float uc, float vc) -> u32x4 {
        const unsigned lx = (unsigned)((int)uc - win.x0), ly = (unsigned)((int)vc - win.y0);
        win_mx = lx > win_mx ? lx : win_mx; win_my = ly > win_my ? ly : win_my;
        const unsigned cx = lx < win.limx ? lx : win.limx, cy = ly < win.limy ? ly : win.limy;      /* memory-safe; a miss redoes the pass */
        const uint32_t* t = win.tile + (__umul24(cy, (unsigned)MI_WIN_TP) + cx);
        u32x4 o; o.x = t[0]; o.y = t[1]; o.z = t[MI_WIN_TP]; o.w = t[MI_WIN_TP + 1];
        return o;
    };
#endif
    auto geom_w = [&](int it, auto use_win) -> Pre {
        Pre q;
        const int iraw = sub + it * L::LPV;
        q.live = iraw < MI_NS;                         /* L::LPV = 16: second trip only for lanes 0..8 */
        const int i = q.live ? iraw : (MI_NS - 1);
        q.i = i;
        const int dj = i / MI_FW - MI_HALF, di = i - (i / MI_FW) * MI_FW - MI_HALF;
        const float fi = (float)di, fj = (float)dj;
        const float g = geo[i];
        const float lam = (ps.depth + fi * ps.dzI + fj * ps.dzJ) * g;            /* computePatchPoints: t g */
        const float vx = nv.ax + fi * nv.bx + fj * nv.dx, vy = nv.ay + fi * nv.by + fj * nv.dy, vz = nv.az + fi * nv.bz + fj * nv.dz;
        const float sx = nv.sx + lam * vx, sy = nv.sy + lam * vy, sz = nv.sz + lam * vz;
        const float iz = fast_rcp(sz);
        const float u = sx * iz - 0.5f, v = sy * iz - 0.5f;                       /* worldToScreen */
        /* memory-safe even when the sample is outside (result discarded through ok): clamped into the image -- the footprint
         * records are edge-clamped themselves, so the last row / column is a valid record */
        const float uc = __builtin_amdgcn_fmed3f(u, kLo, wlim), vc = __builtin_amdgcn_fmed3f(v, kLo, hlim);
        ok = ok && (uc == u && vc == v);                                          /* the strict interior test (see above) */
        q.gu = 0.f; q.gv = 0.f;
        if (MODE != PASS_COLOR) {
            /* the point advanced by `step` along its ray (patch_sampler.cc:101-113); the derivative's 1 / stepSize folded in */
            const float l2 = step * g;
            const float iz1 = fast_rcp(sz + l2 * vz);
            const float u1 = (sx + l2 * vx) * iz1 - 0.5f, v1 = (sy + l2 * vy) * iz1 - 0.5f;
            q.gu = (u1 - u) * dnorm; q.gv = (v1 - v) * dnorm;
        }
        /* the bilinear weights: x - floor(x) in one instruction (v_fract_f32: the same value as the subtraction, which is
         * exact) */
        q.fx = __builtin_amdgcn_fractf(uc); q.fy = __builtin_amdgcn_fractf(vc);
#ifdef MI_LDS_WINDOW
        if constexpr (decltype(use_win)::value) q.t = record_win(uc, vc); else
#endif
        q.t = record(uc, vc);
#ifdef MI_EMU_LIN48
        q.e1 = record_more(uc, vc, 1); q.e2 = record_more(uc, vc, 2);
#endif
        return q;
    };
    auto geom = [&](int it) -> Pre { return geom_w(it, std::false_type{}); };
    auto geom2 = [&](int ita, int itb) -> Pre2 {
        Pre2 q;
        const int ra = sub + ita * L::LPV, rb = sub + itb * L::LPV;
        const bool la = ra < MI_NS, lb = rb < MI_NS;
        const int i0 = la ? ra : (MI_NS - 1), i1 = lb ? rb : (MI_NS - 1);
        q.i0 = i0; q.i1 = i1;
        q.wgt = mk2(la ? 1.f : 0.f, lb ? 1.f : 0.f);
        const int dj0 = i0 / MI_FW - MI_HALF, di0 = i0 - (i0 / MI_FW) * MI_FW - MI_HALF;
        const int dj1 = i1 / MI_FW - MI_HALF, di1 = i1 - (i1 / MI_FW) * MI_FW - MI_HALF;
        const f2 fi = mk2((float)di0, (float)di1), fj = mk2((float)dj0, (float)dj1);
        const f2 g = mk2(geo[i0], geo[i1]);
        const f2 lam = (sp2(ps.depth) + fi * sp2(ps.dzI) + fj * sp2(ps.dzJ)) * g;
        const f2 vx = sp2(nv.ax) + fi * sp2(nv.bx) + fj * sp2(nv.dx), vy = sp2(nv.ay) + fi * sp2(nv.by) + fj * sp2(nv.dy);
        const f2 vz = sp2(nv.az) + fi * sp2(nv.bz) + fj * sp2(nv.dz);
        const f2 sx = sp2(nv.sx) + lam * vx, sy = sp2(nv.sy) + lam * vy, sz = sp2(nv.sz) + lam * vz;
        const f2 iz = mk2(fast_rcp(sz.x), fast_rcp(sz.y));
        const f2 u = sx * iz - sp2(0.5f), v = sy * iz - sp2(0.5f);
        const f2 uc = mk2(__builtin_amdgcn_fmed3f(u.x, kLo, wlim), __builtin_amdgcn_fmed3f(u.y, kLo, wlim));
        const f2 vc = mk2(__builtin_amdgcn_fmed3f(v.x, kLo, hlim), __builtin_amdgcn_fmed3f(v.y, kLo, hlim));
        ok = ok && (uc.x == u.x && vc.x == v.x) && (uc.y == u.y && vc.y == v.y);
        q.gu = sp2(0.f); q.gv = sp2(0.f);
        if (MODE != PASS_COLOR) {
            const f2 l2 = sp2(step) * g;
            const f2 z1 = sz + l2 * vz;
            const f2 iz1 = mk2(fast_rcp(z1.x), fast_rcp(z1.y));
            const f2 u1 = (sx + l2 * vx) * iz1 - sp2(0.5f), v1 = (sy + l2 * vy) * iz1 - sp2(0.5f);
            q.gu = (u1 - u) * sp2(dnorm); q.gv = (v1 - v) * sp2(dnorm);
        }
        q.fx = mk2(__builtin_amdgcn_fractf(uc.x), __builtin_amdgcn_fractf(uc.y));
        q.fy = mk2(__builtin_amdgcn_fractf(vc.x), __builtin_amdgcn_fractf(vc.y));
        q.tA = record(uc.x, vc.x); q.tB = record(uc.y, vc.y);
#ifdef MI_EMU_LIN48
        q.eA1 = record_more(uc.x, vc.x, 1); q.eA2 = record_more(uc.x, vc.x, 2); q.eB1 = record_more(uc.y, vc.y, 1); q.eB2 = record_more(uc.y, vc.y, 2);
#endif
        return q;
    };
    auto consume = [&](const Pre& q) {
#ifdef MI_EMU_LIN48
        asm volatile("" : : "v"(q.e1), "v"(q.e2));
#endif
#ifdef MI_EXTRA_LDS
        /* experiment (make variant VFLAGS=-DMI_EXTRA_LDS=6): that many more look-ups in the sRGB table per sample (addresses as
         * scattered as the real ones), their values thrown away -- is it the LDS the throughput layout waits for? */
        {
#pragma unroll
          for (int k = 0; k < MI_EXTRA_LDS; ++k) { const float xl = s_lut[(q.t.x >> (3 + k)) & 255u]; asm volatile("" : : "v"(xl)); }
        }
#endif
#ifdef MI_EXTRA_VALU
        /* experiment (make variant VFLAGS=-DMI_EXTRA_VALU=12): that many VALU instructions per sample that compute nothing -- is the
         * throughput layout's time the time of its VALU instructions? */
        { float dummy = q.fx;
#pragma unroll
          for (int k = 0; k < MI_EXTRA_VALU; ++k) asm volatile("v_mov_b32 %0, %0" : "+v"(dummy));
        }
#endif
#ifdef MI_EXTRA_FMA
        /* the same question with instructions that are certainly executed at the plain rate: a dependent chain of v_fma_f32 (the
         * micro-benchmark's instruction, 3.7 cycles each at three wavefronts per SIMD) on a live value */
        { float dummy = q.fx;
#pragma unroll
          for (int k = 0; k < MI_EXTRA_FMA; ++k) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(dummy));
          asm volatile("" : : "v"(dummy));
        }
#endif
        const int i = q.i;
        const int dj = i / MI_FW - MI_HALF, di = i - (i / MI_FW) * MI_FW - MI_HALF;
        const float fx = q.fx, fy = q.fy, gu = q.gu, gv = q.gv;
        const float fxy = fx * fy, gm = gv * fx + gu * fy;
        const uint32_t t00 = q.t.x, t10 = q.t.y, t01 = q.t.z, t11 = q.t.w;
        float n[3], dr[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float c00 = s_lut[(t00 >> (8 * c)) & 255u], c10 = s_lut[(t10 >> (8 * c)) & 255u];
            const float c01 = s_lut[(t01 >> (8 * c)) & 255u], c11 = s_lut[(t11 >> (8 * c)) & 255u];
            /* the bilinear surface (mvs_tools.cc:119-128) as c00 + fx d1 + fy d2 + fx fy d3 and its exact directional
             * derivative (mvs_tools.cc:131-143) from the same three differences */
            const float d1 = c10 - c00, d2 = c01 - c00, d3 = (c11 - c10) - d2;
            n[c] = c00 + fx * d1 + fy * d2 + fxy * d3;
            if (MODE != PASS_COLOR) dr[c] = gu * d1 + gv * d2 + gm * d3;
        }
        const float wgt = (L::LPV == 1 || q.live) ? 1.f : 0.f;   /* dead trips (L::LPV > 1 only) contribute nothing */
        const float m0 = mcol[3 * i], m1 = mcol[3 * i + 1], m2 = mcol[3 * i + 2];
        if (MODE == PASS_DUMP) {
            dump_col[3 * i] = n[0]; dump_col[3 * i + 1] = n[1]; dump_col[3 * i + 2] = n[2];
            dump_der[3 * i] = dr[0]; dump_der[3 * i + 1] = dr[1]; dump_der[3 * i + 2] = dr[2];
        } else {
            if (WANT_S) {
            const float a0 = (n[0] - S.s0) * wgt, a1 = (n[1] - S.s1) * wgt, a2 = (n[2] - S.s2) * wgt;
            Pa0.x += a0; Pa1.x += a1; Pa2.x += a2;
            /* ba: sum m (n - s) here, - xbar sum (n - s) after the loop */
            if (PER_CHANNEL) {
                Paa0.x += a0 * a0; Paa1.x += a1 * a1; Paa2.x += a2 * a2;
                Pba0.x += m0 * a0; Pba1.x += m1 * a1; Pba2.x += m2 * a2;
            } else {
                Paa0.x += a0 * a0 + a1 * a1 + a2 * a2;
                Pba0.x += m0 * a0 + m1 * a1 + m2 * a2;
            }
            }
            if (FIXED) {
                const float g0 = ps.cs0 * dr[0], g1 = ps.cs1 * dr[1], g2 = ps.cs2 * dr[2];
                Pnum.x += wgt * (g0 * (m0 - ps.cs0 * n[0]) + g1 * (m1 - ps.cs1 * n[1]) + g2 * (m2 - ps.cs2 * n[2]));
                Pden.x += wgt * (g0 * g0 + g1 * g1 + g2 * g2);
            } else if (MODE == PASS_DEPTH) {
                const float e0 = dr[0] * wgt, e1 = dr[1] * wgt, e2 = dr[2] * wgt;
                Pdr0.x += e0 * (m0 - ps.cs0 * n[0]); Pdr1.x += e1 * (m1 - ps.cs1 * n[1]); Pdr2.x += e2 * (m2 - ps.cs2 * n[2]);
                Pdn0.x += e0 * n[0]; Pdn1.x += e1 * n[1]; Pdn2.x += e2 * n[2];
                Pdd0.x += e0 * dr[0]; Pdd1.x += e1 * dr[1]; Pdd2.x += e2 * dr[2];
            } else if (MODE == PASS_NORMAL) {
                const float g0 = ps.cs0 * dr[0], g1 = ps.cs1 * dr[1], g2 = ps.cs2 * dr[2];
                const float r0_ = m0 - ps.cs0 * n[0], r1_ = m1 - ps.cs1 * n[1], r2_ = m2 - ps.cs2 * n[2];
                const float gg = (g0 * g0 + g1 * g1 + g2 * g2) * wgt;
                const float gr = (g0 * r0_ + g1 * r1_ + g2 * r2_) * wgt;
                const float fi = (float)di, fj = (float)dj;
                A00 += (acc_t)gg; A01 += (acc_t)(fi * gg); A02 += (acc_t)(fj * gg);
                A11 += (acc_t)(fi * fi * gg); A12 += (acc_t)(fi * fj * gg); A22 += (acc_t)(fj * fj * gg);
                B0 += (acc_t)gr; B1 += (acc_t)(fi * gr); B2 += (acc_t)(fj * gr);
            }
        }
    };
    auto consume2 = [&](const Pre2& q) {
#ifdef MI_EMU_LIN48
        asm volatile("" : : "v"(q.eA1), "v"(q.eA2), "v"(q.eB1), "v"(q.eB2));
#endif
        const int i0 = q.i0, i1 = q.i1;
        const f2 fx = q.fx, fy = q.fy, gu = q.gu, gv = q.gv;
        const f2 fxy = fx * fy, gm = gv * fx + gu * fy;
        f2 n[3], dr[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const f2 c00 = mk2(s_lut[(q.tA.x >> (8 * c)) & 255u], s_lut[(q.tB.x >> (8 * c)) & 255u]);
            const f2 c10 = mk2(s_lut[(q.tA.y >> (8 * c)) & 255u], s_lut[(q.tB.y >> (8 * c)) & 255u]);
            const f2 c01 = mk2(s_lut[(q.tA.z >> (8 * c)) & 255u], s_lut[(q.tB.z >> (8 * c)) & 255u]);
            const f2 c11 = mk2(s_lut[(q.tA.w >> (8 * c)) & 255u], s_lut[(q.tB.w >> (8 * c)) & 255u]);
            const f2 d1 = c10 - c00, d2 = c01 - c00, d3 = (c11 - c10) - d2;
            n[c] = c00 + fx * d1 + fy * d2 + fxy * d3;
            if (MODE != PASS_COLOR) dr[c] = gu * d1 + gv * d2 + gm * d3;
        }
        const f2 m0 = mk2(mcol[3 * i0], mcol[3 * i1]), m1 = mk2(mcol[3 * i0 + 1], mcol[3 * i1 + 1]), m2 = mk2(mcol[3 * i0 + 2], mcol[3 * i1 + 2]);
        if (MODE == PASS_DUMP) {
            dump_col[3 * i0] = n[0].x; dump_col[3 * i0 + 1] = n[1].x; dump_col[3 * i0 + 2] = n[2].x;
            dump_der[3 * i0] = dr[0].x; dump_der[3 * i0 + 1] = dr[1].x; dump_der[3 * i0 + 2] = dr[2].x;
            dump_col[3 * i1] = n[0].y; dump_col[3 * i1 + 1] = n[1].y; dump_col[3 * i1 + 2] = n[2].y;
            dump_der[3 * i1] = dr[0].y; dump_der[3 * i1 + 1] = dr[1].y; dump_der[3 * i1 + 2] = dr[2].y;
        } else {
            if (WANT_S) {
            f2 a0 = n[0] - Ss0, a1 = n[1] - Ss1, a2 = n[2] - Ss2;
            if (L::LPV != 1) { a0 *= q.wgt; a1 *= q.wgt; a2 *= q.wgt; }      /* dead trips (L::LPV > 1 only) contribute nothing */
            Pa0 += a0; Pa1 += a1; Pa2 += a2;
            if (PER_CHANNEL) {
                Paa0 += a0 * a0; Paa1 += a1 * a1; Paa2 += a2 * a2;
                Pba0 += m0 * a0; Pba1 += m1 * a1; Pba2 += m2 * a2;
            } else {
                Paa0 += a0 * a0 + a1 * a1 + a2 * a2;
                Pba0 += m0 * a0 + m1 * a1 + m2 * a2;
            }
            }
            if (FIXED) {
                const f2 g0 = Cs0 * dr[0], g1 = Cs1 * dr[1], g2 = Cs2 * dr[2];
                f2 tn = g0 * (m0 - Cs0 * n[0]) + g1 * (m1 - Cs1 * n[1]) + g2 * (m2 - Cs2 * n[2]);
                f2 td = g0 * g0 + g1 * g1 + g2 * g2;
                if (L::LPV != 1) { tn *= q.wgt; td *= q.wgt; }
                Pnum += tn; Pden += td;
            } else if (MODE == PASS_DEPTH) {
                f2 e0 = dr[0], e1 = dr[1], e2 = dr[2];
                if (L::LPV != 1) { e0 *= q.wgt; e1 *= q.wgt; e2 *= q.wgt; }
                Pdr0 += e0 * (m0 - Cs0 * n[0]); Pdr1 += e1 * (m1 - Cs1 * n[1]); Pdr2 += e2 * (m2 - Cs2 * n[2]);
                Pdn0 += e0 * n[0]; Pdn1 += e1 * n[1]; Pdn2 += e2 * n[2];
                Pdd0 += e0 * dr[0]; Pdd1 += e1 * dr[1]; Pdd2 += e2 * dr[2];
            } else if (MODE == PASS_NORMAL) {
                const f2 g0 = Cs0 * dr[0], g1 = Cs1 * dr[1], g2 = Cs2 * dr[2];
                const f2 r0_ = m0 - Cs0 * n[0], r1_ = m1 - Cs1 * n[1], r2_ = m2 - Cs2 * n[2];
                f2 gg = g0 * g0 + g1 * g1 + g2 * g2;
                f2 gr = g0 * r0_ + g1 * r1_ + g2 * r2_;
                if (L::LPV != 1) { gg *= q.wgt; gr *= q.wgt; }
                const int dj0 = i0 / MI_FW - MI_HALF, di0 = i0 - (i0 / MI_FW) * MI_FW - MI_HALF;
                const int dj1 = i1 / MI_FW - MI_HALF, di1 = i1 - (i1 / MI_FW) * MI_FW - MI_HALF;
                const f2 fi = mk2((float)di0, (float)di1), fj = mk2((float)dj0, (float)dj1);
                const f2 igg = fi * gg, jgg = fj * gg, iigg = fi * igg, ijgg = fi * jgg, jjgg = fj * jgg, igr = fi * gr, jgr = fj * gr;
                PA00 += gg; PA01 += igg; PA02 += jgg; PA11 += iigg; PA12 += ijgg; PA22 += jjgg;
                PB0 += gr; PB1 += igr; PB2 += jgr;
            }
        }
    };
    if (L::LPV == 1) {
        /* a row of the window per gather round (5 x 5: its five footprint records are neighbours in memory, 1-2
         * cache lines fetched once, five gathers in flight).  One sample at a time: in this layout the packed form measured
         * SLOWER (profiles/r5_ab_experiments.txt P: three wavefronts per SIMD already fill the issue slots, a packed
         * instruction costs more than half of two plain ones, and the register pairs spill) */
#ifdef MI_TOUCH_ROWS
        /* experiment: the cache lines of the window's LATER rows are asked for before the first row is sampled -- one 4-byte load at
         * the middle sample of each of those rows, its value thrown away at the end of the pass: the rows' own 16-byte gathers then
         * find their lines on the way or there.  (What the throughput layout waits for, r6_ab_experiments.txt H: neither its VALU
         * instructions nor the LDS -- 24 more of the one, 6 more look-ups of the other per sample cost nothing / 5 % -- but the
         * slowest line of every row's gathers.) */
        uint32_t touch[MI_FW];
        touch[0] = 0u;
#pragma unroll
        for (int row = 1; row < MI_FW; ++row) {
            const int i = row * MI_FW + MI_HALF;
            const float fj = (float)(row - MI_HALF);
            const float lam = (ps.depth + fj * ps.dzJ) * geo[i];
            const float vx = nv.ax + fj * nv.dx, vy = nv.ay + fj * nv.dy, vz = nv.az + fj * nv.dz;
            const float iz = fast_rcp(nv.sz + lam * vz);
            const float uc = __builtin_amdgcn_fmed3f((nv.sx + lam * vx) * iz - 0.5f, kLo, wlim), vc = __builtin_amdgcn_fmed3f((nv.sy + lam * vy) * iz - 0.5f, kLo, hlim);
            touch[row] = GU(nv.img + MI_QUAD_WORDS * (size_t)(__umul24((unsigned)vc, (unsigned)nv.w) + (unsigned)uc));
        }
#endif
#ifdef MI_ROW2
        /* experiment: the gathers of the NEXT row are issued before this row's samples are consumed (ten in flight instead of five;
         * needs the registers of two wavefronts per SIMD: -DMI_WAVES_PER_SIMD=2) */
        {
            Pre qa[MI_FW], qb[MI_FW];
#pragma unroll
            for (int k = 0; k < MI_FW; ++k) qa[k] = geom(k);
#pragma unroll 1
            for (int row = 0; row < MI_FW; row += 2) {
                if (row + 1 < MI_FW) {
#pragma unroll
                    for (int k = 0; k < MI_FW; ++k) qb[k] = geom((row + 1) * MI_FW + k);
                }
#pragma unroll
                for (int k = 0; k < MI_FW; ++k) consume(qa[k]);
                if (row + 2 < MI_FW) {
#pragma unroll
                    for (int k = 0; k < MI_FW; ++k) qa[k] = geom((row + 2) * MI_FW + k);
                }
                if (row + 1 < MI_FW) {
#pragma unroll
                    for (int k = 0; k < MI_FW; ++k) consume(qb[k]);
                }
            }
        }
#else
        bool on_window = false;
#ifdef MI_LDS_WINDOW
        if constexpr (WIN) {
            const bool ok_in = ok;
            on_window = window_build(ps, nv, geo, (int)(threadIdx.x & 63u), win);
            if (on_window) {
#pragma unroll 1
                for (int row = 0; row < MI_FW; ++row) {
                    Pre q[MI_FW];
#pragma unroll
                    for (int k = 0; k < MI_FW; ++k) q[k] = geom_w(row * MI_FW + k, std::true_type{});
#pragma unroll
                    for (int k = 0; k < MI_FW; ++k) consume(q[k]);
                }
                if (__ballot(win_mx > win.limx || win_my > win.limy) != 0ull) {
                    /* a sample outside its box (its texels were read from the box's edge): the pass is redone on the global gathers */
                    on_window = false; ok = ok_in;
                    if ((int)(threadIdx.x & 63u) == __ffsll((long long)__ballot(true)) - 1) atomicAdd(&g_wstat[1], 1u);
                    Pa0 = Pa1 = Pa2 = Paa0 = Paa1 = Paa2 = Pba0 = Pba1 = Pba2 = sp2(0.f);
                    Pdr0 = Pdr1 = Pdr2 = Pdn0 = Pdn1 = Pdn2 = Pdd0 = Pdd1 = Pdd2 = sp2(0.f);
                    Pnum = Pden = sp2(0.f);
                    A00 = A01 = A02 = A11 = A12 = A22 = B0 = B1 = B2 = 0;
                }
            }
            const unsigned long long actw = __ballot(true);
            if (on_window && (int)(threadIdx.x & 63u) == __ffsll((long long)actw) - 1) atomicAdd(&g_wstat[0], 1u);
        }
#endif
        if (!on_window) {
#pragma unroll 1
        for (int row = 0; row < MI_FW; ++row) {
            Pre q[MI_FW];
#pragma unroll
            for (int k = 0; k < MI_FW; ++k) q[k] = geom(row * MI_FW + k);
#pragma unroll
            for (int k = 0; k < MI_FW; ++k) consume(q[k]);
        }
        }
#endif
#ifdef MI_TOUCH_ROWS
#pragma unroll
        for (int row = 1; row < MI_FW; ++row) asm volatile("" : : "v"(touch[row]));
#endif
    } else {
        constexpr int NP = NITER / 2;
        Pre2 q[NP > 0 ? NP : 1];
        Pre qs;
#pragma unroll
        for (int b = 0; b < NP; ++b) q[b] = geom2(2 * b, 2 * b + 1);
        if (NITER & 1) qs = geom(NITER - 1);
        TSTAMP(60);
#pragma unroll
        for (int b = 0; b < NP; ++b) consume2(q[b]);
        if (NITER & 1) consume(qs);
        TSTAMP(61);
    }
    /* the two halves of a pair accumulator (the throughput layout only ever used half x) */
    auto hs = [](f2 p) -> float { return L::LPV == 1 ? p.x : p.x + p.y; };
    if (MODE != PASS_DUMP && WANT_S) {
        S.a0 = L::view_sum(hs(Pa0)); S.a1 = L::view_sum(hs(Pa1)); S.a2 = L::view_sum(hs(Pa2));
        S.aa0 = L::view_sum(hs(Paa0)); S.ba0 = L::view_sum(hs(Pba0));
        if (PER_CHANNEL) {
            S.aa1 = L::view_sum(hs(Paa1)); S.aa2 = L::view_sum(hs(Paa2));
            S.ba1 = L::view_sum(hs(Pba1)); S.ba2 = L::view_sum(hs(Pba2));
            S.ba0 -= ps.xbar0 * S.a0; S.ba1 -= ps.xbar1 * S.a1; S.ba2 -= ps.xbar2 * S.a2;
        } else
            S.ba0 -= ps.xbar0 * S.a0 + ps.xbar1 * S.a1 + ps.xbar2 * S.a2;
        cs_out = S;
    }
    if (FIXED) { gn.num = L::view_sum(hs(Pnum)); gn.den = L::view_sum(hs(Pden)); }
    if (MODE == PASS_DEPTH) {
        gn.dr0 = L::view_sum(hs(Pdr0)); gn.dr1 = L::view_sum(hs(Pdr1)); gn.dr2 = L::view_sum(hs(Pdr2));
        gn.dn0 = L::view_sum(hs(Pdn0)); gn.dn1 = L::view_sum(hs(Pdn1)); gn.dn2 = L::view_sum(hs(Pdn2));
        gn.dd0 = L::view_sum(hs(Pdd0)); gn.dd1 = L::view_sum(hs(Pdd1)); gn.dd2 = L::view_sum(hs(Pdd2));
        gn.c00 = ps.cs0; gn.c01 = ps.cs1; gn.c02 = ps.cs2;
    }
    if (MODE == PASS_NORMAL) {
        if (L::LPV != 1) {
            A00 += (acc_t)(hs(PA00)); A01 += (acc_t)(hs(PA01)); A02 += (acc_t)(hs(PA02));
            A11 += (acc_t)(hs(PA11)); A12 += (acc_t)(hs(PA12)); A22 += (acc_t)(hs(PA22));
            B0 += (acc_t)(hs(PB0)); B1 += (acc_t)(hs(PB1)); B2 += (acc_t)(hs(PB2));
        }
        gn.A00 = (double)L::view_sum(A00); gn.A01 = (double)L::view_sum(A01); gn.A02 = (double)L::view_sum(A02);
        gn.A11 = (double)L::view_sum(A11); gn.A12 = (double)L::view_sum(A12); gn.A22 = (double)L::view_sum(A22);
        gn.B0 = (double)L::view_sum(B0); gn.B1 = (double)L::view_sum(B1); gn.B2 = (double)L::view_sum(B2);
    }
    TSTAMP(62);
    return L::view_all(ok);
}

/* getFastNCC from the shifted sums (patch_sampler.cc:135-163) */
__device__ __forceinline__ float ncc_from_sums(const PatchState& ps, const ColorSums& S) {
    const float inv_n = 1.f / (float)MI_NS;
    const float sqrDevY = (S.aa0 - S.a0 * S.a0 * inv_n) + (S.aa1 - S.a1 * S.a1 * inv_n) + (S.aa2 - S.a2 * S.a2 * inv_n);
    const float devXY = S.ba0 + S.ba1 + S.ba2;
    const float tmp = fast_sqrt(ps.sqrDevX * fmaxf(sqrDevY, 0.f));
    return tmp > 0.f ? fast_div(devXY, tmp) : -1.f;
}

/* Colour pass of view `gidx` (index into the job's global list) -> NCC; -1 on failure. */
template <class L>
__device__ __forceinline__ float eval_color(PatchState& ps, const DevView* views, int gidx, const float* s_lut,
                                            const float* geo, const float* mcol, ColorSums& S, bool& ok, bool count, int sub) {
    NView nv; int level; GNSums gn;
    ok = false;
    if (gidx < 0) return -1.f;
    if (!setup_view(views, ps.job->gv[gidx], ps, nv, level)) return -1.f;
    ok = sample_pass<PASS_COLOR, L>(ps, nv, s_lut, geo, mcol, S, gn, nullptr, nullptr, sub);
    ps.n_pass++;
    if (!ok) return -1.f;
    if (count) ps.n_eval++;
    return ncc_from_sums(ps, S);
}

/* PatchSampler::update + the quantities that depend on the state (patch_sampler.cc:258-295) */
__device__ __forceinline__ bool set_state(PatchState& ps, float depth, float dzI, float dzJ) {
    ps.depth = depth; ps.dzI = dzI; ps.dzJ = dzJ;
    /* tmpDepth is linear in (i, j): its minimum over the window is at a corner */
    const float a = (float)MI_HALF * fabsf(dzI) + (float)MI_HALF * fabsf(dzJ);
    const bool ok = (depth - a) > 0.f && depth == depth && a == a;
    /* footPrintScaled of the centre point: its depth along the reference's optical axis is depth x (the z component
     * of the unit pixel ray in camera coordinates) = depth g_c */
    ps.mfp = (depth * ps.inrm_c) * ps.jinv0;
    return ok;
}

/* The centre patch point (patchPoints[nrSamples / 2]) in world coordinates: only the view selection needs it. */
__device__ __forceinline__ void patch_centre(const PatchState& ps, float& px, float& py, float& pz) {
    const DevJob* job = ps.job;
    const float fx = (float)ps.x + 0.5f, fy = (float)ps.y + 0.5f;
    float rx = job->inv_a * fx + job->inv_c, ry = job->inv_b * fy + job->inv_d, rz = 1.f;
    rx *= ps.inrm_c; ry *= ps.inrm_c; rz *= ps.inrm_c;
    px = job->cam_pos[0] + ps.depth * (job->rot_t[0] * rx + job->rot_t[1] * ry + job->rot_t[2] * rz);
    py = job->cam_pos[1] + ps.depth * (job->rot_t[3] * rx + job->rot_t[4] * ry + job->rot_t[5] * rz);
    pz = job->cam_pos[2] + ps.depth * (job->rot_t[6] * rx + job->rot_t[7] * ry + job->rot_t[8] * rz);
}

/* computeColorScale for my view from the colour-pass sums (patch_optimization.cc:81-111).
 * Returns false where the reference sets optiSuccess = false. */
__device__ __forceinline__ bool color_scale_update(PatchState& ps, const ColorSums& S) {
    const float N = (float)MI_NS;
    bool good = true;
    float* cs[3] = {&ps.cs0, &ps.cs1, &ps.cs2};
    const float s[3] = {S.s0, S.s1, S.s2}, a[3] = {S.a0, S.a1, S.a2}, aa_[3] = {S.aa0, S.aa1, S.aa2};
    const float ba[3] = {S.ba0, S.ba1, S.ba2}, xb[3] = {ps.xbar0, ps.xbar1, ps.xbar2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float nn = aa_[c] + 2.f * s[c] * a[c] + N * s[c] * s[c];              /* sum n^2 */
        const float mn = ba[c] + xb[c] * a[c] + N * xb[c] * s[c];                   /* sum m n  */
        const float ab = mn - (*cs[c]) * nn;
        if (fabsf(nn) > 1e-6f) {
            *cs[c] += fast_div(ab, nn);
            if (*cs[c] > 1e3f) good = false;
        } else
            good = false;
    }
    return good;
}

__device__ __forceinline__ float parallax_to_weight(float p) {   /* mvs_tools.h:58-69 */
    if (p < 0.f || p > 180.f) return 0.f;
    const float sigma = (p <= 20.f) ? 5.f : 15.f;
    const float d = p - 20.f;
    return expf(-(d * d) / (2.f * sigma * sigma));
}

__device__ __forceinline__ void unit_dir(const float* cam, float px, float py, float pz, float& dx, float& dy, float& dz) {
    dx = px - cam[0]; dy = py - cam[1]; dz = pz - cam[2];
    const float n = sqrtf(dx * dx + dy * dy + dz * dz);
    dx /= n; dy /= n; dz /= n;
}
__device__ __forceinline__ void unit_cross(float ax, float ay, float az, float bx, float by, float bz,
                                           float& cx, float& cy, float& cz) {
    cx = ay * bz - az * by; cy = az * bx - ax * bz; cz = ax * by - ay * bx;
    const float n = sqrtf(cx * cx + cy * cy + cz * cz);
    cx /= n; cy /= n; cz /= n;
}
#define RAD2DEG 57.29577951308232f

/*
 * LocalViewSelection::performVS (local_view_selection.cc:56-147) for one patch.
 * Candidates (bits of ps.avail) are spread over the view slots of the patch; NCCs go through shared memory (lds_ncc).
 * On return each view slot's ps.sel holds its view (or -1); returns success.
 */
template <class L>
__device__ __forceinline__ bool local_view_selection(PatchState& ps, const DevSettings& st, const DevView* views, int lane) {
    const float* s_lut = g_lut;
    const float* geo = lds_geo<L>(L::patch(lane));
    const float* mcol = lds_mcol<L>(L::patch(lane));
    float* s_ncc = lds_ncc<L>(L::patch(lane), st.ncc_stride);
    const int slot = L::vslot(lane), sub = L::sub(lane);
    const int K = st.K;
    unsigned selmask = L::view_ballot(ps.sel >= 0, lane);
    if (__popc(selmask) == K) return true;
    const DevJob* J = ps.job;
    const int G = J->n_global;
    if (!ps.avail_ready) {
        /* LocalViewSelection ctor (local_view_selection.cc:19-54): every global view but the propagated ones.  Made here, at the
         * first selection a patch runs -- one patch in thousands does, and the two-word mask is a few dozen dependent
         * instructions that every attempt of the latency-bound kernels would otherwise pay in run_begin */
        static_assert(MI_AVAIL_WORDS >= 2, "the words park a propagated set of up to sixteen 8-bit indices");
        const unsigned long long hv = ps.avail.w[0], hx = ps.avail.w[1];
        ps.avail.first(G);
#pragma unroll
        for (int k = 0; k < L::NV; ++k) {
            const unsigned g = (unsigned)((k < 8 ? hv : hx) >> (8 * (k & 7))) & 0xFFu;
            if (g != MI_VIEW_NONE) ps.avail.clear((int)g);
        }
        ps.avail_ready = true;
    }
    /* NCC of every available candidate at the current state; drop those below minNCC */
    Avail drop;
    drop.first(0);
    for (int g = slot; g < G; g += L::NV) {
        if (!ps.avail.test(g)) continue;
        ColorSums S; bool ok;
        const float t = eval_color<L>(ps, views, g, s_lut, geo, mcol, S, ok, true, sub);
        if (t < st.minNCC) drop.set(g);
        if (sub == 0) s_ncc[g] = t;
    }
#pragma unroll
    for (int k = 0; k < MI_AVAIL_WORDS; ++k) ps.avail.w[k] &= ~L::patch_or(drop.w[k]);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    float p0x, p0y, p0z;
    patch_centre(ps, p0x, p0y, p0z);
    float rdx, rdy, rdz;
    unit_dir(J->cam_pos, p0x, p0y, p0z, rdx, rdy, rdz);                /* refDir */
    for (;;) {
        selmask = L::view_ballot(ps.sel >= 0, lane);
        if (__popc(selmask) >= K) break;
        /* the currently selected views, visible to every lane of the patch */
        int sl[L::NV];
        L::from_views(ps.sel, lane, sl);
        float best = 0.f; int bestg = -1;
        for (int g = slot; g < G; g += L::NV) {
            if (!ps.avail.test(g)) continue;
            const DevJobView* V = &J->gv[g];
            float score = s_ncc[g];
            const float z = V->w2c_z[0] * p0x + V->w2c_z[1] * p0y + V->w2c_z[2] * p0z + V->w2c_z[3];
            const float nfp = z * V->inv0;
            if (ps.mfp / nfp < 0.5f) score *= 0.01f;
            float vx, vy, vz;
            unit_dir(V->cam_pos, p0x, p0y, p0z, vx, vy, vz);
            float dp = fminf(fmaxf(rdx * vx + rdy * vy + rdz * vz, -1.f), 1.f);
            score *= parallax_to_weight(acosf(dp) * RAD2DEG);
            float ex, ey, ez;
            unit_cross(vx, vy, vz, rdx, rdy, rdz, ex, ey, ez);          /* epipolarPlane[i] */
#pragma unroll
            for (int k = 0; k < L::NV; ++k) {
                if (sl[k] < 0) continue;
                const DevJobView* U = &J->gv[sl[k]];
                float sx, sy, sz;
                unit_dir(U->cam_pos, p0x, p0y, p0z, sx, sy, sz);
                dp = fminf(fmaxf(sx * vx + sy * vy + sz * vz, -1.f), 1.f);
                score *= parallax_to_weight(acosf(dp) * RAD2DEG);
                float fx, fy, fz;
                unit_cross(sx, sy, sz, rdx, rdy, rdz, fx, fy, fz);
                dp = fminf(fmaxf(ex * fx + ey * fy + ez * fz, -1.f), 1.f);
                float angle = fabsf(acosf(dp) * RAD2DEG);
                if (angle > 90.f) angle = 180.f - angle;
                angle = fmaxf(angle, 1.f);
                if (angle < st.minParallax) score *= angle / st.minParallax;
            }
            if (score > best) { best = score; bestg = g; }
        }
        /* arg-max over the view slots; ties -> lowest index (strict '>' in an ascending scan, :134-138) */
        auto merge = [&](float ob, int og) {
            const bool take = og >= 0 && (bestg < 0 || ob > best || (ob == best && og < bestg));
            if (take) { best = ob; bestg = og; }
        };
        merge(__int_as_float(L::template view_xor<0>(__float_as_int(best))), L::template view_xor<0>(bestg));
        merge(__int_as_float(L::template view_xor<1>(__float_as_int(best))), L::template view_xor<1>(bestg));
        if (L::NV >= 8) merge(__int_as_float(L::template view_xor<2>(__float_as_int(best))), L::template view_xor<2>(bestg));
        if (L::NV == 16) merge(__int_as_float(L::template view_xor<3>(__float_as_int(best))), L::template view_xor<3>(bestg));
        if (bestg < 0) break;                                       /* foundOne == false */
        ps.avail.clear(bestg);
        /* give the view to the lowest free view slot */
        const unsigned freemask = ~selmask & ((1u << K) - 1u);
        const int target = __ffs(freemask) - 1;
        if (slot == target) {
            ps.sel = bestg;
            const float inv = 1.f / ps.mmean;                       /* colorScale init, patch_optimization.cc:73-76 */
            ps.cs0 = ps.cs1 = ps.cs2 = inv;
            ps.ncc = s_ncc[bestg];
        }
    }
    selmask = L::view_ballot(ps.sel >= 0, lane);
    return __popc(selmask) == K;
}

/* true iff every selected view with a smaller id than mine sampled successfully
 * (computeColorScale returns at the first failing view; std::set iterates ascending ids) */
template <class L>
__device__ __forceinline__ bool lower_views_ok(const PatchState& ps, bool my_ok, int lane) {
    int s[L::NV]; int o[L::NV];
    L::from_views(ps.sel, lane, s);
    L::from_views(my_ok ? 1 : 0, lane, o);
    bool all = true;
#pragma unroll
    for (int k = 0; k < L::NV; ++k)
        if (s[k] >= 0 && s[k] < ps.sel && !o[k]) all = false;
    return all;
}

struct PatchResult { float conf, depth, dzI, dzJ, nx, ny, nz; unsigned views, views_hi; int iters; };   /* views_hi: view slots 4..7 (eight-slot layouts) */

/* View slots 8..15 of a sixteen-slot set live apart from the rest -- two words per pixel in DevJob::views_x, per explicit
 * hypothesis in DevJob::hyp_x, per entry of a round's list in DevJob::results_x (all null unless nrReconNeighbors > 8) -- so that
 * the records and the kernels of four and eight slots are what they were. */
__device__ __forceinline__ unsigned long long load_x(const uint32_t* base, size_t i) {
    if (!base) return ~0ull;
    return ((unsigned long long)GU(base + 2 * i + 1) << 32) | GU(base + 2 * i);
}

/* a pixel's local view set from the state maps (slot `one`: the second state slot); the upper four of an eight-slot set
 * live in their own map, which only exists for nrReconNeighbors > 4 */
template <int NV>
__device__ __forceinline__ unsigned long long load_view_set(const DevJob* job, bool one, int p) {
    const unsigned lo = GU((one ? job->views1 : job->views) + p);
    unsigned hi = 0xFFFFFFFFu;
    if (NV >= 8) hi = GU((one ? job->views1_hi : job->views_hi) + p);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long view_set(unsigned lo, unsigned hi) { return ((unsigned long long)hi << 32) | lo; }

/* computeColorScale over the selected views from their colour sums (patch_optimization.cc:81-111):
 * ascending view order, stops at the first view whose sampling failed.  Returns false where the
 * reference sets optiSuccess = false. */
template <class L>
__device__ __forceinline__ bool color_scale_step(PatchState& ps, const DevSettings& st, const ColorSums& S, bool okv, int lane) {
    if (!st.useColorScale) return true;
    const bool active = ps.sel >= 0;
    const bool lower = lower_views_ok<L>(ps, okv || !active, lane);
    bool good = true;
    if (active && okv && lower) good = color_scale_update(ps, S);
    return L::view_ballot(!good, lane) == 0;
}

/*
 * Brings my view slot's cache up to the current patch state: view identity and mip level (patch_sampler.cc:72-91).
 * Returns false where the reference's sampling of this view fails before any texel is read (non-positive footprint).
 */
__device__ __forceinline__ bool view_prepare(const PatchState& ps, ViewC& vc, const DevView* __restrict__ views) {
    const float fx = (float)ps.x + 0.5f, fy = (float)ps.y + 0.5f;
    NView& nv = vc.nv;
    if (vc.sel != ps.sel) {
        vc.sel = ps.sel;
        const DevJobView* V = &ps.job->gv[ps.sel];
        vc.V = V;
        view_row2(nv, *V, fx, fy);
        vc.inv0 = V->inv0; vc.maxl = V->maxl;
        vc.lvl = -1;
    }
    const float z = nv.sz + (ps.depth * ps.inrm_c) * nv.az;          /* (worldToCam . centre point).z */
    const int mm = mip_level(z, vc.inv0, ps.mfp, vc.maxl);
    if (mm < 0) return false;
    if (mm != vc.lvl) {
        vc.lvl = mm;
        const DevJobView* V = vc.V;
        view_rows01(nv, *V, fx, fy);
        const DevView* DV = views + V->view;
        const DevLevel& Lv = DV->lv[mm];
        premultiply(nv, Lv.ax, Lv.ay, Lv.cx, Lv.cy);
        nv.w = Lv.w; nv.h = Lv.h;
        nv.img = DV->quad + MI_QUAD_WORDS * (size_t)Lv.tex_off;
#ifdef MI_LDS_WINDOW
        nv.raw = DV->img + Lv.tex_off;
#endif
    }
    return true;
}

/* One fused pass of my view at the current state; sets ps.ncc (getFastNCC).  Returns success[v]. */
template <int MODE, class L, bool WIN = false>
__device__ __forceinline__ bool run_pass(PatchState& ps, ViewC& vc, const DevView* views, const float* s_lut, const float* geo,
                                         const float* mcol, ColorSums& S, GNSums& gn, bool count_color, int sub) {
    bool okv = true;
    ps.ncc = -1.f;
    if (MODE == PASS_NORMAL) gn.A00 = gn.A01 = gn.A02 = gn.A11 = gn.A12 = gn.A22 = gn.B0 = gn.B1 = gn.B2 = 0.0;
    /* the throughput layout has no registers to spare for the cache: set the view up per pass */
    if (L::LPV == 1) viewc_reset(vc);
    if (ps.sel >= 0) {
        okv = view_prepare(ps, vc, views);
        TSTAMP(52);
        if (okv) okv = sample_pass<MODE, L, WIN>(ps, vc.nv, s_lut, geo, mcol, S, gn, nullptr, nullptr, sub);
        ps.n_pass++;
        if (okv && MODE != PASS_DEPTH_FIXED_NC) {
            ps.ncc = ncc_from_sums(ps, S);
            if (count_color) ps.n_eval++;
        }
    }
    return okv;
}

/*
 * PatchOptimization ctor + doAutoOptimization + computeConfidence for one patch, as three pieces so that
 * the turns of a patch are an explicit loop (optimize_patch):
 *   run_begin  PatchSampler / LocalViewSelection / PatchOptimization constructors
 *   run_turn   one turn of doAutoOptimization's pass-driven state machine
 *   run_end    getLocalViewIDs, computeConfidence, getPatchNormal
 * hyp_views: packed global indices of the propagated local view set (MI_VIEW_NONE = none).
 */
enum { CTX_CTOR, CTX_FIRST4, CTX_STEP, CTX_REPLACED, CTX_REPASS };

struct Run {
    PatchState ps;
    ViewC vc;                    /* my view slot's view, level and texel window */
    bool opti, converged, viewRemoved, step_was_normal, need_vs, count_color;
    bool bail;                   /* FAST kernels: the patch needs a view selection -- left to the general kernel */
    int iter, need, ctx;
    float oldncc;                /* per view slot: getFastNCC before the step (:189-192) */
    /* speculative attempts (k_optimize_spec): what an attempt would do to anything but its own result is only noted --
     * bit 0: the footprint exception, bits 8..15 / 16..23: views replaced / ... by the iteration-14 rule alone -- and carried
     * out by k_apply_spec for the attempts the reference's rule consumes (a discarded attempt must not fail a view) */
    unsigned deferred;
};

/* the pixel ray of (x, y) before / after normalisation (single_view.cc:106-114, mve/depthmap.cc:149-156: K_s^-1 at the
 * pixel centre, normalised, rotated into the world) */
__device__ __forceinline__ float pixel_scale(const DevJob* job, int x, int y) {
    const float rx = job->inv_a * ((float)x + 0.5f) + job->inv_c, ry = job->inv_b * ((float)y + 0.5f) + job->inv_d;
    return fast_rsqrt(rx * rx + ry * ry + 1.f);
}
__device__ __forceinline__ void pixel_ray(const DevJob* job, int x, int y, float& wx, float& wy, float& wz) {
    float rx = job->inv_a * ((float)x + 0.5f) + job->inv_c, ry = job->inv_b * ((float)y + 0.5f) + job->inv_d, rz = 1.f;
    const float inrm = fast_rsqrt(rx * rx + ry * ry + rz * rz);
    rx *= inrm; ry *= inrm; rz *= inrm;
    wx = job->rot_t[0] * rx + job->rot_t[1] * ry + job->rot_t[2] * rz;
    wy = job->rot_t[3] * rx + job->rot_t[4] * ry + job->rot_t[5] * rz;
    wz = job->rot_t[6] * rx + job->rot_t[7] * ry + job->rot_t[8] * rz;
}
/* the unit-ray scales of the window's pixels into LDS (NView: g) */
template <class L>
__device__ __forceinline__ void fill_geo(const DevJob* job, int x, int y, float* geo, int pl) {
    for (int i = pl; i < MI_NS; i += L::NV * L::LPV) {
        const int dj = i / MI_FW - MI_HALF, di = i - (i / MI_FW) * MI_FW - MI_HALF;
        geo[i] = pixel_scale(job, x + di, y + dj);
    }
}

/* getPatchNormal (patch_sampler.cc:242-256): the normalised cross product of the spans of the centre row (right - left
 * end: samples 14, 10 with 5 x 5 windows) and the centre column (top - bottom: 2, 22), and the centre pixel's view ray */
__device__ __forceinline__ void patch_normal(const PatchState& ps, float& nx, float& ny, float& nz, float& cx, float& cy, float& cz) {
    const DevJob* job = ps.job;
    float rx, ry, rz, lx, ly, lz, tx, ty, tz, bx_, by_, bz_;
    pixel_ray(job, ps.x + MI_HALF, ps.y, rx, ry, rz); pixel_ray(job, ps.x - MI_HALF, ps.y, lx, ly, lz);
    pixel_ray(job, ps.x, ps.y - MI_HALF, tx, ty, tz); pixel_ray(job, ps.x, ps.y + MI_HALF, bx_, by_, bz_);
    pixel_ray(job, ps.x, ps.y, cx, cy, cz);
    const float tr = ps.depth + (float)MI_HALF * ps.dzI, tl = ps.depth - (float)MI_HALF * ps.dzI;
    const float tt = ps.depth - (float)MI_HALF * ps.dzJ, tb = ps.depth + (float)MI_HALF * ps.dzJ;
    const float ax = tr * rx - tl * lx, ay = tr * ry - tl * ly, az = tr * rz - tl * lz;
    const float bx = tt * tx - tb * bx_, by = tt * ty - tb * by_, bz = tt * tz - tb * bz_;
    unit_cross(ax, ay, az, bx, by, bz, nx, ny, nz);
}

/* Returns false if the optimisation is over before it started (the result keeps confidence 0). */
template <class L>
__device__ __forceinline__ bool run_begin(Run& R, const DevJob* job, const DevSettings& st, const DevView* views, int x, int y,
                                          float depth0, float dzI0, float dzJ0, unsigned long long hyp_views, int lane, unsigned& err,
                                          DevCounters* counters, bool defer = false, unsigned long long hyp_x = ~0ull) {
    PatchState& ps = R.ps;
    const float* s_lut = g_lut;
    float* geo = lds_geo<L>(L::patch(lane));
    float* mcol = lds_mcol<L>(L::patch(lane));
    const int slot = L::vslot(lane), sub = L::sub(lane);
    const int pl = slot * L::LPV + sub;                 /* lane index inside the patch */
    ps.job = job; ps.x = x; ps.y = y; ps.n_eval = 0; ps.n_pass = 0; ps.counters = counters;
    ps.sel = -1; ps.cs0 = ps.cs1 = ps.cs2 = 1.f; ps.ncc = -1.f;
    ps.depth = depth0; ps.dzI = dzI0; ps.dzJ = dzJ0;
    viewc_reset(R.vc);
    R.opti = true; R.converged = false; R.viewRemoved = false; R.step_was_normal = false;
    R.iter = 0; R.need = PASS_DEPTH; R.ctx = CTX_CTOR; R.oldncc = -1.f; R.need_vs = false; R.count_color = false; R.bail = false;
    R.deferred = 0;
    /* --- PatchSampler ctor: border test (patch_sampler.cc:44-50) */
    if (x - MI_HALF < 0 || y - MI_HALF < 0 || x + MI_HALF > job->w - 1 || y + MI_HALF > job->h - 1) return false;
    ps.jinv0 = job->inv0_s;
    ps.inrm_c = pixel_scale(job, x, y);
    fill_geo<L>(job, x, y, geo, pl);
    /* raw master colours */
    const DevView* RV = views + job->ref_view;
    const DevLevel& RL = RV->lv[job->scale];
    const uint32_t* rimg = RV->img + RL.tex_off;
    /* computeMasterSamples (patch_sampler.cc:297-345) */
    float mm, x0, x1, x2, sd;
    if constexpr (!L::LAT) {
        /* raw master colours */
        for (int i = pl; i < MI_NS; i += L::NV * L::LPV) {
            const int dj = i / MI_FW - MI_HALF, di = i - (i / MI_FW) * MI_FW - MI_HALF;
            const uint32_t t = GU(rimg + (size_t)(y + dj) * RL.w + (x + di));
            mcol[3 * i] = s_lut[t & 255u]; mcol[3 * i + 1] = s_lut[(t >> 8) & 255u]; mcol[3 * i + 2] = s_lut[(t >> 16) & 255u];
        }
        /* every lane of the patch redundantly, in the reference's summation order */
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        mm = 0.f;
        for (int k = 0; k < 3 * MI_NS; ++k) mm += mcol[k];
        mm /= 3.f * (float)MI_NS;
        if (mm < 0.01f || mm > 0.99f) return false;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int i = pl; i < MI_NS; i += L::NV * L::LPV) { mcol[3 * i] /= mm; mcol[3 * i + 1] /= mm; mcol[3 * i + 2] /= mm; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        x0 = 0.f; x1 = 0.f; x2 = 0.f;
        for (int i = 0; i < MI_NS; ++i) { x0 += mcol[3 * i]; x1 += mcol[3 * i + 1]; x2 += mcol[3 * i + 2]; }
        x0 /= (float)MI_NS; x1 /= (float)MI_NS; x2 /= (float)MI_NS;
        sd = 0.f;
        for (int i = 0; i < MI_NS; ++i) {
            const float a = mcol[3 * i] - x0, b = mcol[3 * i + 1] - x1, c = mcol[3 * i + 2] - x2;
            sd += a * a + b * b + c * c;
        }
    } else {
        /* the wavefront is the patch: lane pl holds samples pl, pl + 64, ... (one per lane up to 7 x 7 windows, two with
         * 9 x 9 and 11 x 11); wave-wide DPP reductions instead of 3 x 75 LDS reads */
        constexpr int NPL = (MI_NS + WAVE - 1) / WAVE;
        float raw0[NPL], raw1[NPL], raw2[NPL];
        float part = 0.f;
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int i = pl + q * WAVE;
            raw0[q] = raw1[q] = raw2[q] = 0.f;
            if (i < MI_NS) {
                const int dj = i / MI_FW - MI_HALF, di = i - (i / MI_FW) * MI_FW - MI_HALF;
                const uint32_t t = GU(rimg + (size_t)(y + dj) * RL.w + (x + di));
                raw0[q] = s_lut[t & 255u]; raw1[q] = s_lut[(t >> 8) & 255u]; raw2[q] = s_lut[(t >> 16) & 255u];
            }
            part += raw0[q] + raw1[q] + raw2[q];
        }
        mm = L::wave_sum(part) / (3.f * (float)MI_NS);
        if (mm < 0.01f || mm > 0.99f) return false;
        const float im = fast_rcp(mm);
        float p0 = 0.f, p1 = 0.f, p2 = 0.f;
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int i = pl + q * WAVE;
            raw0[q] *= im; raw1[q] *= im; raw2[q] *= im;
            if (i < MI_NS) { mcol[3 * i] = raw0[q]; mcol[3 * i + 1] = raw1[q]; mcol[3 * i + 2] = raw2[q]; }
            p0 += raw0[q]; p1 += raw1[q]; p2 += raw2[q];
        }
        x0 = L::wave_sum(p0) / (float)MI_NS;
        x1 = L::wave_sum(p1) / (float)MI_NS;
        x2 = L::wave_sum(p2) / (float)MI_NS;
        part = 0.f;
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const float a = raw0[q] - x0, b = raw1[q] - x1, c = raw2[q] - x2;
            part += (pl + q * WAVE < MI_NS) ? (a * a + b * b + c * c) : 0.f;
        }
        sd = L::wave_sum(part);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    ps.mmean = mm;
    ps.xbar0 = x0; ps.xbar1 = x1; ps.xbar2 = x2;
    ps.sqrDevX = sd;
    /* computePatchPoints */
    if (!set_state(ps, depth0, dzI0, dzJ0)) return false;
    if (!(ps.mfp > 0.f)) {                                 /* reference throws std::out_of_range here: the VIEW fails */
        if (defer) { R.deferred |= 1u; return false; }
        err |= 1u;
        atomicOr(const_cast<int32_t*>(&job->flags), (int)MI_JOB_EFOOTPRINT);
        return false;
    }

    TSTAMP(11);
    /* --- LocalViewSelection ctor (local_view_selection.cc:19-54) */
    /* (the availability mask itself is made by the first view selection the patch runs, from the set parked here) */
    ps.avail.first(0);
    ps.avail.w[0] = hyp_views; ps.avail.w[1] = L::NV == 16 ? hyp_x : ~0ull; ps.avail_ready = false;
    {
        int nprop = 0;
#pragma unroll
        for (int k = 0; k < L::NV; ++k) {
            const unsigned g = (unsigned)((k < 8 ? hyp_views : hyp_x) >> (8 * (k & 7))) & 0xFFu;
            if (g != MI_VIEW_NONE) { ++nprop; if (k == slot) ps.sel = (int)g; }
        }
        if (nprop > st.K) { ps.sel = -1; }          /* "Too many local neighbors propagated" */
        if (slot >= st.K) ps.sel = -1;
    }
    const float inv_mm = 1.f / mm;
    ps.cs0 = ps.cs1 = ps.cs2 = inv_mm;
    /* LocalViewSelection::performVS runs at exactly one place (top of run_turn): in the ctor when
     * fewer than K views were propagated (:56-62), and after replaceViews (:149-160) */
    R.need_vs = __popc(L::view_ballot(ps.sel >= 0, lane)) != st.K;
    R.count_color = !R.need_vs;   /* samples of views picked by the view selection are already cached there */
    return true;
}

/*
 * One turn of doAutoOptimization (patch_optimization.cc:170-242) as a pass-driven state machine: run ONE
 * fused pass at the current state (colour sums + the Gauss-Newton sums the NEXT step needs), finish the
 * decision of the step that led here, take the next step.  Returns false when the optimisation is over.
 */
template <class L, bool FAST, bool DEFER = false>
__device__ __forceinline__ bool run_turn(Run& R, const DevSettings& st, const DevView* views, int lane) {
    PatchState& ps = R.ps;
    const float* s_lut = g_lut;
    const float* geo = lds_geo<L>(L::patch(lane));
    const float* mcol = lds_mcol<L>(L::patch(lane));
    const int slot = L::vslot(lane), sub = L::sub(lane);
    const bool active = slot < st.K;               /* view slots 0..K-1 carry a view once the selection succeeded */
    /* the sums of a pass are consumed within the same turn */
    ColorSums S; GNSums gn;
    S.s0 = S.s1 = S.s2 = S.a0 = S.a1 = S.a2 = S.aa0 = S.aa1 = S.aa2 = S.ba0 = S.ba1 = S.ba2 = 0.f;
    if (R.need_vs) {
        /* FAST: no view selection in this kernel (it is rare once hypotheses propagate with their view sets: one patch
         * in a few thousand replaces a view) -- the attempt is abandoned and redone by the general kernel */
        if (FAST) { R.bail = true; return false; }
        R.need_vs = false;
        if (!local_view_selection<L>(ps, st, views, lane)) { R.opti = false; return false; }
    }
    bool okv;
    TSTAMP(20 + R.need);
#ifdef MI_LDS_WINDOW
    constexpr bool WIN = FAST && !L::LAT && L::NV == 4 && MI_FW == 5;      /* the first attempts of the large rounds (see g_win) */
#else
    constexpr bool WIN = false;
#endif
    if (R.need == PASS_DEPTH) okv = run_pass<PASS_DEPTH, L, WIN>(ps, R.vc, views, s_lut, geo, mcol, S, gn, R.count_color, sub);
    /* (the passes behind the depth-only steps of iterations 1..3: nobody asks for their NCC -- the reference's main loop, which
     * does, starts at iteration 4, patch_optimization.cc:177-192 --, so they carry no colour sums) */
    else if (R.need == PASS_DEPTH_FIXED && R.ctx == CTX_FIRST4 && !R.count_color)
        okv = run_pass<PASS_DEPTH_FIXED_NC, L, WIN>(ps, R.vc, views, s_lut, geo, mcol, S, gn, false, sub);
    else if (R.need == PASS_DEPTH_FIXED) okv = run_pass<PASS_DEPTH_FIXED, L, WIN>(ps, R.vc, views, s_lut, geo, mcol, S, gn, R.count_color, sub);
    else if (R.need == PASS_NORMAL) okv = run_pass<PASS_NORMAL, L, WIN>(ps, R.vc, views, s_lut, geo, mcol, S, gn, R.count_color, sub);
    else okv = run_pass<PASS_COLOR, L, WIN>(ps, R.vc, views, s_lut, geo, mcol, S, gn, R.count_color, sub);
    TSTAMP(30);
    /* ---- finish what led to this pass */
    if (R.ctx == CTX_CTOR || R.ctx == CTX_REPLACED) {
        /* computeColorScale() at the end of the ctor (:77) / after replaceViews (:231) */
        if (!color_scale_step<L>(ps, st, S, okv, lane)) { R.opti = false; return false; }
    } else if (R.ctx == CTX_STEP) {
        if (R.step_was_normal) {
            /* optimizeDepthAndNormal is followed by computeColorScale on the new state (:197-199) */
            if (!color_scale_step<L>(ps, st, S, okv, lane)) { R.opti = false; return false; }
        }
        /* convergence / view replacement (:207-239) */
        const float dn = fabsf(ps.ncc - R.oldncc);
        const bool moving = active && dn > st.minRefineDiff;
        const bool replace = active && (ps.ncc < st.acceptNCC || (R.iter == 14 && dn > st.minRefineDiff));
        const unsigned rmask = L::view_ballot(replace, lane);
        const bool conv = L::view_ballot(moving, lane) == 0;
        const unsigned r14 = L::view_ballot(replace && !(ps.ncc < st.acceptNCC), lane);
        if (rmask) {
            if (DEFER) R.deferred += ((unsigned)__popc(rmask) << 8) + ((unsigned)__popc(r14) << 16);   /* (a patch replaces a handful) */
            else if (!FAST && slot == 0 && sub == 0) {
                /* diagnostics (rare events): views replaced, and how many of them by the iteration-14 rule alone (not in the FAST
                 * kernels: a patch that replaces a view is abandoned there and redone -- and counted -- by the general kernel) */
                atomicAdd(&ps.counters->n_view_replaced, (unsigned long long)__popc(rmask));
                if (r14) atomicAdd(&ps.counters->n_iter14, (unsigned long long)__popc(r14));
            }
            R.viewRemoved = true;
            if (replace) ps.sel = -1;              /* available[] is already false for selected views */
            R.need_vs = true;
            ++R.iter;
            R.need = PASS_COLOR; R.ctx = CTX_REPLACED; R.count_color = false;   /* cached / VS-evaluated samples */
            return true;
        }
        if (conv) { R.converged = true; return false; }
        ++R.iter;
    }
    TSTAMP(31);                                        /* (what led to this pass is finished: colour scale / convergence / replacement) */
    /* ---- loop condition of the main loop (:185-186) */
    if (R.iter >= 4 && R.iter >= st.maxIterations) return false;
    /* ---- take the step of iteration `iter` from the sums of this pass */
    const bool first4 = R.iter < 4;
    const bool want_normal = !first4 && (R.iter % 5 == 4 || R.viewRemoved);
    const bool have_normal = R.need == PASS_NORMAL, have_depth = R.need == PASS_DEPTH || R.need == PASS_DEPTH_FIXED;
    if (want_normal ? !have_normal : !have_depth) {
        R.need = want_normal ? PASS_NORMAL : PASS_DEPTH_FIXED; R.ctx = CTX_REPASS; R.count_color = false;
        return true;
    }
    if (L::view_ballot(!okv, lane)) { R.opti = false; return false; }       /* fastColAndDeriv failed (:277-280,:321-324) */
    if (active) ps.n_eval++;                                                /* this pass stood in for fastColAndDeriv */
    R.oldncc = ps.ncc;
    bool step_ok = false;
    if (want_normal) {
        const double m0 = L::patch_sum(gn.A00), m1 = L::patch_sum(gn.A01), m2 = L::patch_sum(gn.A02);
        const double m4 = L::patch_sum(gn.A11), m5 = L::patch_sum(gn.A12), m8 = L::patch_sum(gn.A22);
        const double b0 = L::patch_sum(gn.B0), b1 = L::patch_sum(gn.B1), b2 = L::patch_sum(gn.B2);
        const double m3 = m1, m6 = m2, m7 = m5;
        /* libs/math/matrix_tools.h:392-399 (determinant), :462-476 (inverse) */
        const double det = m0 * m4 * m8 + m1 * m5 * m6 + m2 * m3 * m7 - m2 * m4 * m6 - m1 * m3 * m8 - m0 * m5 * m7;
        if (det == 0.0 || !(det == det)) { R.opti = false; return false; }
        const double i0 = m4 * m8 - m5 * m7, i1 = m2 * m7 - m1 * m8, i2 = m1 * m5 - m2 * m4;
        const double i3 = m5 * m6 - m3 * m8, i4 = m0 * m8 - m2 * m6, i5 = m2 * m3 - m0 * m5;
        const double i6 = m3 * m7 - m4 * m6, i7 = m1 * m6 - m0 * m7, i8 = m0 * m4 - m1 * m3;
        const float X0 = (float)((i0 * b0 + i1 * b1 + i2 * b2) / det);
        const float X1 = (float)((i3 * b0 + i4 * b1 + i5 * b2) / det);
        const float X2 = (float)((i6 * b0 + i7 * b1 + i8 * b2) / det);
        step_ok = set_state(ps, ps.depth + X0, ps.dzI + X1, ps.dzJ + X2);
        R.viewRemoved = false;
        R.step_was_normal = true;
    } else {
        /* optimizeDepthOnly (:265-299) from the colour-scale independent sums */
        float num = 0.f, den = 0.f;
        if (active && okv && R.need == PASS_DEPTH_FIXED) { num = gn.num; den = gn.den; }
        else if (active && okv) {
            num = ps.cs0 * (gn.dr0 - (ps.cs0 - gn.c00) * gn.dn0) + ps.cs1 * (gn.dr1 - (ps.cs1 - gn.c01) * gn.dn1)
                + ps.cs2 * (gn.dr2 - (ps.cs2 - gn.c02) * gn.dn2);
            den = ps.cs0 * ps.cs0 * gn.dd0 + ps.cs1 * ps.cs1 * gn.dd1 + ps.cs2 * ps.cs2 * gn.dd2;
        }
        num = L::patch_sum(num); den = L::patch_sum(den);
        if (den > 0.f) step_ok = set_state(ps, ps.depth + fast_div(num, den), ps.dzI, ps.dzJ);
        else step_ok = first4;                     /* the first four iterations tolerate denom <= 0 (:177-180) */
        R.step_was_normal = false;
    }
    TSTAMP(33);                                        /* (the step is taken: sums across the views, solve, set_state) */
    if (!step_ok) { R.opti = false; return false; }
    /* the colour scale only changes right after a normal step (and in the ctor): every other pass can
     * bake it in, which leaves 7 instead of 21 values to reduce across the view slot */
    if (first4) {
        ++R.iter;
        R.need = (R.iter >= 4 && R.iter % 5 == 4) ? PASS_NORMAL : PASS_DEPTH_FIXED;
        R.ctx = CTX_FIRST4;
        R.count_color = (R.iter == 4);             /* the reference first asks for NCCs when the main loop starts */
    } else {
        /* after a normal step computeColorScale follows: that pass must carry per-channel colour sums.  In the LATENCY layouts it is
         * a COLOUR pass: 94 % of the patches converge right there (their sixth and last pass: a third fewer instructions in it),
         * and a patch that goes on runs the pass of its next step with the new colour scale baked in (the REPASS above:
         * optimizeDepthOnly's sums as the reference forms them, patch_optimization.cc:283-290).  In the THROUGHPUT layouts it stays a
         * depth pass with colour-scale independent Gauss-Newton sums: a wavefront's 16 patches run in lockstep and two waves in
         * three hold a patch that goes on -- the extra pass of that one patch would cost the wavefront more than the lighter pass
         * of the fifteen others saves (measured: no gain from both changes together, profiles/r6_ab_experiments.txt G). */
        R.need = R.step_was_normal ? (L::LAT ? PASS_COLOR : PASS_DEPTH) : (((R.iter + 1) % 5 == 4) ? PASS_NORMAL : PASS_DEPTH_FIXED);
        R.ctx = CTX_STEP;
        R.count_color = true;
    }
    TSTAMP(32);
    return true;
}

template <class L>
__device__ __forceinline__ void run_end(Run& R, const DevSettings& st, int lane, PatchResult& res,
                                        unsigned& n_eval, unsigned& n_pass, unsigned long long* res_x = nullptr) {
    PatchState& ps = R.ps;
    n_eval += ps.n_eval; n_pass += ps.n_pass;
    res.conf = 0.f; res.nx = res.ny = res.nz = 0.f;
    res.iters = R.iter;
    res.depth = ps.depth; res.dzI = ps.dzI; res.dzJ = ps.dzJ;
    /* local view ids, ascending (std::set order) */
    constexpr int NV = L::NV;
    int s[NV];
    L::from_views(ps.sel, lane, s);
    {
        int t[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) t[k] = s[k];
#pragma unroll
        for (int a = 0; a < NV - 1; ++a)
#pragma unroll
            for (int b = 0; b < NV - 1 - a; ++b) {
                const unsigned ua = t[b] < 0 ? 0xFFFu : (unsigned)t[b], ub = t[b + 1] < 0 ? 0xFFFu : (unsigned)t[b + 1];
                if (ua > ub) { const int tmp = t[b]; t[b] = t[b + 1]; t[b + 1] = tmp; }
            }
        unsigned packed = 0, packed_hi = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < 4; ++k) packed |= (t[k] < 0 ? MI_VIEW_NONE : (unsigned)t[k]) << (8 * k);
        unsigned packed_x[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
        if (NV >= 8) {
            packed_hi = 0;
#pragma unroll
            for (int k = 4; k < 8; ++k) packed_hi |= (t[k] < 0 ? MI_VIEW_NONE : (unsigned)t[k]) << (8 * (k - 4));
        }
        if (NV == 16) {
            packed_x[0] = packed_x[1] = 0;
#pragma unroll
            for (int k = 8; k < NV; ++k) packed_x[(k - 8) >> 2] |= (t[k] < 0 ? MI_VIEW_NONE : (unsigned)t[k]) << (8 * (k & 3));
        }
        res.views = packed; res.views_hi = packed_hi;
        if (NV == 16 && res_x) *res_x = ((unsigned long long)packed_x[1] << 32) | packed_x[0];   /* view slots 8..15 */
    }
    if (!R.converged) return;
    /* --- computeConfidence (patch_optimization.cc:114-142): NCCs summed in ascending view order */
    float c[NV];
    {
        int ci[NV];
        L::from_views(__float_as_int(ps.ncc), lane, ci);
#pragma unroll
        for (int k = 0; k < NV; ++k) c[k] = __int_as_float(ci[k]);
    }
    float mean = 0.f; int cnt = 0;
    unsigned used = 0;
#pragma unroll
    for (int r = 0; r < NV; ++r) {
        int bi = -1;
#pragma unroll
        for (int k = 0; k < NV; ++k)
            if (s[k] >= 0 && !((used >> k) & 1u) && (bi < 0 || s[k] < s[bi])) bi = k;
        if (bi >= 0) { used |= 1u << bi; mean += c[bi]; ++cnt; }
    }
    mean /= (float)cnt;
    const float score = (mean - st.acceptNCC) / (1.f - st.acceptNCC);
    float nx, ny, nz, cx, cy, cz;
    patch_normal(ps, nx, ny, nz, cx, cy, cz);
    res.nx = nx; res.ny = ny; res.nz = nz;
    const float dotP = -(nx * cx + ny * cy + nz * cz);                     /* viewRayScaled(midx, midy) */
    res.conf = (dotP < 0.2f) ? 0.f : score;
    TSTAMP(41);
}

/* Returns false if the attempt was abandoned (FAST kernels only: it needs a view selection); res is void then and only the
 * passes actually run are counted. */
template <class L, bool FAST = false, bool DEFER = false>
__device__ __forceinline__ bool optimize_patch(const DevJob* job, const DevSettings& st, const DevView* views, int x, int y,
                               float depth0, float dzI0, float dzJ0, unsigned long long hyp_views, int lane,
                               PatchResult& res, unsigned& n_eval, unsigned& n_pass, unsigned& err, DevCounters* counters,
                               unsigned* deferred = nullptr, unsigned long long hyp_x = ~0ull, unsigned long long* res_x = nullptr) {
    Run R;
    TSTAMP(10);
    if (run_begin<L>(R, job, st, views, x, y, depth0, dzI0, dzJ0, hyp_views, lane, err, counters, DEFER, hyp_x)) {
#ifdef MI_ACTIVITY
        /* development build (make variant VFLAGS=-DMI_ACTIVITY): how many of a wavefront's patches are still at work in a
         * turn -- [0] patch-turns, [1] wavefront-turns (k_optimize flushes them into n_stage / n_gather_pass) */
        bool more;
        do {
            if (L::vslot(lane) == 0 && L::sub(lane) == 0) atomicAdd(&g_act[0], 1u);
            if (lane == __ffsll((long long)__ballot(true)) - 1) atomicAdd(&g_act[1], 1u);
            more = run_turn<L, FAST, DEFER>(R, st, views, lane);
        } while (more);
#else
        while (run_turn<L, FAST, DEFER>(R, st, views, lane)) { }
#endif
    }
    TSTAMP(40);
    if (DEFER) *deferred = R.deferred;
    if (FAST && R.bail) { n_pass += R.ps.n_pass; return false; }
    run_end<L>(R, st, lane, res, n_eval, n_pass, res_x);
    return true;
}

/* ------------------------------------------------------------------------- */

struct OptArgs {
    const DevJob* jobs;
    const DevView* views;
    const float* lut;
    DevSettings st;
    const DevEntry* work;
    const DevHyp* hyp;        /* explicit hypotheses (seeds / hook); null in propagate mode */
    DevResult* results;
    const unsigned* n_work_ptr;   /* device-side entry count (propagate, blind rounds) or null */
    unsigned n_work;
    unsigned min_work, max_work;  /* this launch only acts if min_work <= n < max_work (layout selection on device) */
    int round;
    DevCounters* counters;
    unsigned long long* tbuf;     /* MI_PROBE builds only: the debug buffer */
    /* One attempt per launch (throughput layout, host-visible rounds): an entry whose pixel has several candidate
     * hypotheses runs them in successive launches over compacted follow-up lists, so that wavefronts stay full
     * (16 patches) instead of idling 15 quads while one entry tries its second neighbour. */
    int max_attempts;             /* 4 = all attempts of an entry back to back; 1 = one, then hand over to follow_out */
    const unsigned* follow_in;    /* entry indices to continue (their state is in results[]), or null = first attempt */
    const unsigned* follow_in_n;
    unsigned* follow_out;         /* entries that still have untried candidates after this launch */
    unsigned* follow_out_n;
    /* A follow-up list is MI_XCDS SEGMENTS of follow_seg entries, each with a counter of its own (follow_*_n[segment]): the
     * workgroups b with equal b % MI_XCDS -- one XCD's, XcdRange -- append to their segment and, in the next launch, continue
     * it.  The first launch walks XCD x through the x-th eighth of the round's list (ordered by reference view and image tile);
     * its segment then holds entries of that eighth only, in roughly the order of the list -- where ONE list appended to by all
     * XCDs interleaved the eight ranges and mixed the views: the second attempts then ran 1.7 x slower per sampling pass than
     * the first ones (L2 hits 63 % against 85 %; profiles/r6_ab_experiments.txt). */
    unsigned follow_seg;
    unsigned follow_seg_in;       /* ... the same of the list this launch CONTINUES (0: one list with one counter, walked in eighths) */
    /* The FIRST follow-up list of a round in the ORDER of the round's list (the first launch of a large round): instead of appending,
     * every wavefront leaves the ballot of its patches that go on (bit = the patch's first lane) in follow_mask[unit];
     * k_follow_count / k_follow_scatter turn the masks into the list.  Appended by atomics a wavefront of that list held 16
     * strangers -- entries from all over an XCD's eighth --, whose footprint gathers share no cache line: such wavefronts run
     * 1.76 x slower per sampling pass than wavefronts of 16 neighbouring patches (measured by scrambling the first launch's own
     * entries: profiles/r6_ab_experiments.txt); in list order the 16 entries of a follow-up wavefront come from ~80 consecutive
     * ones: the same views, the same image rows. */
    unsigned long long* follow_mask;
    unsigned scramble;            /* experiment (MI_DMRECON_DEBUG_SCRAMBLE, first attempts only): 1 = the entries of a list are dealt to the
                                   * wavefronts in a scrambled order (a wavefront's 16 patches are no neighbours any more), 2 = whole
                                   * wavefront units in a scrambled order (neighbours within a wavefront, strangers across); same maps */
};

/*
 * All optimisation attempts of one work-list entry (pixel x, y of `job`) of a host-visible round, result into
 * a.results[e].  Explicit mode (seeds, parity hook): the one hypothesis given.  Propagate mode: the queue semantics of
 * dmrecon.cc:365-392 for the hypotheses pulled from the 4-neighbours that were written last round, best
 * confidence first.  Nothing but `best` and a 4-bit mask is kept in registers across an optimisation (the
 * candidates are re-read from the state, which a host-visible round does not write -- k_apply does).
 * Returns true if the pixel state must be overwritten.
 */
template <class L, bool FAST>
__device__ __forceinline__ bool process_entry(const OptArgs& a, unsigned e, const DevJob* job, int x, int y, int lane,
                                              unsigned& n_eval, unsigned& n_pass, unsigned& n_patch, unsigned& err, bool& more) {
    const bool writer = L::vslot(lane) == 0 && L::sub(lane) == 0;
    const bool explicit_hyp = a.hyp != nullptr;
    const bool resume = a.follow_in != nullptr;
    const int W = job->w;
    const int pix = y * W + x;
    float own = 0.f;
    if (!explicit_hyp) own = GF(job->conf + pix);
    float best = own;
    unsigned tried = 0;
    bool accepted = false;
    if (resume) {
        const DevResult prev = a.results[e];
        tried = prev.tried; accepted = prev.accepted != 0;
        if (accepted) best = prev.conf;
    } else if (writer) {
        DevResult z;
        z.conf = 0.f; z.depth = 0.f; z.dzI = z.dzJ = 0.f; z.nx = z.ny = z.nz = 0.f;
        z.views = 0xFFFFFFFFu; z.views_hi = 0xFFFFFFFFu; z.iters = 0; z.accepted = 0; z.tried = 0;
        a.results[e] = z;
    }
    more = false;
    int attempts = 0;
    const bool self_round = a.st.self_round != 0 && !explicit_hyp;
    for (int t = 0; t < 4; ++t) {
        float hd, hi, hj; unsigned long long hv, hx = ~0ull;
        const unsigned tried_before = tried;
        if (explicit_hyp) {
            if (t > 0) break;
            const DevHyp h = a.hyp[e];
            hd = h.depth; hi = h.dzI; hj = h.dzJ; hv = view_set(h.views, h.views_hi);
            if (L::NV == 16) hx = load_x(job->hyp_x, e);
        } else if (self_round) {
            /* the seed re-optimisation round: the pixel's own converged state is the one hypothesis (DevSettings::self_round) */
            if (t > 0) break;
            hd = GF(job->depth + pix); hi = GF(job->dz + 2 * pix); hj = GF(job->dz + 2 * pix + 1); hv = load_view_set<L::NV>(job, false, pix);
            if (L::NV == 16) hx = load_x(job->views_x, (size_t)pix);
        } else {
            const int nb[4] = {pix - 1, pix + 1, pix - W, pix + W};
            int bi = -1; float bc = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if ((tried >> k) & 1u) continue;
                const float c = GF(job->conf + nb[k]);
                const bool use = GI(job->upd + nb[k]) == a.round - 1 && (own < c - 0.05f || own == 0.f);
                if (use && (bi < 0 || c > bc)) { bi = k; bc = c; }
            }
            if (bi < 0) break;
            if (attempts >= a.max_attempts) {
                /* continued by the next launch -- unless the pop-time test (dmrecon.cc:371) would skip this candidate,
                 * and with it every remaining one (they come in descending confidence) */
                more = !(best > bc);
                break;
            }
            tried |= 1u << bi;
            if (best > bc) continue;                           /* dmrecon.cc:371 */
            const int p = nb[bi];
            hd = GF(job->depth + p); hi = GF(job->dz + 2 * p); hj = GF(job->dz + 2 * p + 1); hv = load_view_set<L::NV>(job, false, p);
            if (L::NV == 16) hx = load_x(job->views_x, (size_t)p);
        }
        PatchResult r; unsigned long long rx = ~0ull;
        if (!optimize_patch<L, FAST>(job, a.st, a.views, x, y, hd, hi, hj, hv, lane, r, n_eval, n_pass, err, a.counters, nullptr, hx, &rx)) {
            /* abandoned (FAST): the candidate stays untried, the follow-up launch of the general kernel takes the entry */
            tried = tried_before; more = true;
            break;
        }
        ++n_patch; ++attempts;
        const bool accept = explicit_hyp ? true : (r.conf > 0.f && best < r.conf);   /* dmrecon.cc:378,391 */
        if (accept) {
            best = r.conf;
            accepted = explicit_hyp ? (r.conf > 0.f) : true;
            if (writer) {
                DevResult o;
                o.conf = r.conf; o.depth = r.depth; o.dzI = r.dzI; o.dzJ = r.dzJ;
                o.nx = r.nx; o.ny = r.ny; o.nz = r.nz; o.views = r.views; o.views_hi = r.views_hi; o.iters = r.iters;
                o.accepted = accepted ? 1 : 0; o.tried = tried;
                a.results[e] = o;
                if (L::NV == 16 && job->results_x) { job->results_x[2 * (size_t)e] = (unsigned)rx; job->results_x[2 * (size_t)e + 1] = (unsigned)(rx >> 32); }
            }
        }
    }
    if (more && writer) a.results[e].tried = tried;
    return accepted;
}

/*
 * ONE optimisation attempt of a work-list entry -- the form of the large host-visible rounds: the first launch runs every
 * entry's first attempt (FAST), a second launch the second attempt of the entries the reference's rule still asks one of,
 * the rest goes to process_entry above (all remaining attempts in a row, a few per cent of the entries).  Same candidates in
 * the same order, same pop-time and acceptance tests (dmrecon.cc:365-392), same records as process_entry -- but nothing of
 * an entry's chain of attempts is live across the optimisation except `tried`, `best` and the runner-up's confidence: no
 * loop around optimize_patch, hence none of the 260-300 bytes of scratch per lane the loop form needs.
 * more: the entry has a further candidate the rule still asks for (or its attempt was abandoned: FAST).
 */
template <class L, bool FAST, bool SEED = false>
__device__ __forceinline__ void process_entry_single(const OptArgs& a, unsigned e, const DevJob* job, int x, int y, int lane,
                                                     unsigned& n_eval, unsigned& n_pass, unsigned& n_patch, unsigned& err, bool& more) {
    const bool writer = L::vslot(lane) == 0 && L::sub(lane) == 0;
    const bool resume = a.follow_in != nullptr;
    const int W = job->w;
    const int pix = y * W + x;
    if (!FAST && a.hyp != nullptr) {
        /* explicit mode (seeds, parity hook): the one hypothesis given; the result is always recorded */
        const DevHyp h = a.hyp[e];
        PatchResult r; unsigned long long rx = ~0ull;
        optimize_patch<L, false>(job, a.st, a.views, x, y, h.depth, h.dzI, h.dzJ, view_set(h.views, h.views_hi), lane, r, n_eval, n_pass, err, a.counters,
                                 nullptr, L::NV == 16 ? load_x(job->hyp_x, e) : ~0ull, &rx);
        ++n_patch;
        more = false;
        /* The SEED launch with the reference's seed semantics (DevSettings::seed_reopt): the reference pushes a seed's OWN pixel
         * (dmrecon.cc:316-326) and, when it pops it, optimises it once more from its converged state and view set; only if that
         * strictly raises the confidence is the pixel rewritten and are its neighbours pushed (:365-398).  Done here, in the
         * seed's own launch: the record carries the result that stands, the FIRST confidence (what processFeatures compared
         * when several features fall on one pixel: k_apply_seeds arbitrates by it) in `accepted`, and "propagates" in `tried`. */
        float c1 = r.conf; bool propagates = false;
        if (SEED && a.st.seed_reopt && r.conf > 0.f) {
            PatchResult r2; unsigned long long rx2 = ~0ull;
            optimize_patch<L, false>(job, a.st, a.views, x, y, r.depth, r.dzI, r.dzJ, view_set(r.views, r.views_hi), lane, r2, n_eval, n_pass, err, a.counters,
                                     nullptr, rx, &rx2);
            ++n_patch;
            if (r2.conf > 0.f && c1 < r2.conf) { r = r2; rx = rx2; propagates = true; }
        }
        if (writer) {
            DevResult o;
            o.conf = r.conf; o.depth = r.depth; o.dzI = r.dzI; o.dzJ = r.dzJ;
            o.nx = r.nx; o.ny = r.ny; o.nz = r.nz; o.views = r.views; o.views_hi = r.views_hi; o.iters = r.iters;
            o.accepted = r.conf > 0.f ? 1 : 0; o.tried = 0;
            if (SEED && a.st.seed_reopt) { o.accepted = c1 > 0.f ? (int32_t)__float_as_uint(c1) : 0; o.tried = propagates ? 1u : 0u; }
            a.results[e] = o;
            if (L::NV == 16 && job->results_x) { job->results_x[2 * (size_t)e] = (unsigned)rx; job->results_x[2 * (size_t)e + 1] = (unsigned)(rx >> 32); }
        }
        return;
    }
    const float own = GF(job->conf + pix);
    float best = own;
    unsigned tried = 0;
    if (resume) {
        const DevResult* prev = a.results + e;
        tried = GU(&prev->tried);
        if (GI(&prev->accepted) != 0) best = GF(&prev->conf);
    }
    /* the best untried candidate (highest source confidence, lowest direction on ties: process_entry's strict '>') and the
     * confidence of the one that would come after it */
    const int nb[4] = {pix - 1, pix + 1, pix - W, pix + W};
    /* (confidences are >= 0: a runner-up of -1 stands for "none" -- the pop-time test best > bc2 then holds by itself) */
    int bi = -1; float bc = 0.f, bc2 = -1.f;
    const bool self_round = a.st.self_round != 0;      /* the seed re-optimisation round: the one candidate is the pixel itself */
    if (self_round && !(tried & 1u)) { bi = 4; bc = own; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (self_round) break;
        if ((tried >> k) & 1u) continue;
        const float c = GF(job->conf + nb[k]);
        const bool use = GI(job->upd + nb[k]) == a.round - 1 && (own < c - 0.05f || own == 0.f);
        if (!use) continue;
        if (bi < 0 || c > bc) { if (bi >= 0) bc2 = fmaxf(bc2, bc); bi = k; bc = c; }
        else bc2 = fmaxf(bc2, c);
    }
    more = false;
    DevResult z;
    z.conf = 0.f; z.depth = 0.f; z.dzI = z.dzJ = 0.f; z.nx = z.ny = z.nz = 0.f;
    z.views = 0xFFFFFFFFu; z.views_hi = 0xFFFFFFFFu; z.iters = 0; z.accepted = 0; z.tried = tried;
    if (bi < 0 || best > bc) {                                 /* nothing (left) to try: dmrecon.cc:371 skips this candidate and every later one */
        if (!resume && writer) a.results[e] = z;
        return;
    }
    const int p = bi == 0 ? nb[0] : bi == 1 ? nb[1] : bi == 2 ? nb[2] : bi == 3 ? nb[3] : pix;
    const float hd = GF(job->depth + p), hi = GF(job->dz + 2 * p), hj = GF(job->dz + 2 * p + 1);
    const unsigned long long hv = load_view_set<L::NV>(job, false, p);
    PatchResult r; unsigned long long rx = ~0ull;
    if (!optimize_patch<L, FAST>(job, a.st, a.views, x, y, hd, hi, hj, hv, lane, r, n_eval, n_pass, err, a.counters,
                                 nullptr, L::NV == 16 ? load_x(job->views_x, (size_t)p) : ~0ull, &rx)) {
        /* abandoned (FAST): the candidate stays untried, the next launch (the general kernel) takes the entry */
        if (!resume && writer) a.results[e] = z;
        more = true;
        return;
    }
    ++n_patch;
    tried |= bi == 4 ? 1u : (1u << bi);
    const bool accept = r.conf > 0.f && best < r.conf;         /* dmrecon.cc:378,391 */
    if (accept) best = r.conf;
    more = !(best > bc2);
    if (writer) {
        if (accept) {
            DevResult o;
            o.conf = r.conf; o.depth = r.depth; o.dzI = r.dzI; o.dzJ = r.dzJ;
            o.nx = r.nx; o.ny = r.ny; o.nz = r.nz; o.views = r.views; o.views_hi = r.views_hi; o.iters = r.iters;
            o.accepted = 1; o.tried = tried;
            a.results[e] = o;
            if (L::NV == 16 && job->results_x) { job->results_x[2 * (size_t)e] = (unsigned)rx; job->results_x[2 * (size_t)e + 1] = (unsigned)(rx >> 32); }
        } else if (!resume) { z.tried = tried; a.results[e] = z; }
        else if (more) a.results[e].tried = tried;
    }
}

/*
 * Which part of a work list a workgroup takes.  The workgroups of a launch are dealt round-robin over the 8 XCDs (block b runs
 * on XCD b % 8: observed, not promised -- a wrong guess costs locality, nothing else), and every XCD has an L2 of its own.  A
 * round's list is ordered by reference view and image tile (k_generate's workgroups append in dispatch order), i.e. entries that
 * are neighbours in the list sample the same neighbour images at nearby positions.  Dealt out block by block, consecutive
 * wavefronts of the list land on eight different XCDs and every L2 fetches every image region for itself; here XCD x takes the
 * x-th EIGHTH of the list: its workgroups walk through one contiguous range, and what one wavefront's patches pulled into the
 * XCD's L2 is what the next wavefront's patches ask for.  (The launchers round grids up to multiples of 8.)
 * Usage: for (XcdRange r(n_units); r.next(u); ) -- u: the unit (a wavefront's worth of the list) this workgroup takes next.
 */
#ifndef MI_XCDS
#define MI_XCDS 8
#endif
struct XcdRange {
    unsigned per_x, base, j, jn;
    /* own_segment: n_units are the units of THIS XCD's own list (a follow-up segment), not of a list shared by all */
    __device__ __forceinline__ explicit XcdRange(unsigned n_units, bool own_segment = false) {
        const unsigned nx = (gridDim.x % MI_XCDS == 0 && gridDim.x >= MI_XCDS) ? MI_XCDS : 1u;   /* (a grid that is no multiple: block order) */
        per_x = own_segment ? n_units : (n_units + nx - 1) / nx;
        base = own_segment ? 0u : (blockIdx.x % nx) * per_x; j = blockIdx.x / nx; jn = gridDim.x / nx;
    }
    __device__ __forceinline__ bool next(unsigned& unit) {
        if (j >= per_x) return false;
        unit = base + j; j += jn;
        return true;
    }
};

/* flush counters: one atomic per wave (per-view counters live in the first lane of each view slot,
 * the patch counter in the first lane of each patch) */
template <class L>
__device__ __forceinline__ void flush_counters(DevCounters* counters, int lane, unsigned n_eval, unsigned n_pass,
                                               unsigned n_patch, unsigned n_filled, unsigned err, int kind) {
    if (L::sub(lane) != 0) { n_eval = 0; n_pass = 0; }
    if (L::vslot(lane) != 0 || L::sub(lane) != 0) { n_patch = 0; n_filled = 0; }
    for (int off = 32; off > 0; off >>= 1) {
        n_eval += __shfl_down(n_eval, off);
        n_pass += __shfl_down(n_pass, off);
        n_patch += __shfl_down(n_patch, off);
        n_filled += __shfl_down(n_filled, off);
        err |= __shfl_down(err, off);
    }
    if (lane == 0) {
        if (n_eval) { atomicAdd(&counters->n_eval, (unsigned long long)n_eval); atomicAdd(&counters->k_eval[kind], (unsigned long long)n_eval); }
        if (n_pass) { atomicAdd(&counters->n_pass, (unsigned long long)n_pass); atomicAdd(&counters->k_pass[kind], (unsigned long long)n_pass);
                      atomicAdd(&counters->k_pass_exec[kind], (unsigned long long)n_pass); }
        if (n_patch) { atomicAdd(&counters->n_patch, (unsigned long long)n_patch); atomicAdd(&counters->k_patch[kind], (unsigned long long)n_patch); }
        if (n_filled) atomicAdd(&counters->n_filled, (unsigned long long)n_filled);
        if (err) atomicOr(&counters->error_flags, err);
    }
}

/*
 * The hot kernel.  L::LPV = 1: 16 patches per wavefront (throughput); L::LPV = 16: one patch per
 * wavefront (latency).  Grid-stride over the work list, so the grid need not match its size.
 */
template <class L, bool FAST, bool SINGLE = FAST, bool SEED = false>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu((L::LAT ? 2 : MI_BULK_WAVES), (L::LAT ? 2 : MI_WAVES_PER_SIMD)))) void k_optimize(OptArgs a) {
    static_assert(SINGLE || !FAST, "the FAST kernel runs one attempt per entry");
    static_assert(!SEED || (SINGLE && !FAST), "the seed launch is the single-attempt form of the general kernel");
    const int lane = threadIdx.x;
    /* (follow-up lists: the segment of this workgroup's XCD, see OptArgs::follow_seg) */
    const bool grid8 = gridDim.x % MI_XCDS == 0 && gridDim.x >= MI_XCDS;
    const unsigned seg_n = (a.follow_seg != 0u && grid8) ? MI_XCDS : 1u, seg = blockIdx.x % seg_n;   /* (follow_seg 0: one list for all) */
    const unsigned seg_in_n = (a.follow_seg_in != 0u && grid8) ? MI_XCDS : 1u, seg_in = blockIdx.x % seg_in_n;
    const unsigned n = a.follow_in ? a.follow_in_n[seg_in] : (a.n_work_ptr ? *a.n_work_ptr : a.n_work);
    if (!a.follow_in && (n < a.min_work || n >= a.max_work)) return;
    for (int i = lane; i < 256; i += WAVE) g_lut[i] = a.lut[i];
#ifdef MI_ACTIVITY
    if (lane < 2) g_act[lane] = 0;
#endif
#ifdef MI_LDS_WINDOW
    if (FAST && !L::LAT && L::NV == 4 && lane < 4) g_wstat[lane] = 0;
#endif
    __syncthreads();
    /* (DevCounters::clk_shader / clk_real: the shader clock this launch runs at, sampled by every 1024th wavefront) */
    const bool clk_probe = (blockIdx.x & 1023u) == 0u;
    const unsigned long long clk_s0 = clk_probe ? (unsigned long long)clock64() : 0ull, clk_r0 = clk_probe ? (unsigned long long)wall_clock64() : 0ull;
    unsigned n_eval = 0, n_pass = 0, n_patch = 0, err = 0;
    unsigned unit;
    const unsigned* const fin = a.follow_in ? a.follow_in + (size_t)seg_in * a.follow_seg_in : nullptr;
    for (XcdRange xr((n + L::PATCHES - 1) / L::PATCHES, a.follow_in != nullptr && seg_in_n > 1u); xr.next(unit); ) {
        const unsigned i = unit * L::PATCHES + L::patch(lane);
        const bool live = i < n;                             /* (the last wavefront of the list: lanes without an entry idle) */
        unsigned e = !live ? 0u : (fin ? fin[i] : i);
        if (a.scramble && !fin && live && !a.follow_mask) {
            /* (a bijection of [0, n) resp. of the full units: multiplication by a prime modulo the size) */
            const unsigned n_full = n / L::PATCHES;
            if (a.scramble == 1u) e = (unsigned)(((unsigned long long)i * 1000003ull) % n);
            else if (unit < n_full) e = (unsigned)(((unsigned long long)unit * 1000003ull) % n_full) * L::PATCHES + L::patch(lane);
        }
        bool more = false;
        if (live) {
        const DevEntry ent = a.work[e];
        const DevJob* job = a.jobs + ent.job;
        if (GI(&job->flags) != 0) {
            /* the view failed (footprint exception) or was cancelled: nothing of it is touched any more */
            if (L::vslot(lane) == 0 && L::sub(lane) == 0) a.results[e].accepted = 0;
        } else if (SINGLE)
            process_entry_single<L, FAST, SEED>(a, e, job, ent.xy & 0xFFFF, ent.xy >> 16, lane, n_eval, n_pass, n_patch, err, more);
        else
            process_entry<L, false>(a, e, job, ent.xy & 0xFFFF, ent.xy >> 16, lane, n_eval, n_pass, n_patch, err, more);
        }
        if (a.follow_mask) {
            /* (the entries that go on, as a mask per wavefront unit: the list is made in list order afterwards, see OptArgs) */
            const unsigned long long m = __ballot(more && L::vslot(lane) == 0 && L::sub(lane) == 0);
            if (lane == 0) a.follow_mask[unit] = m;
        } else if (a.follow_out) {
            /* wave-aggregated append of the entries that still have candidates (one atomic per wavefront) */
            const bool mine = more && L::vslot(lane) == 0 && L::sub(lane) == 0;
            const unsigned long long m = __ballot(mine);
            if (m) {
                const int leader = __ffsll((long long)__ballot(true)) - 1;
                unsigned base = 0;
                if (lane == leader) base = atomicAdd(a.follow_out_n + seg, (unsigned)__popcll(m));
                base = (unsigned)__shfl((int)base, leader);
                if (mine) a.follow_out[(size_t)seg * a.follow_seg + base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = e;
            }
        }
    }
    flush_counters<L>(a.counters, lane, n_eval, n_pass, n_patch, 0u, err,
                      L::LAT ? MI_KIND_LAT : SEED ? MI_KIND_SEED : FAST ? MI_KIND_FAST : SINGLE ? MI_KIND_FOLLOW : MI_KIND_LOOP);
    if (clk_probe && lane == 0) {
        const unsigned long long ds = (unsigned long long)clock64() - clk_s0, dr = (unsigned long long)wall_clock64() - clk_r0;
        if (dr > 100ull) { atomicAdd(&a.counters->clk_shader, ds); atomicAdd(&a.counters->clk_real, dr); }   /* (wavefronts that found no work: too short to say anything) */
    }
#ifdef MI_LDS_WINDOW
    if constexpr (FAST && !L::LAT && L::NV == 4) {
        /* (wavefront-passes on LDS windows / on global gathers, reported as mi_dmrecon_stats::n_patch_turns / n_wave_turns) */
        __syncthreads();
        if (lane == 0) { atomicAdd(&a.counters->n_stage, (unsigned long long)g_wstat[0] | ((unsigned long long)g_wstat[1] << 32));
                         atomicAdd(&a.counters->n_gather_pass, (unsigned long long)g_wstat[2] | ((unsigned long long)g_wstat[3] << 32)); }
    }
#endif
#ifdef MI_ACTIVITY
    __syncthreads();
    if (lane == 0 && !L::LAT) { atomicAdd(&a.counters->n_stage, (unsigned long long)g_act[0]); atomicAdd(&a.counters->n_gather_pass, (unsigned long long)g_act[1] * L::PATCHES); }
#endif
}

/*
 * Small rounds of the throughput layout, speculatively.  A round whose list does not fill the GPU lasts as long as its
 * slowest wavefront, and that is one whose entry tries its second, third and fourth candidate hypothesis one after the
 * other (an attempt alone on a SIMD is ~100 us of exposed latencies).  The attempts of an entry do not depend on each other
 * -- they all read the state frozen at the start of the round; only the decisions which of them the reference would have
 * made, and which result stands, are sequential (dmrecon.cc:365-392) -- so here every (entry, rank) pair gets a quad of
 * its own, and the sequential rule is applied afterwards from the records (k_apply_spec): same maps and counters, bit for
 * bit, as process_entry's attempts in a row.
 */
struct SpecArgs {
    OptArgs o;               /* jobs, views, lut, st, work, n_work_ptr / n_work, min_work / max_work, round, counters */
    DevSpec* spec;           /* [4 x entries] */
    const unsigned* items;   /* the round's (entry, rank) pairs (k_generate), their number in *n_items */
    const unsigned* n_items;
};
template <class L>
__device__ __forceinline__ unsigned patch_sum_u(unsigned v) {
    v += (unsigned)L::template view_xor<0>((int)v); v += (unsigned)L::template view_xor<1>((int)v);
    if (L::NV >= 8) v += (unsigned)L::template view_xor<2>((int)v);
    if (L::NV == 16) v += (unsigned)L::template view_xor<3>((int)v);
    return v;
}
template <class L>
#ifndef MI_SPEC_WAVES
#define MI_SPEC_WAVES MI_WAVES_PER_SIMD
#endif
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu((MI_FW >= 7 ? 1 : MI_SPEC_WAVES), MI_SPEC_WAVES))) void k_optimize_spec(SpecArgs t) {
    const OptArgs& a = t.o;
    const int lane = threadIdx.x;
    const unsigned n = a.n_work_ptr ? *a.n_work_ptr : a.n_work;
    if (n < a.min_work || n >= a.max_work) return;
    for (int i = lane; i < 256; i += WAVE) g_lut[i] = a.lut[i];
    __syncthreads();
    const bool writer = L::vslot(lane) == 0 && L::sub(lane) == 0;
    unsigned err = 0, n_exec = 0;                         /* n_exec: passes this wavefront executed (DevCounters::k_pass_exec) */
    const unsigned n_items = *t.n_items;
    unsigned unit;
    for (XcdRange xr((n_items + L::PATCHES - 1) / L::PATCHES); xr.next(unit); ) {
        const unsigned i = unit * L::PATCHES + L::patch(lane);
        if (i >= n_items) continue;
        const unsigned item = t.items[i];
        const unsigned e = item >> 2; const int s = (int)(item & 3u);
        const DevEntry ent = a.work[e];
        const DevJob* job = a.jobs + ent.job;
        DevSpec* rec = t.spec + i;                        /* (an entry's items are consecutive, rank 0 first: k_generate) */
        if (GI(&job->flags) != 0) {                       /* the view failed or was cancelled: nothing of it is touched any more */
            if (writer && s == 0) rec->n_cand = 0;
            continue;
        }
        const int x = ent.xy & 0xFFFF, y = ent.xy >> 16, W = job->w, pix = y * W + x;
        const float own = GF(job->conf + pix);
        const int nb[4] = {pix - 1, pix + 1, pix - W, pix + W};
        /* the candidates in the reference's order of trial: descending source confidence, lowest direction first on ties
         * (process_entry picks them one by one with a strict '>') */
        float c[4]; unsigned elig = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            c[k] = GF(job->conf + nb[k]);
            if (GI(job->upd + nb[k]) == a.round - 1 && (own < c[k] - 0.05f || own == 0.f)) elig |= 1u << k;
        }
        int n_cand = 0, dir = -1; float bc = 0.f;
        if (a.st.self_round) { n_cand = 1; if (s == 0) { dir = 4; bc = own; } }       /* the seed re-optimisation round: the pixel itself */
        else {
            unsigned left = elig;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int bi = -1; float bv = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (((left >> k) & 1u) && (bi < 0 || c[k] > bv)) { bi = k; bv = c[k]; }
                if (bi >= 0) { if (r == s) { dir = bi; bc = bv; } ++n_cand; left &= ~(1u << bi); }
            }
        }
        if (writer && s == 0) rec->n_cand = n_cand;
        if (s >= n_cand) continue;
        const int p = dir == 0 ? nb[0] : dir == 1 ? nb[1] : dir == 2 ? nb[2] : dir == 3 ? nb[3] : pix;
        const float hd = GF(job->depth + p), hi = GF(job->dz + 2 * p), hj = GF(job->dz + 2 * p + 1);
        const unsigned long long hv = load_view_set<L::NV>(job, false, p);
        PatchResult r; unsigned ne = 0, np = 0;
        unsigned deferred = 0;
        optimize_patch<L, false, true>(job, a.st, a.views, x, y, hd, hi, hj, hv, lane, r, ne, np, err, a.counters, &deferred);
        if (L::sub(lane) != 0) { ne = 0; np = 0; }
        ne = patch_sum_u<L>(ne); np = patch_sum_u<L>(np);
        if (writer) n_exec += np;
        if (writer) {
            DevSpec o;
            o.conf = r.conf; o.depth = r.depth; o.dzI = r.dzI; o.dzJ = r.dzJ; o.nx = r.nx; o.ny = r.ny; o.nz = r.nz;
            o.views = r.views; o.views_hi = r.views_hi; o.iters = r.iters; o.bc = bc; o.own = own; o.n_eval = ne; o.n_pass = np;
            o.n_cand = n_cand; o.pad = (int32_t)deferred;
            *rec = o;
        }
    }
    for (int off = 32; off > 0; off >>= 1) { err |= __shfl_down(err, off); n_exec += __shfl_down(n_exec, off); }
    if (lane == 0 && err) atomicOr(&a.counters->error_flags, err);
    if (lane == 0 && n_exec) atomicAdd(&a.counters->k_pass_exec[MI_KIND_SPEC], (unsigned long long)n_exec);
}

/*
 * One fused round of the propagation tail (replaces generate -> optimise -> apply, three dependent launches, by one).
 *
 * A CANDIDATE is (pixel p accepted in the previous round, one of its 4-neighbours q) for which the push rule
 * (dmrecon.cc:400-431) holds against the state frozen at the end of the previous round.  The reference would try
 * q's candidates one after the other, best source confidence first, skipping a candidate once q has got a
 * confidence above its source's (pop-time test, dmrecon.cc:371) and accepting a result only if it beats the best
 * so far (:391).  The optimisations themselves do not depend on each other (they all read the frozen state), so
 * they run speculatively at the same time -- one WAVEFRONT each (latency layout), the up to four wavefronts of a
 * workgroup -- and the sequential rule is applied afterwards from their results in LDS: same maps, but a round
 * is as long as ONE patch optimisation instead of up to four in a row.
 *
 * Work distribution without claims or atomics: a workgroup is started for every (p, direction); it goes on only
 * if p is q's BEST source (highest confidence, lowest direction on ties) -- exactly one of q's candidates is.
 * Its wavefront w takes q's w-th best source; wavefronts without a source end at once.  The first wavefront
 * resolves, writes an accepted result into the pixel's other state slot (see DevJob) and appends q to this
 * round's list.  The order of the list depends on timing; the results do not.
 */
#define MI_TAIL_WAVES 4
static_assert(MI_TAIL_WAVES <= MI_LAT_SLOTS, "the latency layout's LDS holds one patch per wavefront of a tail workgroup");
/* wavefronts per SIMD the speculative tail kernel is compiled for: 2 = 222 VGPRs, no spills (experiment: 3 = 168, so
 * that a tail wavefront displaces one bulk wavefront of another call instead of two) */
#ifndef MI_TAIL_SPEC_WAVES
#define MI_TAIL_SPEC_WAVES 2
#endif
struct TailArgs {
    OptArgs o;                    /* o.work / o.results: this round's list and results (written here) */
    const DevEntry* prev_work;    /* previous round's list, results and entry count */
    const DevResult* prev_results;
    unsigned* round_work;         /* [MI_MAX_ROUNDS] accepted entries per round */
};

/* state of pixel p as the optimisations of round `round` must see it (frozen at the end of round - 1):
 * the slot with the larger stamp that is older than `round`.  Both slots are loaded at once (one latency). */
struct Frozen { float conf; int upd; bool one; };
__device__ __forceinline__ Frozen frozen_state(const DevJob* job, int p, int round) {
    const int s0 = GI(job->upd + p), s1 = GI(job->upd1 + p);
    const float c0 = GF(job->conf + p), c1 = GF(job->conf1 + p);
    Frozen f;
    f.one = s0 >= round || (s1 < round && s1 > s0);
    f.conf = f.one ? c1 : c0; f.upd = f.one ? s1 : s0;
    return f;
}

struct TailRes { PatchResult r; unsigned n_eval, n_pass; };
__shared__ TailRes g_tail_res[MI_TAIL_WAVES];

template <bool SPEC, int NV>
__global__ __launch_bounds__((SPEC ? MI_TAIL_WAVES : 1) * WAVE) __attribute__((amdgpu_waves_per_eu((SPEC ? MI_TAIL_SPEC_WAVES : (MI_FW > 7 ? 1 : MI_FW == 7 ? 2 : MI_WAVES_PER_SIMD)), (SPEC ? MI_TAIL_SPEC_WAVES : MI_WAVES_PER_SIMD)))) void k_tail(TailArgs t) {
    typedef typename LatLay<NV>::type LL;
    const OptArgs& a = t.o;
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6;
    const unsigned n_prev = t.round_work[a.round - 1];
    if (blockIdx.x >= 4u * n_prev) return;
    /* one candidate per workgroup (the usual case): wavefronts without a source can end; otherwise the workgroup
     * strides over the candidates and all its wavefronts stay for the barriers */
    const bool single = 4u * n_prev <= gridDim.x;
    /* the sRGB table: three workgroups in four end at the candidate tests below and never sample -- with one candidate
     * per workgroup a wavefront stages the table itself (every wavefront the same 256 values) just before its first
     * patch, instead of all of them waiting on a load and a barrier up front */
    bool lut_ready = !single;
    if (!single) {
        for (int i = threadIdx.x; i < 256; i += (SPEC ? MI_TAIL_WAVES : 1) * WAVE) g_lut[i] = a.lut[i];
        __syncthreads();
    }
    unsigned n_eval = 0, n_pass = 0, n_patch = 0, n_filled = 0, err = 0;
    for (unsigned cand = blockIdx.x; cand < 4u * n_prev; cand += gridDim.x) {
        const unsigned ep = cand >> 2, k = cand & 3u;
        const DevResult pr = t.prev_results[ep];
        const DevEntry src = t.prev_work[ep];
        if (!pr.accepted) continue;
        const DevJob* job = a.jobs + src.job;
        if (GI(&job->flags) != 0) continue;                                    /* failed / cancelled view */
        const int W = job->w, H = job->h;
        const int qx = (src.xy & 0xFFFF) + (k == 0 ? -1 : k == 1 ? 1 : 0), qy = (src.xy >> 16) + (k == 2 ? -1 : k == 3 ? 1 : 0);
        if (qx < MI_HALF || qy < MI_HALF || qx >= W - MI_HALF || qy >= H - MI_HALF) continue;   /* patch_sampler.cc:47-50 */
        const int q = qy * W + qx;
        /* frozen state of q and of its four neighbours, all loads in flight together */
        const int nb[4] = {q - 1, q + 1, q - W, q + W};
        const Frozen me = frozen_state(job, q, a.round);
        Frozen nf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) nf[j] = frozen_state(job, nb[j], a.round);
        const float own = me.conf;
        if (!(own < pr.conf - 0.05f || own == 0.f)) continue;
        /* q's candidates of this round: neighbours written last round for which the push rule holds, in the
         * reference's order of trial: descending source confidence, lowest direction first on ties */
        unsigned elig = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (nf[j].upd == a.round - 1 && (own < nf[j].conf - 0.05f || own == 0.f)) elig |= 1u << j;
        /* (registers only: a dynamically indexed private array would be placed in LDS) */
        unsigned order = 0; int n_cand = 0;                                    /* 2 bits per rank: the direction */
        {
            unsigned left = elig;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                int bi = -1; float bc = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (((left >> j) & 1u) && (bi < 0 || nf[j].conf > bc)) { bi = j; bc = nf[j].conf; }
                if (bi >= 0) { order |= (unsigned)bi << (2 * s); ++n_cand; left &= ~(1u << bi); }
            }
        }
        auto dir_of = [&](int s) -> int { return (int)((order >> (2 * s)) & 3u); };
        auto conf_of = [&](int j) -> float { return j == 0 ? nf[0].conf : j == 1 ? nf[1].conf : j == 2 ? nf[2].conf : nf[3].conf; };
        const int mine = (int)(k ^ 1u);                                        /* my source seen from q */
        if (n_cand == 0 || dir_of(0) != mine) continue;                        /* the workgroup of q's best source does q */
        /* hypothesis of q's rank-s candidate = its source's result (rank 0: the previous round's record in hand;
         * the others: their source's frozen state) */
        auto hypothesis = [&](int s, float& hd, float& hi, float& hj, unsigned long long& hv) {
            hd = pr.depth; hi = pr.dzI; hj = pr.dzJ; hv = view_set(pr.views, pr.views_hi);
            if (s > 0) {
                const int j = dir_of(s);
                const int p = j == 0 ? nb[0] : j == 1 ? nb[1] : j == 2 ? nb[2] : nb[3];
                const bool one = j == 0 ? nf[0].one : j == 1 ? nf[1].one : j == 2 ? nf[2].one : nf[3].one;
                hd = GF((one ? job->depth1 : job->depth) + p);
                hi = GF((one ? job->dz1 : job->dz) + 2 * p); hj = GF((one ? job->dz1 : job->dz) + 2 * p + 1);
                hv = load_view_set<NV>(job, one, p);
            }
        };
        auto attempt = [&](int s, PatchResult& r, unsigned& ce, unsigned& cp) {
            float hd, hi, hj; unsigned long long hv;
            hypothesis(s, hd, hi, hj, hv);
            if (!lut_ready) {
                for (int i = lane; i < 256; i += WAVE) g_lut[i] = a.lut[i];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                lut_ready = true;
            }
            ce = 0; cp = 0;
            optimize_patch<LL>(job, a.st, a.views, qx, qy, hd, hi, hj, hv, lane, r, ce, cp, err, a.counters);
            /* counters of optimize_patch are per view slot (row leaders): bring them to lane 0 */
            ce = LL::rows_to_lane0(ce); cp = LL::rows_to_lane0(cp);
        };
        float best = own;
        bool accepted = false;
        PatchResult fin;
        fin.conf = 0.f; fin.depth = fin.dzI = fin.dzJ = fin.nx = fin.ny = fin.nz = 0.f; fin.views = 0xFFFFFFFFu; fin.views_hi = 0xFFFFFFFFu; fin.iters = 0;
        unsigned done = 0;
        if (SPEC) {
            const bool active = wave < n_cand;
            if (!active && single) return;
            if (active) {
                PatchResult r; unsigned ce, cp;
                attempt(wave, r, ce, cp);
                if (lane == 0) { g_tail_res[wave].r = r; g_tail_res[wave].n_eval = ce; g_tail_res[wave].n_pass = cp; }
            }
            if (n_cand > 1 || !single) __syncthreads();
            if (wave == 0) {
                /* the reference's sequential rule over q's candidates */
                for (int s = 0; s < n_cand; ++s) {
                    const float bc = conf_of(dir_of(s));
                    if (best > bc) break;                                      /* dmrecon.cc:371 (and every later one) */
                    done |= 1u << dir_of(s);
                    const PatchResult c = g_tail_res[s].r;
                    if (lane == 0) { n_eval += g_tail_res[s].n_eval; n_pass += g_tail_res[s].n_pass; ++n_patch; }   /* attempts the reference makes */
                    if (c.conf > 0.f && best < c.conf) { best = c.conf; accepted = true; fin = c; }   /* dmrecon.cc:378,391 */
                }
            }
        } else {
            /* one wavefront, the candidates one after the other exactly as the reference pops them: nothing
             * speculative -- the form for rounds large enough to fill the GPU anyway */
            for (int s = 0; s < n_cand; ++s) {
                const float bc = conf_of(dir_of(s));
                if (best > bc) break;                                          /* dmrecon.cc:371 (and every later one) */
                done |= 1u << dir_of(s);
                PatchResult c; unsigned ce, cp;
                attempt(s, c, ce, cp);
                if (lane == 0) { n_eval += ce; n_pass += cp; ++n_patch; }
                if (c.conf > 0.f && best < c.conf) { best = c.conf; accepted = true; fin = c; }       /* dmrecon.cc:378,391 */
            }
        }
        if (wave == 0) {
            if (accepted && lane == 0) {
                const unsigned e = atomicAdd(&t.round_work[a.round], 1u);
                DevEntry we; we.job = src.job; we.xy = qx | (qy << 16);
                const_cast<DevEntry*>(a.work)[e] = we;
                DevResult o;
                o.conf = fin.conf; o.depth = fin.depth; o.dzI = fin.dzI; o.dzJ = fin.dzJ;
                o.nx = fin.nx; o.ny = fin.ny; o.nz = fin.nz; o.views = fin.views; o.views_hi = fin.views_hi; o.iters = fin.iters;
                o.accepted = 1; o.tried = done;
                a.results[e] = o;
                const bool one = me.one;                                       /* slot holding the old state */
                float* dp = one ? job->depth : job->depth1; float* zp = one ? job->dz : job->dz1;
                float* cq = one ? job->conf : job->conf1; float* np = one ? job->normal : job->normal1;
                uint32_t* vp = one ? job->views : job->views1; int32_t* up = one ? job->upd : job->upd1;
                dp[q] = fin.depth; zp[2 * q] = fin.dzI; zp[2 * q + 1] = fin.dzJ;
                np[3 * q] = fin.nx; np[3 * q + 1] = fin.ny; np[3 * q + 2] = fin.nz;
                cq[q] = fin.conf; vp[q] = fin.views; up[q] = a.round;
                if (NV == 8) (one ? job->views_hi : job->views1_hi)[q] = fin.views_hi;
                if (own <= 0.f) { ++n_filled; atomicAdd(const_cast<uint32_t*>(&job->n_filled), 1u); }
            }
        }
        if (SPEC && !single) __syncthreads();                                  /* g_tail_res is reused by the next candidate */
    }
    /* n_eval / n_pass / n_patch / n_filled were kept by lane 0 of wavefront 0 only */
    if (threadIdx.x == 0) {
        if (n_eval) { atomicAdd(&a.counters->n_eval, (unsigned long long)n_eval); atomicAdd(&a.counters->k_eval[MI_KIND_TAIL], (unsigned long long)n_eval); }
        if (n_pass) { atomicAdd(&a.counters->n_pass, (unsigned long long)n_pass); atomicAdd(&a.counters->k_pass[MI_KIND_TAIL], (unsigned long long)n_pass);
                      atomicAdd(&a.counters->k_pass_exec[MI_KIND_TAIL], (unsigned long long)n_pass); }
        if (n_patch) { atomicAdd(&a.counters->n_patch, (unsigned long long)n_patch); atomicAdd(&a.counters->k_patch[MI_KIND_TAIL], (unsigned long long)n_patch); }
        if (n_filled) atomicAdd(&a.counters->n_filled, (unsigned long long)n_filled);
    }
    for (int off = 32; off > 0; off >>= 1) err |= __shfl_down(err, off);
    if (lane == 0 && err) atomicOr(&a.counters->error_flags, err);
}



/*
 * k_front: the END of the propagation tail, one persistent workgroup per reference view.
 *
 * Once the fronts are down to a few pixels per view, a round of k_tail is one cold patch optimisation long: every
 * launch starts with empty caches and walks the chain previous record -> job -> pixel states -> hypothesis -> master
 * window -> view records -> level -> texels again (~30 us for five entries, ~500 such rounds per call).  The reference
 * views of a call never interact (each mvs::DMRecon owns its maps, dmrecon.cc:89-172), so here every view gets ONE
 * workgroup that stays: it keeps the view's lists to itself (written and read by the same CU: its L1 / the XCD's L2,
 * ordered by workgroup barriers -- no agent-scope traffic, no tickets, no co-residency assumption), runs its rounds
 * at its own pace -- no view waits for another view's round -- and ends when its front is empty.  The rounds
 * themselves are k_tail's: same candidates, same order of trial, same acceptance rule, same two-slot pixel state,
 * same patch code (optimize_patch<16>), so the maps are bit-identical to one k_tail launch per round
 * (tests/test_gpu_parity.py::test_front_kernel_equals_one_launch_per_round); only the round NUMBERS (stamps) of a
 * view now count that view's own rounds.
 *
 * A round of a view:
 *   1. every (entry of the previous round, direction) pair is looked at by one LANE: frozen state of the neighbour q
 *      and of q's four neighbours, push rule (dmrecon.cc:400-431), q's candidates in the reference's order of trial;
 *      the lane of q's best source files q (FQ) -- exactly one lane does;
 *   2. the attempts (q, rank) run one per WAVEFRONT, taken from an LDS counter: all of them at once when they fit the
 *      workgroup (speculative, as k_tail<SPEC>), else rank by rank -- a later rank only if the reference's sequential
 *      rule still asks for it (dmrecon.cc:371);
 *   3. one lane per q applies that rule to the results (dmrecon.cc:378,391), appends q to the view's next list and
 *      writes its other state slot.
 * Rounds with more than MI_FRONT_QCAP candidate pairs are processed in chunks (all chunks read the frozen state).
 */
#ifndef MI_FRONT_WAVES
#define MI_FRONT_WAVES 8          /* wavefronts of a front workgroup = patch optimisations of a view in flight */
#endif
#define MI_FRONT_QCAP 256
#define MI_FRONT_GRAN 12          /* 8-byte granules {word, pass tag} of one attempt's result in a team's mailbox */
static_assert(MI_FRONT_WAVES <= MI_LAT_SLOTS, "the latency layout's LDS holds one patch per wavefront of a front workgroup");
struct FrontArgs {
    OptArgs o;                    /* jobs, views, lut, st, counters; o.round = the first round to run */
    DevEntry* work[2];            /* per-view lists (view j's at job_off[j]); [0] holds the lists of round o.round - 1 */
    DevResult* results[2];
    const unsigned* job_off;      /* [n_jobs] */
    const unsigned* job_count;    /* [n_jobs] entries of view j in work[0] (a first launch: the view starts at o.round from there) */
    /* ... or, a launch that CONTINUES an earlier one: where every view goes on -- round << 32 | entries << 1 | list
     * buffer (MI_FRONT_DONE: nothing left to do); null in a first launch */
    const unsigned long long* job_start;
    /* [n_jobs], zeroed before a first launch: how far the view has got, same packing, raised with atomicMax: a view that
     * ran to its end says MI_FRONT_DONE; a team that gave up (a member did not show up in time: the GPU is shared with
     * something that holds its compute units) leaves the round to go on from -- every round up to it is complete in
     * memory, and a round can be run again from its start (it only writes state slots that hold nothing older rounds
     * read, and the same values) */
    unsigned long long* job_resume;
    unsigned* job_stats;          /* [n_jobs][4]: rounds run, attempts run, sum of list sizes, 100 MHz ticks (added to) */
    int max_rounds;               /* a view stops here (int32 stamps would last; a guard against an endless front) */
    /* TEAM: the workgroups of a view's team and what each block of the grid is -- job | member << 16 | team size << 24,
     * 0xFFFFFFFF for a block without work -- decided by the host: the members of a team are blocks with the same
     * b % n_xcd, and the views whose lists are longest get the teams of the XCDs that hold fewer views */
    int team;                     /* (the smallest team of the launch; 1 = no teams) */
    const unsigned* block_map;
    unsigned long long* mail;     /* [n_jobs][2][MI_FRONT_QCAP * 4][MI_FRONT_GRAN] */
    unsigned* team_flags;         /* [n_jobs][MI_FRONT_TEAM_MAX], zeroed before the launch */
    unsigned* team_filled;        /* [n_jobs], zeroed before the launch: pixels newly filled by the view's team (atomicMax of each
                                   * member's own running count: they all count the same), added to the job by k_front_commit */
    unsigned spin_ticks;          /* a member waits this long (100 MHz ticks) for the others at an exchange, then the team gives up */
    int n_jobs;                   /* TEAM: views of the launch (the grid is padded to whole XCD rows) */
    int n_xcd;                    /* TEAM: XCDs the blocks are dealt over (block b runs on XCD b % n_xcd: observed, not promised -- checked) */
    unsigned* host_done;          /* page-locked HOST memory, [n_jobs] zeroed, or null: a view that has run to its end says so here, after its
                                   * state has been written back from the cache to memory -- the host can flatten and copy its maps while
                                   * the other views still run */
    int l2_exchange;              /* TEAM: a team on one XCD exchanges through its L2 (0: always through memory) */
    int force_write_through;      /* test hook (MI_DMRECON_DEBUG_TEAM_WT): as if a team's members had been found on different XCDs */
    int fault_member, fault_round; /* test hook (MI_DMRECON_DEBUG_FRONT_FAULT): this member of every team vanishes at that round of its
                                    * view (0 = it never shows up), as one that is not given a compute unit would; -1 = none */
    const unsigned* job_order;    /* one workgroup per view: the view block b runs (null: view b).  A batch with more views than
                                   * the GPU holds front workgroups (one per CU: 256 registers x 8 wavefronts) runs them in waves, and
                                   * the launch lasts as long as the last workgroup: the host puts the views with the most left to
                                   * fill FIRST (longest processing time first), so that the long fronts start at once and the short
                                   * ones fill the CUs that become free */
};
#define MI_FRONT_DONE 0xFFFFFFFF00000000ull
/* a team member's flag word: bits 0..29 the number of its last finished pass (only ever grows), bit 30 "my team gives up"
 * (a member did not show up), bit 31 "the view has ended" (footprint exception / cancel); a member that leaves early
 * leaves all epoch bits set, so that nobody waits for it */
#define MI_FLAG_EPOCH 0x3FFFFFFFu
#define MI_FLAG_GAVE_UP 0x40000000u
#define MI_FLAG_ENDED 0x80000000u
struct FQ {                       /* a pixel this round may rewrite */
    int xy, src;                  /* qx | qy << 16; (unused) index of its best source in the previous list */
    float own;                    /* its frozen confidence */
    unsigned info;                /* 0..7: direction of the rank-s source (2 bits each), 8..10: candidates, 11: slot of my frozen
                                   * state, 12..15: slot of my neighbours' frozen state */
    float nconf[4];               /* frozen confidence of my four neighbours */
    float hd, hi, hj; unsigned hv, hv_hi;   /* the rank-0 hypothesis: the best source's result of the previous round */
    float best; int fin;          /* sequential rule so far: best confidence, rank of the accepted result (-1: none) */
    unsigned done; int next;      /* directions consumed; next rank to consume (= candidates: finished) */
};
struct FR { PatchResult r; unsigned n_eval, n_pass; int ready; };
static_assert(sizeof(PatchResult) == 40 && offsetof(FR, n_eval) == 40 && offsetof(FR, n_pass) == 44 && offsetof(FR, ready) == 48,
              "a team's mailbox carries the first MI_FRONT_GRAN words of an FR");
__shared__ FQ g_fq[MI_FRONT_QCAP];
__shared__ FR g_fr[MI_FRONT_QCAP][4];
__shared__ unsigned g_fatt[MI_FRONT_QCAP * 4];
__shared__ unsigned g_fcnt[8];    /* 2: attempts taken, 5: newly filled pixels of the round, 6: the view's flags, 7: team: abort */
__shared__ unsigned g_ftmp[4];

/* exclusive prefix sum of v over the threads tid < MI_FRONT_QCAP (the first wavefronts; the others pass 0) in thread order,
 * the sum of all in `total` -- called by the WHOLE workgroup (two barriers inside).  Numbering by prefix instead of by
 * atomic counter: the workgroups of a team number pixels, attempts and list entries identically. */
__device__ __forceinline__ unsigned front_scan(unsigned v, int tid, unsigned& total) {
    constexpr int NW = MI_FRONT_QCAP / WAVE;
    static_assert(NW >= 1 && NW <= 4 && NW <= MI_FRONT_WAVES && MI_FRONT_QCAP % WAVE == 0, "front_scan: g_ftmp holds four wavefront sums");
    const int lane = tid & (WAVE - 1), wave = tid >> 6;
    unsigned x = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) { const unsigned y = __shfl_up(x, d); if (lane >= d) x += y; }
    if (lane == WAVE - 1 && wave < NW) g_ftmp[wave] = x;
    __syncthreads();
    unsigned before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { const unsigned s = g_ftmp[w]; all += s; if (w < wave) before += s; }
    total = all;
    __syncthreads();
    return x - v + before;
}

template <int NV, bool TEAM>
#ifndef MI_FRONT_SOLO_OCC
#define MI_FRONT_SOLO_OCC (MI_FRONT_WAVES >= 8 ? MI_FRONT_WAVES / 4 : 2)
#endif
__global__ __launch_bounds__(MI_FRONT_WAVES * WAVE) __attribute__((amdgpu_waves_per_eu((TEAM ? (MI_FRONT_WAVES >= 8 ? MI_FRONT_WAVES / 4 : 1) : MI_FRONT_SOLO_OCC), (TEAM ? (MI_FRONT_WAVES >= 8 ? MI_FRONT_WAVES / 4 : 2) : MI_FRONT_SOLO_OCC)))) void k_front(FrontArgs t) {
    typedef typename LatLay<NV>::type LL;
    typedef __attribute__((address_space(1))) unsigned long long* gmail_t;
    typedef __attribute__((address_space(1))) unsigned* gflag_t;
    const OptArgs& a = t.o;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    /* TEAM: the members of a view's team are blocks with the same b % n_xcd -- in practice the same XCD, i.e. ONE L2: what
     * the members write (all of them every word, plain stores) and read back then lives in one coherent cache.  Placement is
     * not promised: every member registers its XCC id, and a team found on several XCDs writes through instead (below). */
    int jobi = (int)blockIdx.x, member = 0, T = 1;
    if (!TEAM && t.job_order) jobi = (int)t.job_order[blockIdx.x];
    if (TEAM) {
        const unsigned m = t.block_map[blockIdx.x];
        if (m == 0xFFFFFFFFu) return;
        jobi = (int)(m & 0xFFFFu); member = (int)((m >> 16) & 0xFFu); T = (int)(m >> 24);
        if (jobi >= t.n_jobs || T < 1 || member >= T) return;
    }
    const DevJob* job = a.jobs + jobi;
    unsigned n_prev; int cur, round;
    if (t.job_start) {
        const unsigned long long s0 = t.job_start[jobi];
        if (s0 >= MI_FRONT_DONE) return;
        round = (int)(s0 >> 32); n_prev = (unsigned)(s0 & 0xFFFFFFFFull) >> 1; cur = (int)(s0 & 1ull);
    } else { round = a.round; n_prev = t.job_count[jobi]; cur = 0; }
    if (n_prev == 0) {
        if (tid == 0) atomicMax(t.job_resume + jobi, MI_FRONT_DONE);
        return;
    }
    gflag_t fl = TEAM ? (gflag_t)(t.team_flags + (size_t)jobi * MI_FRONT_FLAG_STRIDE) : nullptr;
    bool write_through = false;                    /* TEAM: set after the first exchange (uniform over the team) */
    bool via_l2 = false;                           /* TEAM: the exchanges go through the one L2 of the team's XCD (from the second on) */
    if (TEAM) {
        if (tid == 0) {
            /* my XCC id into the team's registration word; a second id there marks the team as spread over several XCDs */
            const unsigned me = (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) + 1u;     /* HW_REG_XCC_ID, bits 0..3 */
            const unsigned old = atomicCAS(t.team_flags + (size_t)jobi * MI_FRONT_FLAG_STRIDE + MI_FRONT_TEAM_MAX, 0u, me);
            if ((old != 0u && old != me) || t.force_write_through) atomicExch(t.team_flags + (size_t)jobi * MI_FRONT_FLAG_STRIDE + MI_FRONT_TEAM_MAX + 1, 1u);
        }
        /* a member that only starts when the others have given up (it found no compute unit in time) leaves at once */
        unsigned f = tid < T ? __hip_atomic_load(fl + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        if (__syncthreads_or((f & MI_FLAG_GAVE_UP) != 0)) {
            if (tid == 0) __hip_atomic_store(fl + member, MI_FLAG_EPOCH | MI_FLAG_GAVE_UP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
    for (int i = tid; i < 256; i += MI_FRONT_WAVES * WAVE) g_lut[i] = a.lut[i];
    const unsigned long long t0 = wall_clock64();
    const unsigned off = t.job_off[jobi];
    const int W = job->w, H = job->h;
    unsigned n_eval = 0, n_pass = 0, n_patch = 0, n_filled = 0, err = 0;   /* per lane; summed at the end */
    unsigned st_rounds = 0, st_att = 0, st_list = 0;
    unsigned epoch = 0;                                                     /* TEAM: exchanges so far (the same in every member) */
    unsigned team_filled = 0;                                               /* TEAM: pixels the view's team has newly filled so far */
    bool abort = false, gave_up = false;
    if (tid == 0) g_fcnt[7] = 0;
    __syncthreads();
    for (; n_prev != 0 && round < t.max_rounds && !abort; ++round) {
        /* a footprint exception (patch_sampler.cc:78-82) or the host's cancel ends the view: one lane looks, all agree
         * (the members of a team see the flag at different times: they tell each other at the next exchange) */
        if (TEAM && member == t.fault_member && (int)st_rounds >= t.fault_round) return;
        /* (looked at every eighth round: an agent-scope load the whole workgroup waits for costs a microsecond of a round
         * that lasts twenty; a cancelled view goes on for 0.2 ms more) */
        if ((st_rounds & 7u) == 0u) {
            if (tid == 0) g_fcnt[6] = (unsigned)__hip_atomic_load((gi32_t)&job->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
        }
        if (!TEAM && g_fcnt[6] != 0) break;
        const DevEntry* pw = t.work[cur] + off; const DevResult* prs = t.results[cur] + off;
        DevEntry* ow = t.work[cur ^ 1] + off; DevResult* ors = t.results[cur ^ 1] + off;
        if (tid == 0) g_fcnt[5] = 0;
        ++st_rounds; st_list += n_prev;
        unsigned n_next = 0;                                                /* entries of this round's list so far */
        for (unsigned base = 0; base < 4u * n_prev && !abort; base += MI_FRONT_QCAP) {
            if (tid == 0) g_fcnt[2] = 0;
            /* ---- 1. one lane per (entry, direction) */
            const unsigned cand = base + (unsigned)tid;
            bool mine = false; int my_cand = 0;
            FQ Q;
            if (tid < MI_FRONT_QCAP && cand < 4u * n_prev) {
                const unsigned ep = cand >> 2, k = cand & 3u;
                const DevResult* pr = prs + ep;
                const float pconf = GF(&pr->conf);
                const int sxy = GI(&pw[ep].xy);
                const int qx = (sxy & 0xFFFF) + (k == 0 ? -1 : k == 1 ? 1 : 0), qy = (sxy >> 16) + (k == 2 ? -1 : k == 3 ? 1 : 0);
                if (!(qx < MI_HALF || qy < MI_HALF || qx >= W - MI_HALF || qy >= H - MI_HALF)) {   /* patch_sampler.cc:47-50 */
                    const int q = qy * W + qx;
                    const int nb[4] = {q - 1, q + 1, q - W, q + W};
                    const Frozen me = frozen_state(job, q, round);
                    Frozen nf[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) nf[j] = frozen_state(job, nb[j], round);
                    const float own = me.conf;
                    if (own < pconf - 0.05f || own == 0.f) {
                        unsigned elig = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (nf[j].upd == round - 1 && (own < nf[j].conf - 0.05f || own == 0.f)) elig |= 1u << j;
                        unsigned order = 0; int n_cand = 0;
                        unsigned left = elig;
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            int bi = -1; float bc = 0.f;
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (((left >> j) & 1u) && (bi < 0 || nf[j].conf > bc)) { bi = j; bc = nf[j].conf; }
                            if (bi >= 0) { order |= (unsigned)bi << (2 * s); ++n_cand; left &= ~(1u << bi); }
                        }
                        if (n_cand > 0 && (int)(order & 3u) == (int)(k ^ 1u)) {        /* I am q's best source: q is mine */
                            mine = true; my_cand = n_cand;
                            Q.xy = qx | (qy << 16); Q.src = (int)ep; Q.own = own;
                            Q.info = order | ((unsigned)n_cand << 8) | (me.one ? 1u << 11 : 0u)
                                   | (nf[0].one ? 1u << 12 : 0u) | (nf[1].one ? 1u << 13 : 0u) | (nf[2].one ? 1u << 14 : 0u) | (nf[3].one ? 1u << 15 : 0u);
                            Q.nconf[0] = nf[0].conf; Q.nconf[1] = nf[1].conf; Q.nconf[2] = nf[2].conf; Q.nconf[3] = nf[3].conf;
                            Q.hd = GF(&pr->depth); Q.hi = GF(&pr->dzI); Q.hj = GF(&pr->dzJ); Q.hv = GU(&pr->views); Q.hv_hi = GU(&pr->views_hi);
                            Q.best = own; Q.fin = -1; Q.done = 0; Q.next = 0;
                        }
                    }
                }
            }
            unsigned tot;
            const unsigned qi0 = front_scan(mine ? (1u | ((unsigned)my_cand << 16)) : 0u, tid, tot) & 0xFFFFu;
            const unsigned nq = tot & 0xFFFFu, sum_cand = tot >> 16;
            if (mine) {
                g_fq[qi0] = Q;
                g_fr[qi0][0].ready = 0; g_fr[qi0][1].ready = 0; g_fr[qi0][2].ready = 0; g_fr[qi0][3].ready = 0;
            }
            __syncthreads();
            /* the first pass: every attempt at once if they fit the wavefronts of the view, else the best candidate of every pixel */
            const bool all_at_once = sum_cand <= (unsigned)(T * MI_FRONT_WAVES);
            unsigned natt;
            {
                const unsigned cnt = (unsigned)tid < nq ? (all_at_once ? ((g_fq[tid].info >> 8) & 7u) : 1u) : 0u;
                const unsigned pos = front_scan(cnt, tid, natt);
                for (unsigned s = 0; s < cnt; ++s) g_fatt[pos + s] = (unsigned)tid | (s << 16);
            }
            for (;;) {
                __syncthreads();
                if (natt == 0) break;
                if (TEAM) ++epoch;
                gmail_t box = TEAM ? (gmail_t)(t.mail + ((size_t)jobi * 2 + (epoch & 1u)) * (MI_FRONT_QCAP * 4) * MI_FRONT_GRAN) : nullptr;
                /* ---- 2. the attempts of this pass, one per wavefront; a team deals them out by index */
                for (;;) {
                    unsigned kk = 0;
                    if (lane == 0) kk = atomicAdd(&g_fcnt[2], 1u);
                    kk = (unsigned)__builtin_amdgcn_readfirstlane((int)kk);
                    const unsigned idx = (unsigned)member + kk * (unsigned)T;
                    if (idx >= natt) break;
                    const unsigned att = g_fatt[idx];
                    const int qi = (int)(att & 0xFFFFu), s = (int)(att >> 16);
                    const FQ& Qa = g_fq[qi];
                    const int qx = Qa.xy & 0xFFFF, qy = Qa.xy >> 16;
                    float hd = Qa.hd, hi = Qa.hi, hj = Qa.hj; unsigned long long hv = view_set(Qa.hv, Qa.hv_hi);
                    if (s > 0) {
                        /* a later candidate's hypothesis = its source's frozen state */
                        const int j = (int)((Qa.info >> (2 * s)) & 3u);
                        const int q = qy * W + qx;
                        const int p = j == 0 ? q - 1 : j == 1 ? q + 1 : j == 2 ? q - W : q + W;
                        const bool one = ((Qa.info >> (12 + j)) & 1u) != 0;
                        hd = GF((one ? job->depth1 : job->depth) + p);
                        hi = GF((one ? job->dz1 : job->dz) + 2 * p); hj = GF((one ? job->dz1 : job->dz) + 2 * p + 1);
                        hv = load_view_set<NV>(job, one, p);
                    }
                    PatchResult r; unsigned ce = 0, cp = 0;
#ifdef MI_PROBE
                    if (lane == 0) g_pidx[(tid >> 6) & 7] = 0;
                    const unsigned long long rt0 = wall_clock64();
                    TSTAMP(1);
#endif
                    optimize_patch<LL>(job, a.st, a.views, qx, qy, hd, hi, hj, hv, lane, r, ce, cp, err, a.counters);
#ifdef MI_PROBE
                    TSTAMP(2);
                    if (lane == 0 && a.tbuf) {
                        /* record: [0] view | round << 16 | entries << 32, [1] 100 MHz ticks of the attempt, [2] stamps, then the stamps */
                        const unsigned long long rec = atomicAdd(a.tbuf, 1ull);
                        if (rec < 400) {
                            unsigned long long* o = a.tbuf + 8 + rec * (MI_PROBE_LOG + 4);
                            const unsigned np = g_pidx[(tid >> 6) & 7];
                            o[0] = (unsigned long long)jobi | ((unsigned long long)(round - a.round) << 16) | ((unsigned long long)n_prev << 32) | ((unsigned long long)natt << 48);
                            o[1] = wall_clock64() - rt0; o[2] = np; o[3] = (unsigned long long)r.iters | ((unsigned long long)(r.conf > 0.f) << 32);
                            for (unsigned k = 0; k < np; ++k) o[4 + k] = g_plog[(tid >> 6) & 7][k];
                        }
                    }
#endif
                    ce = LL::rows_to_lane0(ce); cp = LL::rows_to_lane0(cp);
                    if (lane == 0) {
                        FR& o = g_fr[qi][s]; o.r = r; o.n_eval = ce; o.n_pass = cp; o.ready = 1; ++st_att;
                        if (TEAM) {
                            /* to the other members: every word with the pass's tag in one 8-byte store.  A team on ONE XCD (known
                             * after the first exchange) hands over through that XCD's L2: plain stores stay there (the L1 writes
                             * through), the readers' agent-scope loads bypass their L1 and are served from it.  Before that, and
                             * for a team on several XCDs: agent-scope stores, through to memory. */
                            const unsigned* wsrc = (const unsigned*)&o;
                            gmail_t rec = box + (size_t)idx * MI_FRONT_GRAN;
                            if (via_l2) {
#pragma unroll
                                for (int k = 0; k < MI_FRONT_GRAN; ++k) rec[k] = ((unsigned long long)epoch << 32) | wsrc[k];
                            } else {
#pragma unroll
                                for (int k = 0; k < MI_FRONT_GRAN; ++k)
                                    __hip_atomic_store(rec + k, ((unsigned long long)epoch << 32) | wsrc[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
                    }
                }
                if (TEAM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   /* my results have left before my flag does */
                __syncthreads();
                if (tid == 0) g_fcnt[2] = 0;
                if (TEAM) {
                    /* ---- the exchange: my flag = this pass (+ "my view has ended"), wait for every member's, fetch their
                     * results.  Flags only ever grow, a member is waited for until its flag has REACHED this pass (it may be
                     * a pass ahead by the time it is looked at: it cannot be two, the next exchange needs my flag), and not
                     * for ever: the wait is bounded by the wall clock */
                    if (tid == 0) {
                        const unsigned fv = epoch | (g_fcnt[6] != 0 ? MI_FLAG_ENDED : 0u);
                        if (via_l2) *(volatile unsigned*)(t.team_flags + (size_t)jobi * MI_FRONT_FLAG_STRIDE + member) = fv;
                        else __hip_atomic_store(fl + member, fv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (tid < T) {
                        unsigned f = 0; bool seen = false;
                        const unsigned long long w0 = wall_clock64();
                        for (;;) {
                            f = __hip_atomic_load(fl + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((f & MI_FLAG_EPOCH) >= epoch || (f & (MI_FLAG_GAVE_UP | MI_FLAG_ENDED))) { seen = true; break; }
                            if (wall_clock64() - w0 > (unsigned long long)t.spin_ticks) break;
                            __builtin_amdgcn_s_sleep(4);
                        }
                        if (!seen || (f & MI_FLAG_GAVE_UP)) atomicOr(&g_fcnt[7], 2u);
                        else if (f & MI_FLAG_ENDED) atomicOr(&g_fcnt[7], 1u);
                    }
                    __syncthreads();
                    if (g_fcnt[7] != 0) {
                        /* the view has ended (nothing of it is used any more), or the team gives up: this member says so
                         * to whoever still waits for it and leaves where the host can pick the view up again */
                        gave_up = (g_fcnt[7] & 2u) != 0;
                        if (tid == 0)
                            __hip_atomic_store(fl + member, MI_FLAG_EPOCH | (gave_up ? MI_FLAG_GAVE_UP : MI_FLAG_ENDED), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        abort = true; break;
                    }
                    /* everybody has registered by now (a member registers before its first flag) */
                    if (epoch == 1u) {
                        write_through = __hip_atomic_load(fl + MI_FRONT_TEAM_MAX + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
                        via_l2 = !write_through && t.l2_exchange != 0;
                    }
                    for (unsigned u = (unsigned)tid; u < natt * MI_FRONT_GRAN; u += MI_FRONT_WAVES * WAVE) {
                        const unsigned idx = u / MI_FRONT_GRAN, k = u - idx * MI_FRONT_GRAN;
                        if ((int)(idx % (unsigned)T) == member) continue;
                        const unsigned long long g = __hip_atomic_load(box + (size_t)idx * MI_FRONT_GRAN + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((unsigned)(g >> 32) != epoch) err |= 16u;          /* cannot happen: the flag came after the results */
                        const unsigned att = g_fatt[idx];
                        FR& o = g_fr[att & 0xFFFFu][att >> 16];
                        ((unsigned*)&o)[k] = (unsigned)g;
                        if (k == 0) o.ready = 1;
                    }
                }
                __syncthreads();
                /* ---- 3. the reference's sequential rule, one lane per pixel; what it still asks for is the next pass */
                bool again = false; int again_rank = 0;
                if ((unsigned)tid < nq) {
                    FQ& Qr = g_fq[tid];
                    const int n_cand = (int)((Qr.info >> 8) & 7u);
                    int next = Qr.next;
                    if (next < n_cand) {
                        float best = Qr.best; int fin = Qr.fin; unsigned done = Qr.done;
                        while (next < n_cand) {
                            const int j = (int)((Qr.info >> (2 * next)) & 3u);
                            const float bc = Qr.nconf[j];
                            if (best > bc) { next = n_cand; break; }                   /* dmrecon.cc:371 (and every later one) */
                            const FR& c = g_fr[tid][next];
                            if (!c.ready) break;
                            done |= 1u << j;
                            n_eval += c.n_eval; n_pass += c.n_pass; ++n_patch;         /* attempts the reference makes */
                            if (c.r.conf > 0.f && best < c.r.conf) { best = c.r.conf; fin = next; }   /* dmrecon.cc:378,391 */
                            ++next;
                        }
                        Qr.best = best; Qr.fin = fin; Qr.done = done; Qr.next = next;
                        if (next < n_cand) { again = true; again_rank = next; }
                    }
                }
                const unsigned pos = front_scan(again ? 1u : 0u, tid, natt);
                if (again) g_fatt[pos] = (unsigned)tid | ((unsigned)again_rank << 16);
            }
            if (abort) break;
            /* ---- the accepted pixels: this round's list, the other state slot (every member of a team writes them all:
             * each reads its own copy next round, nobody waits for another's stores) */
            const bool acc = (unsigned)tid < nq && g_fq[tid].fin >= 0;
            unsigned n_acc;
            const unsigned en = n_next + front_scan(acc ? 1u : 0u, tid, n_acc);
            n_next += n_acc;
            if (acc) {
                const FQ& Qw = g_fq[tid];
                const PatchResult fin = g_fr[tid][Qw.fin].r;
                const int qx = Qw.xy & 0xFFFF, qy = Qw.xy >> 16, q = qy * W + qx;
                DevEntry we; we.job = jobi; we.xy = Qw.xy;
                DevResult o;
                o.conf = fin.conf; o.depth = fin.depth; o.dzI = fin.dzI; o.dzJ = fin.dzJ;
                o.nx = fin.nx; o.ny = fin.ny; o.nz = fin.nz; o.views = fin.views; o.views_hi = fin.views_hi; o.iters = fin.iters;
                o.accepted = 1; o.tried = Qw.done;
                const bool one = ((Qw.info >> 11) & 1u) != 0;                       /* slot holding the old state */
                float* dp = one ? job->depth : job->depth1; float* zp = one ? job->dz : job->dz1;
                float* cq = one ? job->conf : job->conf1; float* np = one ? job->normal : job->normal1;
                uint32_t* vp = one ? job->views : job->views1; int32_t* up = one ? job->upd : job->upd1;
                /* TEAM: every member writes every word, and reads back only what it wrote itself -- through its XCD's L2.
                 * With the whole team on ONE XCD that is one coherent copy.  Members on DIFFERENT XCDs each keep a dirty copy
                 * of the line in their own L2, written back whenever that cache sees fit -- and if that is after another
                 * member (one pass ahead) has written the pixel's NEXT value and lost its own copy of the line, that member
                 * fetches the old value back from memory (seen once, before teams were kept on one XCD: one pixel of one
                 * view, with a second process thrashing the caches).  Such a team writes THROUGH its L2s (agent scope: the
                 * line is dropped, the read-back comes from memory: ~9 us per round slower, so only then). */
                auto put = [&](void* dst, unsigned v) {
                    if (TEAM && write_through) __hip_atomic_store((unsigned*)dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else *(unsigned*)dst = v;
                };
                put(&ow[en].job, (unsigned)we.job); put(&ow[en].xy, (unsigned)we.xy);
                {
                    const unsigned* src = (const unsigned*)&o; unsigned* dst = (unsigned*)&ors[en];
#pragma unroll
                    for (int k = 0; k < (int)(sizeof(DevResult) / 4); ++k) put(dst + k, src[k]);
                }
                put(dp + q, __float_as_uint(fin.depth)); put(zp + 2 * q, __float_as_uint(fin.dzI)); put(zp + 2 * q + 1, __float_as_uint(fin.dzJ));
                put(np + 3 * q, __float_as_uint(fin.nx)); put(np + 3 * q + 1, __float_as_uint(fin.ny)); put(np + 3 * q + 2, __float_as_uint(fin.nz));
                put(cq + q, __float_as_uint(fin.conf)); put(vp + q, fin.views); put(up + q, (unsigned)round);
                if (NV == 8) put((one ? job->views_hi : job->views1_hi) + q, fin.views_hi);
                if (Qw.own <= 0.f) atomicAdd(&g_fcnt[5], 1u);
            }
            __syncthreads();
        }
        if (abort) break;
        n_prev = n_next;
        /* Progress::filled.  The members of a team all count the same pixels: each raises the view's count to its own */
        team_filled += g_fcnt[5];
        if (tid == 0 && g_fcnt[5]) {
            if (TEAM) atomicMax(t.team_filled + jobi, team_filled);
            else { atomicAdd(const_cast<uint32_t*>(&job->n_filled), g_fcnt[5]); n_filled += g_fcnt[5]; }
        }
        cur ^= 1;
        __syncthreads();
    }
    /* a view that ran to its end (nothing accepted in its last round: every member's writes of the rounds before are in the
     * L2 -- they were acknowledged before the last exchange): one lane writes the cache back and tells the host */
    if (t.host_done && tid == 0 && member == 0 && n_prev == 0 && !abort) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(t.host_done + jobi, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    /* how far the view has got (see FrontArgs::job_resume): done, or the round a given-up team stopped in */
    if (tid == 0) {
        if (gave_up) {
            atomicMax(t.job_resume + jobi, ((unsigned long long)(unsigned)round << 32) | ((unsigned long long)n_prev << 1) | (unsigned long long)cur);
            atomicOr(&a.counters->error_flags, 32u);
        } else if (n_prev == 0 || abort || g_fcnt[6] != 0) atomicMax(t.job_resume + jobi, MI_FRONT_DONE);
    }
    /* counters: per lane so far (what the sequential rule counts is the same in every member of a team: the first reports) */
    if (member != 0) { n_eval = 0; n_pass = 0; n_patch = 0; }
    for (int off2 = 32; off2 > 0; off2 >>= 1) {
        n_eval += __shfl_down(n_eval, off2); n_pass += __shfl_down(n_pass, off2);
        n_patch += __shfl_down(n_patch, off2); n_filled += __shfl_down(n_filled, off2);
        err |= __shfl_down(err, off2); st_att += __shfl_down(st_att, off2);
    }
    if (lane == 0) {
        if (n_eval) { atomicAdd(&a.counters->n_eval, (unsigned long long)n_eval); atomicAdd(&a.counters->k_eval[MI_KIND_FRONT], (unsigned long long)n_eval); }
        if (n_pass) { atomicAdd(&a.counters->n_pass, (unsigned long long)n_pass); atomicAdd(&a.counters->k_pass[MI_KIND_FRONT], (unsigned long long)n_pass);
                      atomicAdd(&a.counters->k_pass_exec[MI_KIND_FRONT], (unsigned long long)n_pass); }
        if (n_patch) { atomicAdd(&a.counters->n_patch, (unsigned long long)n_patch); atomicAdd(&a.counters->k_patch[MI_KIND_FRONT], (unsigned long long)n_patch); }
        if (n_filled) atomicAdd(&a.counters->n_filled, (unsigned long long)n_filled);
        if (err) atomicOr(&a.counters->error_flags, err);
        if (st_att) atomicAdd(&t.job_stats[4 * jobi + 1], st_att);
    }
    if (tid == 0 && member == 0) {
        atomicAdd(&t.job_stats[4 * jobi], st_rounds - (gave_up ? 1u : 0u)); atomicAdd(&t.job_stats[4 * jobi + 2], st_list - (gave_up ? n_prev : 0u));
        atomicAdd(&t.job_stats[4 * jobi + 3], (unsigned)(wall_clock64() - t0));
        if (n_prev != 0 && !abort && round >= t.max_rounds) atomicOr(&a.counters->error_flags, 8u);    /* the front did not end */
    }
}

/* The accepted entries of the last k_tail round (all views mixed) dealt out to the views' own lists for k_front:
 * view j's entries at job_off[j] of the output buffers, their number in job_count[j] (zeroed before). */
struct FrontSplitArgs {
    const DevEntry* work; const DevResult* results; const unsigned* n_ptr;
    DevEntry* owork; DevResult* oresults; const unsigned* job_off; unsigned* job_count;
};
__global__ __launch_bounds__(256) void k_front_split(FrontSplitArgs a) {
    const unsigned n = *a.n_ptr;
    for (unsigned e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        const DevResult r = a.results[e];
        if (!r.accepted) continue;
        const DevEntry w = a.work[e];
        const unsigned i = a.job_off[w.job] + atomicAdd(&a.job_count[w.job], 1u);
        a.owork[i] = w; a.oresults[i] = r;
    }
}

/* After a team launch: the pixels the teams have newly filled (FrontArgs::team_filled) go to their jobs' Progress::filled
 * and to the call's counter; one lane per view. */
__global__ __launch_bounds__(256) void k_front_commit(const DevJob* jobs, unsigned* team_filled, DevCounters* counters, int n_jobs) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_jobs) return;
    const unsigned n = team_filled[j];
    if (!n) return;
    team_filled[j] = 0;
    atomicAdd(const_cast<uint32_t*>(&jobs[j].n_filled), n);
    atomicAdd(&counters->n_filled, (unsigned long long)n);
}

/* Fold the second state slot back into the first where it is the newer one (after the last tail round). */
struct FlattenArgs {
    float* depth; float* dz; float* conf; float* normal; uint32_t* views; int32_t* upd;
    const float* depth1; const float* dz1; const float* conf1; const float* normal1; const uint32_t* views1; const int32_t* upd1;
    uint32_t* views_hi; const uint32_t* views1_hi;      /* null unless nrReconNeighbors > 4 */
    unsigned n;
};
__global__ __launch_bounds__(256) void k_flatten(FlattenArgs a) {
    const unsigned p = blockIdx.x * 256 + threadIdx.x;
    if (p >= a.n) return;
    const int s1 = a.upd1[p];
    if (s1 <= a.upd[p]) return;
    a.depth[p] = a.depth1[p]; a.dz[2 * p] = a.dz1[2 * p]; a.dz[2 * p + 1] = a.dz1[2 * p + 1];
    a.normal[3 * p] = a.normal1[3 * p]; a.normal[3 * p + 1] = a.normal1[3 * p + 1]; a.normal[3 * p + 2] = a.normal1[3 * p + 2];
    a.conf[p] = a.conf1[p]; a.views[p] = a.views1[p]; a.upd[p] = s1;
    if (a.views_hi) a.views_hi[p] = a.views1_hi[p];
}

/*
 * The pixels of one view that were written from round r0 on, as records {view, pixel, depth, conf, dzI, dzJ [, nx, ny, nz]}
 * appended to ONE list in page-locked host memory (all views of a batch; one atomic per wavefront).  What it is for
 * (BatchRun::front_rounds): the maps of a large batch are copied to the caller while the front kernel still runs -- a
 * snapshot of the state at the hand-over, 16 bytes per pixel over PCIe, hidden behind the kernel -- and what the front
 * changed afterwards (a few per cent of the pixels) follows as this list, which the host writes over the snapshot.  A pixel's
 * state is the slot with the larger stamp (k_flatten).  Records beyond the list's capacity are dropped: the host sees the
 * count and copies the maps in full instead.
 */
struct EmitArgs {
    const float* depth; const float* dz; const float* conf; const float* normal; const int32_t* upd;
    const float* depth1; const float* dz1; const float* conf1; const float* normal1; const int32_t* upd1;
    unsigned n; int r0; unsigned view; unsigned stride;     /* stride: words per record, 6 or 9 (with the normal) */
    unsigned* count; unsigned cap; uint32_t* out;
};
__global__ __launch_bounds__(256) void k_emit_changed(EmitArgs a) {
    const unsigned p = blockIdx.x * 256 + threadIdx.x;
    int s0 = -1, s1 = -1;
    if (p < a.n) { s0 = a.upd[p]; s1 = a.upd1[p]; }
    const bool mine = p < a.n && (s0 >= a.r0 || s1 >= a.r0);
    const unsigned long long b = __ballot(mine);
    if (b == 0) return;
    const int lane = (int)(threadIdx.x & 63u);
    unsigned base = 0;
    if (lane == __ffsll((long long)b) - 1) base = atomicAdd(a.count, (unsigned)__popcll(b));
    base = (unsigned)__shfl((int)base, __ffsll((long long)b) - 1);
    if (!mine) return;
    const unsigned k = base + (unsigned)__popcll(b & ((1ull << lane) - 1ull));
    if (k >= a.cap) return;
    const bool one = s1 > s0;
    uint32_t* o = a.out + (size_t)k * a.stride;
    o[0] = a.view; o[1] = p;
    o[2] = __float_as_uint(one ? a.depth1[p] : a.depth[p]);
    o[3] = __float_as_uint(one ? a.conf1[p] : a.conf[p]);
    o[4] = __float_as_uint(one ? a.dz1[2 * p] : a.dz[2 * p]);
    o[5] = __float_as_uint(one ? a.dz1[2 * p + 1] : a.dz[2 * p + 1]);
    if (a.stride >= 9) {
        o[6] = __float_as_uint(one ? a.normal1[3 * p] : a.normal[3 * p]);
        o[7] = __float_as_uint(one ? a.normal1[3 * p + 1] : a.normal[3 * p + 1]);
        o[8] = __float_as_uint(one ? a.normal1[3 * p + 2] : a.normal[3 * p + 2]);
    }
}

/* Parity hook: one hypothesis against every global view; one quad lane per 4 views. */
struct EvalArgs {
    const DevJob* job; const DevView* views; const float* lut; DevSettings st;
    int x, y; float depth, dzI, dzJ;
    float* master; float* ncc; int32_t* ok; float* col; float* deriv; int32_t* level;
};
__global__ __launch_bounds__(WAVE) void k_patch_eval(EvalArgs a) {
    float* s_lut = g_lut;
    float* s_geo = g_geo[0];
    float* s_mcol = g_mcol[0];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += WAVE) s_lut[i] = a.lut[i];
    __syncthreads();
    /* reuse optimize_patch's setup by replicating its prologue on quad 0 only */
    const DevJob* job = a.job;
    PatchState ps; ps.job = job; ps.x = a.x; ps.y = a.y; ps.n_eval = ps.n_pass = 0; ps.sel = -1; ps.counters = nullptr;
    ps.jinv0 = job->inv0_s;
    ps.inrm_c = pixel_scale(job, a.x, a.y);
    if (lane == 0) for (int k = 0; k < 5; ++k) a.master[k] = 0.f;
    if (a.x - MI_HALF < 0 || a.y - MI_HALF < 0 || a.x + MI_HALF > job->w - 1 || a.y + MI_HALF > job->h - 1) return;
    const DevView* RV = a.views + job->ref_view;
    const DevLevel& RL = RV->lv[job->scale];
    const uint32_t* rimg = RV->img + RL.tex_off;
    for (int i = lane; i < MI_NS; i += WAVE) {
        const int dj = i / MI_FW - MI_HALF, di = i - (i / MI_FW) * MI_FW - MI_HALF;
        s_geo[i] = pixel_scale(job, a.x + di, a.y + dj);
        const uint32_t t = rimg[(size_t)(a.y + dj) * RL.w + (a.x + di)];
        s_mcol[3 * i] = s_lut[t & 255u]; s_mcol[3 * i + 1] = s_lut[(t >> 8) & 255u]; s_mcol[3 * i + 2] = s_lut[(t >> 16) & 255u];
    }
    __syncthreads();
    float mm = 0.f;
    for (int k = 0; k < 3 * MI_NS; ++k) mm += s_mcol[k];
    mm /= 3.f * (float)MI_NS;
    if (mm < 0.01f || mm > 0.99f) return;
    __syncthreads();
    for (int i = lane; i < 3 * MI_NS; i += WAVE) s_mcol[i] /= mm;
    __syncthreads();
    float x0 = 0.f, x1 = 0.f, x2 = 0.f;
    for (int i = 0; i < MI_NS; ++i) { x0 += s_mcol[3 * i]; x1 += s_mcol[3 * i + 1]; x2 += s_mcol[3 * i + 2]; }
    x0 /= (float)MI_NS; x1 /= (float)MI_NS; x2 /= (float)MI_NS;
    float sd = 0.f;
    for (int i = 0; i < MI_NS; ++i) {
        const float aa = s_mcol[3 * i] - x0, b = s_mcol[3 * i + 1] - x1, c = s_mcol[3 * i + 2] - x2;
        sd += aa * aa + b * b + c * c;
    }
    ps.mmean = mm; ps.xbar0 = x0; ps.xbar1 = x1; ps.xbar2 = x2; ps.sqrDevX = sd;
    ps.cs0 = ps.cs1 = ps.cs2 = 1.f / mm;
    if (!set_state(ps, a.depth, a.dzI, a.dzJ)) return;
    if (lane == 0) {
        a.master[0] = 1.f; a.master[1] = mm;
        float nx, ny, nz, cx, cy, cz;
        patch_normal(ps, nx, ny, nz, cx, cy, cz);
        a.master[2] = nx; a.master[3] = ny; a.master[4] = nz;
    }
    if (lane < job->n_global) {
        const int g = lane;
        ColorSums S; bool okc;
        const float ncc = eval_color<Lay<1, 4> >(ps, a.views, g, s_lut, s_geo, s_mcol, S, okc, true, 0);
        a.ncc[g] = ncc;
        NView nv; int level = -1; GNSums gn;
        bool okd = setup_view(a.views, job->gv[g], ps, nv, level)
            && sample_pass<PASS_DUMP, Lay<1, 4> >(ps, nv, s_lut, s_geo, s_mcol, S, gn, a.col + g * 3 * MI_NS, a.deriv + g * 3 * MI_NS, 0);
        a.ok[g] = okd ? 1 : 0;
        a.level[g] = level;
    }
}

/* ------------------------------------------------------------------------- */
/* Sweep kernels: one lane per reference-image pixel.                          */

struct SweepArgs {
    const DevJob* jobs;
    DevEntry* work;        /* this round's list of the views in the throughput layout ... */
    DevEntry* work_lat;    /* ... and of the views that have handed over to the latency layout */
    unsigned* round_work;      /* [round] = size of the throughput list of that round (zeroed before the call) */
    unsigned* round_work_lat;  /* [round] = size of the latency list */
    unsigned* view_count;  /* [3][n_jobs]: entries per view of round r at [(r % 3) * n_jobs + view] */
    unsigned* view_mode;   /* [n_jobs]: 0 = throughput layout, else the round from which the view is in the latency layout */
    unsigned handover;     /* a view hands over once a round's list of ITS OWN is shorter than this */
    int n_jobs;
    int round;
    /* speculative rounds (k_optimize_spec): besides the entries of the throughput list, one ITEM per (entry, candidate
     * hypothesis) pair -- entry index << 2 | rank -- so that every attempt of the round gets a quad of its own; null: none */
    unsigned* items;
    unsigned* round_items; /* [round] = number of items */
    int self_round;        /* 1: the seed re-optimisation round -- the entries are the pixels written in the round before THEMSELVES
                            * (DevSettings::self_round), one candidate each */
};

/* Which pixels must be (re)optimised this round: the push rule of dmrecon.cc:400-431 as a pull.
 * One lane per pixel, 8 pixels per lane; the block's hits are compacted through ballots + one LDS
 * prefix so that the global work-list counter sees ONE atomic per 2048 pixels (a per-wave atomic on a
 * single word costs ~10 ns each and dominated this kernel at 40 000 waves).
 *
 * The lane layout of a view's patch optimisations is the VIEW's own affair: a view leaves the throughput layout for
 * good in the round after the first one whose list -- its own entries, not the batch's -- had fewer than `handover`
 * entries.  The two layouts sum a pass's 25 samples in different orders (1e-7), so whatever decides the layout decides
 * the last bits of the maps: decided per view from the view's own history, a view's maps do not depend on what it was
 * batched with (other reference views, merged calls, GPU slots).  Decided here, on the device, from the counts the
 * previous round left: the host can enqueue rounds without reading anything back. */
#define GEN_PER_THREAD 8
/* A workgroup scans a MI_GEN_TILE_W x MI_GEN_TILE_H pixel tile; a wavefront takes a 64 x 8 strip of it as eight
 * 8 x 8 sub-tiles (one per trip, lane = pixel of the sub-tile).  The ballot-compacted entries therefore come out
 * sub-tile by sub-tile: the 16 consecutive entries that form a wavefront of k_optimize<1> lie within a few pixels
 * of each other, their sample windows overlap in every neighbour view, and the texel gathers of a wavefront touch
 * a fraction of the cache lines that 16 hits strung along an image row do (the propagation front crosses a row at
 * isolated pixels).  The order of the list has no influence on the results. */
__global__ __launch_bounds__(256) void k_generate(SweepArgs a) {
    __shared__ unsigned s_wave_cnt[4];
    __shared__ unsigned s_base, s_items, s_ibase;
    const int jobi = blockIdx.y;
    const DevJob* job = a.jobs + jobi;
    /* the view's layout this round (every workgroup of the view computes the same from last round's count; the first
     * one records the hand-over and clears the count slot of the next round) */
    const unsigned prev_cnt = a.view_count[(unsigned)((a.round + 2) % 3) * (unsigned)a.n_jobs + (unsigned)jobi];
    const bool lat = a.view_mode[jobi] != 0 || prev_cnt < a.handover;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (lat && a.view_mode[jobi] == 0) a.view_mode[jobi] = (unsigned)a.round;
        a.view_count[(unsigned)((a.round + 1) % 3) * (unsigned)a.n_jobs + (unsigned)jobi] = 0;
    }
    if (job->flags != 0) return;                 /* failed / cancelled view */
    /* a view whose list was empty last round wrote nothing then: its propagation is over, its pixels need not be read again
     * (the views of a large batch end hundreds of rounds apart) */
    if (a.round >= 2 && prev_cnt == 0u) return;
    const int W = job->w, H = job->h;
    const int tiles_x = (W + MI_GEN_TILE_W - 1) / MI_GEN_TILE_W, tiles_y = (H + MI_GEN_TILE_H - 1) / MI_GEN_TILE_H;
    if ((int)blockIdx.x >= tiles_x * tiles_y) return;
    const int tcol = (int)blockIdx.x % tiles_x, trow = (int)blockIdx.x / tiles_x;
    const int tx0 = tcol * MI_GEN_TILE_W, ty0 = trow * MI_GEN_TILE_H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lx = lane & 7, ly = lane >> 3;
    /* The stamps of the tile and a one-pixel rim, once, through LDS: a row of 66 stamps per load instruction of the workgroup --
     * nine coalesced loads per lane instead of four scattered ones per pixel (32 per lane): the kernel is bound by the NUMBER of
     * its load instructions (0.48 of a step's 12 ms; a bit per pixel in front of the same four loads made it slower,
     * profiles/r6_ab_experiments.txt M), and the own confidence is only read where a neighbour's stamp says so.  Outside the
     * image: -1, the stamp of a pixel never written. */
    constexpr int SW = MI_GEN_TILE_W + 2, SH = MI_GEN_TILE_H + 2;
    __shared__ int s_upd[SH * SW];
    static_assert(SW == 66 && SH * SW <= 65536 / 993 * 66, "i / 66 below is (i * 993) >> 16: exact for i < 4356");
#ifndef MI_GEN_GLOBAL_STAMPS           /* (-DMI_GEN_GLOBAL_STAMPS: the stamps straight from memory, as until round 6 -- the A/B build) */
    for (int i = (int)threadIdx.x; i < SH * SW; i += 256) {
        const int r = (i * 993) >> 16, c = i - r * SW;
        const int y = ty0 - 1 + r, x = tx0 - 1 + c;
        s_upd[i] = (y >= 0 && y < H && x >= 0 && x < W) ? job->upd[(size_t)y * W + x] : -1;
    }
    __syncthreads();
#define MI_STAMP(si, gi) s_upd[si]
#else
    (void)s_upd;
#define MI_STAMP(si, gi) job->upd[gi]
#endif
    unsigned hits = 0;                       /* bit t = my pixel of sub-tile t is a hit */
    unsigned before[GEN_PER_THREAD];         /* hits of lower lanes of my wave in trip t */
    unsigned wave_total = 0;
    const bool want_items = a.items != nullptr && !lat;
    unsigned cands = 0;                      /* 3 bits per trip: candidate hypotheses of my pixel */
    unsigned ipos[GEN_PER_THREAD];           /* where my pixel's items start among the workgroup's */
    if (want_items && threadIdx.x == 0) s_items = 0;
    if (want_items) __syncthreads();
#pragma unroll
    for (int t = 0; t < GEN_PER_THREAD; ++t) {
        const int x = tx0 + t * 8 + lx, y = ty0 + wave * 8 + ly;
        bool any = false;
        unsigned cnt = 0;
        ipos[t] = 0;
        /* a patch needs a 2-pixel margin (patch_sampler.cc:47-50) */
        if (x >= MI_HALF && y >= MI_HALF && x < W - MI_HALF && y < H - MI_HALF) {
            const int pix = y * W + x;
            const int nb[4] = {pix - 1, pix + 1, pix - W, pix + W};
            const int sc = (wave * 8 + ly + 1) * SW + (t * 8 + lx + 1);                   /* my pixel in s_upd */
            const int snb[4] = {sc - 1, sc + 1, sc - SW, sc + SW};
            if (a.self_round) { if (MI_STAMP(sc, pix) == a.round - 1) { any = true; cnt = 1; } }
            else {
                unsigned fresh = 0;                                                        /* bit k: neighbour k was written last round */
#pragma unroll
                for (int k = 0; k < 4; ++k) fresh |= (MI_STAMP(snb[k], nb[k]) == a.round - 1 ? 1u : 0u) << k;
                if (fresh) {
                    const float own = job->conf[pix];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if ((fresh >> k) & 1u) {
                            const float c = job->conf[nb[k]];
                            if (own < c - 0.05f || own == 0.f) { any = true; ++cnt; }
                        }
                }
            }
        }
        if (want_items && any) { ipos[t] = atomicAdd(&s_items, cnt); cands |= cnt << (3 * t); }
        const unsigned long long m = __ballot(any);
        before[t] = wave_total + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        wave_total += (unsigned)__popcll(m);
        if (any) hits |= 1u << t;
    }
    if (lane == 0) s_wave_cnt[wave] = wave_total;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned tot = s_wave_cnt[0] + s_wave_cnt[1] + s_wave_cnt[2] + s_wave_cnt[3];
        s_base = tot ? atomicAdd(&(lat ? a.round_work_lat : a.round_work)[a.round], tot) : 0u;
        if (tot) atomicAdd(&a.view_count[(unsigned)(a.round % 3) * (unsigned)a.n_jobs + (unsigned)jobi], tot);
        if (want_items) s_ibase = s_items ? atomicAdd(&a.round_items[a.round], s_items) : 0u;
    }
    __syncthreads();
    unsigned off = s_base;
    for (int w = 0; w < wave; ++w) off += s_wave_cnt[w];
    DevEntry* out = lat ? a.work_lat : a.work;
#pragma unroll
    for (int t = 0; t < GEN_PER_THREAD; ++t)
        if ((hits >> t) & 1u) {
            DevEntry e; e.job = jobi; e.xy = (tx0 + t * 8 + lx) | ((ty0 + wave * 8 + ly) << 16);
            out[off + before[t]] = e;
            if (want_items) {
                const unsigned cnt = (cands >> (3 * t)) & 7u, ei = off + before[t];
                for (unsigned r = 0; r < cnt; ++r) a.items[s_ibase + ipos[t] + r] = (ei << 2) | r;
            }
        }
}

struct ApplyArgs {
    const DevJob* jobs;
    const DevEntry* work;
    const DevResult* results;
    const DevSpec* spec;             /* k_apply_spec: the speculative attempts, one record per item */
    const unsigned* items;           /* ... the round's (entry << 2 | rank) items, an entry's consecutive and rank 0 first */
    const unsigned* n_items;
    const unsigned* n_work_ptr;
    unsigned n_work;
    unsigned min_work, max_work;     /* the launch only acts if min_work <= n < max_work */
    int round;
    DevCounters* counters;
    unsigned long long* seed_keys;   /* seeds only: per job pixel arbitration keys */
    const unsigned* key_off;         /* seeds only: per job offset into seed_keys */
    int phase;                       /* seeds: 0 = vote, 1 = write */
    int seed_reopt;                  /* seeds: the records come from the re-optimising seed launch (DevSettings::seed_reopt) */
    unsigned* seed_count;            /* ... per job: the pixels the seeds wrote (what k_generate takes for the size of "round 1") */
};

/* e: the result's entry in the round's list (its view slots 8..15 are DevJob::results_x[2 e ..], sixteen-slot sets only) */
__device__ __forceinline__ void write_pixel(const DevJob* job, int pix, const DevResult& r, int round, unsigned e) {
    if (job->views_x) { job->views_x[2 * (size_t)pix] = job->results_x[2 * (size_t)e]; job->views_x[2 * (size_t)pix + 1] = job->results_x[2 * (size_t)e + 1]; }
    job->depth[pix] = r.depth;
    job->dz[2 * pix] = r.dzI; job->dz[2 * pix + 1] = r.dzJ;
    job->normal[3 * pix] = r.nx; job->normal[3 * pix + 1] = r.ny; job->normal[3 * pix + 2] = r.nz;
    job->conf[pix] = r.conf;
    job->views[pix] = r.views;
    if (job->views_hi) job->views_hi[pix] = r.views_hi;
    job->upd[pix] = round;
}

/* Jacobi write-back of one propagation round (dmrecon.cc:391-398). */
__global__ __launch_bounds__(256) void k_apply(ApplyArgs a) {
    const unsigned n = a.n_work_ptr ? *a.n_work_ptr : a.n_work;
    if (n < a.min_work || n >= a.max_work) return;
    unsigned filled = 0;
    for (unsigned base = blockIdx.x * 256; base < n; base += gridDim.x * 256) {
        const unsigned e = base + threadIdx.x;
        bool newly = false;
        int myjob = -1;
        if (e < n) {
            const DevResult r = a.results[e];
            if (r.accepted) {
                const DevEntry ent = a.work[e];
                const DevJob* job = a.jobs + ent.job;
                myjob = ent.job;
                const int pix = (ent.xy >> 16) * job->w + (ent.xy & 0xFFFF);
                newly = job->conf[pix] <= 0.f;
                write_pixel(job, pix, r, a.round, e);
            }
        }
        filled += (unsigned)__popcll(__ballot(newly));
        /* Progress::filled per view: one atomic per (wavefront, job) */
        unsigned long long todo = __ballot(newly);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int lj = __shfl(myjob, leader);
            const unsigned long long same = __ballot(newly && myjob == lj);
            if ((int)(threadIdx.x & 63) == leader) atomicAdd(const_cast<uint32_t*>(&a.jobs[lj].n_filled), (unsigned)__popcll(same));
            todo &= ~same;
        }
    }
    /* the call's counter: one atomic per workgroup (an atomic on one word costs ~10 ns; per wavefront they add up to
     * milliseconds in the launches of a large batch) */
    __shared__ unsigned s_filled;
    if (threadIdx.x == 0) s_filled = 0;
    __syncthreads();
    if (filled && (threadIdx.x & 63) == 0) atomicAdd(&s_filled, filled);
    __syncthreads();
    if (threadIdx.x == 0 && s_filled) atomicAdd(&a.counters->n_filled, (unsigned long long)s_filled);
}

/* The write-back of a speculative round (k_optimize_spec): the reference's sequential rule over an entry's candidate
 * attempts (pop-time test dmrecon.cc:371, acceptance :378,391) from their records, then the Jacobi write-back of k_apply.
 * Counts what the reference would have run -- not the speculative extras. */
__global__ __launch_bounds__(256) void k_apply_spec(ApplyArgs a) {
    const unsigned n = a.n_work_ptr ? *a.n_work_ptr : a.n_work;
    if (n < a.min_work || n >= a.max_work) return;
    unsigned filled = 0, n_eval = 0, n_pass = 0, n_patch = 0;
    const unsigned n_items = *a.n_items;
    for (unsigned base = blockIdx.x * 256; base < n_items; base += gridDim.x * 256) {
        const unsigned i = base + threadIdx.x;
        bool newly = false;
        int myjob = -1;
        const unsigned item = i < n_items ? a.items[i] : 1u;
        if ((item & 3u) == 0u) {                             /* one lane per entry: the one that holds its rank-0 item */
            const unsigned e = item >> 2;
            const DevSpec* rec = a.spec + i;
            const int n_cand = rec[0].n_cand;
            if (n_cand > 0) {
                const float own = rec[0].own;
                float best = own; int fin = -1;
                for (int s = 0; s < n_cand; ++s) {
                    if (best > rec[s].bc) break;                                       /* dmrecon.cc:371 (and every later one) */
                    ++n_patch; n_eval += rec[s].n_eval; n_pass += rec[s].n_pass;
                    if (const unsigned d = (unsigned)rec[s].pad) {
                        /* what this attempt noted instead of doing (Run::deferred) -- it is one the rule consumes: the views it
                         * replaced, and the footprint exception, which ends the VIEW as in every other form of a round */
                        if ((d >> 8) & 0xFFu) atomicAdd(&a.counters->n_view_replaced, (unsigned long long)((d >> 8) & 0xFFu));
                        if ((d >> 16) & 0xFFu) atomicAdd(&a.counters->n_iter14, (unsigned long long)((d >> 16) & 0xFFu));
                        if (d & 1u) {
                            atomicOr(const_cast<int32_t*>(&a.jobs[a.work[e].job].flags), (int)MI_JOB_EFOOTPRINT);
                            atomicOr(&a.counters->error_flags, 1u);
                            fin = -1;
                            break;
                        }
                    }
                    const float cf = rec[s].conf;
                    if (cf > 0.f && best < cf) { best = cf; fin = s; }                 /* dmrecon.cc:378,391 */
                }
                if (fin >= 0) {
                    const DevSpec f = rec[fin];
                    const DevEntry ent = a.work[e];
                    const DevJob* job = a.jobs + ent.job;
                    myjob = ent.job;
                    const int pix = (ent.xy >> 16) * job->w + (ent.xy & 0xFFFF);
                    DevResult r;
                    r.conf = f.conf; r.depth = f.depth; r.dzI = f.dzI; r.dzJ = f.dzJ; r.nx = f.nx; r.ny = f.ny; r.nz = f.nz;
                    r.views = f.views; r.views_hi = f.views_hi; r.iters = f.iters; r.accepted = 1; r.tried = 0;
                    newly = own <= 0.f;
                    write_pixel(job, pix, r, a.round, e);
                }
            }
        }
        filled += (unsigned)__popcll(__ballot(newly));
        unsigned long long todo = __ballot(newly);
        while (todo) {                                       /* Progress::filled per view: one atomic per (wavefront, job) */
            const int leader = __ffsll((long long)todo) - 1;
            const int lj = __shfl(myjob, leader);
            const unsigned long long same = __ballot(newly && myjob == lj);
            if ((int)(threadIdx.x & 63) == leader) atomicAdd(const_cast<uint32_t*>(&a.jobs[lj].n_filled), (unsigned)__popcll(same));
            todo &= ~same;
        }
    }
    /* the call's counters: one atomic per workgroup and counter (an atomic on one word costs ~10 ns: per wavefront they
     * were most of this kernel's time) */
    __shared__ unsigned s_red[4];
    if (threadIdx.x < 4) s_red[threadIdx.x] = 0;
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) {
        n_eval += __shfl_down(n_eval, off); n_pass += __shfl_down(n_pass, off); n_patch += __shfl_down(n_patch, off);
    }
    if ((threadIdx.x & 63) == 0) {
        if (filled) atomicAdd(&s_red[0], filled);
        if (n_eval) atomicAdd(&s_red[1], n_eval);
        if (n_pass) atomicAdd(&s_red[2], n_pass);
        if (n_patch) atomicAdd(&s_red[3], n_patch);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_red[0]) atomicAdd(&a.counters->n_filled, (unsigned long long)s_red[0]);
        if (s_red[1]) { atomicAdd(&a.counters->n_eval, (unsigned long long)s_red[1]); atomicAdd(&a.counters->k_eval[MI_KIND_SPEC], (unsigned long long)s_red[1]); }
        if (s_red[2]) { atomicAdd(&a.counters->n_pass, (unsigned long long)s_red[2]); atomicAdd(&a.counters->k_pass[MI_KIND_SPEC], (unsigned long long)s_red[2]); }
        if (s_red[3]) { atomicAdd(&a.counters->n_patch, (unsigned long long)s_red[3]); atomicAdd(&a.counters->k_patch[MI_KIND_SPEC], (unsigned long long)s_red[3]); }
    }
}

/* Seeds (dmrecon.cc:297-330): several features may round to the same pixel; the sequential
 * reference keeps the highest confidence, the earlier feature on ties.  Two phases:
 * vote (64-bit atomicMax of conf | ~index) then write by the winner. */
__global__ __launch_bounds__(256) void k_apply_seeds(ApplyArgs a) {
    const unsigned e = blockIdx.x * 256 + threadIdx.x;
    bool newly = false, okseed = false;
    if (e < a.n_work) {
        const DevResult r = a.results[e];
        /* the confidence the features of a pixel are compared by is the one processFeatures saw: of the seed's FIRST optimisation
         * (with the re-optimising launch it travels in `accepted`, the record's own may be the re-optimised one) */
        const float c1 = a.seed_reopt ? __uint_as_float((unsigned)r.accepted) : r.conf;
        if (c1 > 0.f) {
            const DevEntry ent = a.work[e];
            const DevJob* job = a.jobs + ent.job;
            const int pix = (ent.xy >> 16) * job->w + (ent.xy & 0xFFFF);
            const unsigned long long key = ((unsigned long long)__float_as_uint(c1) << 32) | (0xFFFFFFFFu - e);
            unsigned long long* slot = a.seed_keys + a.key_off[ent.job] + pix;
            if (a.phase == 0) {
                atomicMax(slot, key);
                okseed = true;
            } else if (*slot == key) {
                newly = job->conf[pix] <= 0.f;
                /* a seed whose re-optimisation did not raise its confidence keeps the stamp of round 0 and the propagation starts
                 * with round 2: it is nobody's source; the ones that propagate are stamped as written in round 1, which is where
                 * the reference's pop would have rewritten them */
                write_pixel(job, pix, r, a.seed_reopt ? (r.tried ? 1 : 0) : a.round, e);
                if (a.seed_reopt && a.seed_count) atomicAdd(a.seed_count + ent.job, 1u);
                if (newly) atomicAdd(const_cast<uint32_t*>(&job->n_filled), 1u);
            }
        }
    }
    const unsigned long long m = __ballot(newly);
    if (m && (threadIdx.x & 63) == __ffsll((long long)m) - 1)
        atomicAdd(&a.counters->n_filled, (unsigned long long)__popcll(m));
    const unsigned long long m2 = __ballot(okseed);
    if (m2 && (threadIdx.x & 63) == __ffsll((long long)m2) - 1)
        atomicAdd(&a.counters->n_seeds_ok, (unsigned long long)__popcll(m2));
}

/* ------------------------------------------------------------------------- */
/* Image staging.                                                              */

/* The job records of a batch as they were uploaded -- packed, words_per_job words each: the part of DevJob its global views use
 * (BatchRun::upload) -- to their places in the job array. */
__global__ __launch_bounds__(256) void k_unpack_jobs(const uint32_t* __restrict__ packed, unsigned words_per_job, uint32_t* __restrict__ jobs, unsigned words_per_slot) {
    const unsigned j = blockIdx.y;
    for (unsigned w = blockIdx.x * 256 + threadIdx.x; w < words_per_job; w += gridDim.x * 256)
        jobs[(size_t)j * words_per_slot + w] = packed[(size_t)j * words_per_job + w];
}

/* interleaved 1..4 channel uint8 -> RGBA8 (grey expanded, alpha dropped; image_pyramid.cc:65-73) */
__global__ __launch_bounds__(256) void k_pack_rgba(const uint8_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                  int n, int channels) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t r, g, b;
    if (channels <= 2) { r = g = b = src[(size_t)i * channels]; }
    else { r = src[(size_t)i * channels]; g = src[(size_t)i * channels + 1]; b = src[(size_t)i * channels + 2]; }
    dst[i] = r | (g << 8) | (b << 16) | 0xFF000000u;
}

__global__ __launch_bounds__(256) void k_unpack_rgb(const uint32_t* __restrict__ src, uint8_t* __restrict__ dst, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = src[i];
    dst[3 * (size_t)i] = t & 255u; dst[3 * (size_t)i + 1] = (t >> 8) & 255u; dst[3 * (size_t)i + 2] = (t >> 16) & 255u;
}

/*
 * One level of the Gaussian pyramid: mve::image::rescale_half_size_gaussian<uint8_t>(img, 1.f)
 * (libs/mve/image_tools.h:619-690) with math::Accum<unsigned char> (libs/math/accum.h:146-171):
 * 16 taps accumulated in float in the reference's order (no FMA contraction, so the bytes
 * match a non-contracting CPU build), clamped borders, divide by the accumulated weight,
 * round half away from zero.
 */
__global__ __launch_bounds__(256) void k_pyramid(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                int iw, int ih, int ow, int oh, float w1, float w2, float w3) {
    /* hipcc contracts a*b+c into an FMA by default (even through __fmul_rn/__fadd_rn); the reference
     * bytes come from separately rounded multiply and add, and ties at x.5 do occur in practice. */
#pragma clang fp contract(off)
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= ow || y >= oh) return;
    const int y2 = 2 * y, x2 = 2 * x;
    int ry[4], rx[4];
    ry[0] = max(0, y2 - 1); ry[1] = y2; ry[2] = min(ih - 1, y2 + 1); ry[3] = min(ih - 1, y2 + 2);
    rx[0] = max(0, x2 - 1); rx[1] = x2; rx[2] = min(iw - 1, x2 + 1); rx[3] = min(iw - 1, x2 + 2);
    const float wt[4][4] = {{w3, w2, w2, w3}, {w2, w1, w1, w2}, {w2, w1, w1, w2}, {w3, w2, w2, w3}};
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, wsum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t t = src[(size_t)ry[r] * iw + rx[k]];
            /* plain operators: they (unlike the __f*_rn header wrappers) obey the pragma above */
            const float p0 = (float)(t & 255u) * wt[r][k];
            const float p1 = (float)((t >> 8) & 255u) * wt[r][k];
            const float p2 = (float)((t >> 16) & 255u) * wt[r][k];
            v0 = v0 + p0; v1 = v1 + p1; v2 = v2 + p2;
            wsum = wsum + wt[r][k];
        }
    /* math::round: x > 0 ? floor(x + 0.5) : ceil(x - 0.5) */
    const float q0 = v0 / wsum, q1 = v1 / wsum, q2 = v2 / wsum;
    const uint32_t b0 = (uint32_t)(q0 > 0.f ? floorf(q0 + 0.5f) : 0.f);
    const uint32_t b1 = (uint32_t)(q1 > 0.f ? floorf(q1 + 0.5f) : 0.f);
    const uint32_t b2 = (uint32_t)(q2 > 0.f ? floorf(q2 + 0.5f) : 0.f);
    dst[(size_t)y * ow + x] = (b0 & 255u) | ((b1 & 255u) << 8) | ((b2 & 255u) << 16) | 0xFF000000u;
}

/* RGBA8 level -> 16-byte footprint records (DevView::quad): one lane per texel, neighbours edge-clamped. */
/* What the host wants to know about a round (or a chunk of rounds), written to page-locked host memory in ONE dispatch:
 * the list sizes (two runs of words), the batch counters and every job's flags / n_filled pair -- they live in four
 * places on the device, and four 4-to-100-byte copies are four blit dispatches (4 us each plus the gaps between
 * dependent dispatches) on the critical path of every host-visible round.  One workgroup. */
__global__ __launch_bounds__(256) void k_round_report(const unsigned* __restrict__ a, int n_a, const unsigned* __restrict__ b, int n_b,
                                                       const DevCounters* __restrict__ counters, const DevJob* __restrict__ jobs, int n_jobs,
                                                       unsigned* __restrict__ out_rw, unsigned* __restrict__ out_hc, unsigned* __restrict__ out_dyn,
                                                       const unsigned* __restrict__ view_count) {
    const int tid = (int)threadIdx.x;
    for (int i = tid; i < n_a; i += 256) out_rw[i] = a[i];
    for (int i = tid; i < n_b; i += 256) out_rw[n_a + i] = b[i];
    if (tid < (int)(sizeof(DevCounters) / sizeof(unsigned))) out_hc[tid] = reinterpret_cast<const unsigned*>(counters)[tid];
    static_assert(offsetof(DevJob, n_filled) == offsetof(DevJob, flags) + 4, "flags and n_filled are read as a pair");
    for (int j = tid; j < n_jobs; j += 256) {
        out_dyn[3 * j] = (unsigned)jobs[j].flags;
        out_dyn[3 * j + 1] = jobs[j].n_filled;
        out_dyn[3 * j + 2] = view_count ? view_count[j] : 0u;    /* the view's list of this round (k_generate), where there is one */
    }
}

__global__ __launch_bounds__(256) void k_quadify(const uint32_t* __restrict__ src, u32x4* __restrict__ dst, int w, int h) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w * h) return;
    const int y = i / w, x = i - y * w;
    const int x1 = x + 1 < w ? x + 1 : x, y1 = y + 1 < h ? y + 1 : y;
    u32x4 o;
    o.x = src[y * w + x]; o.y = src[y * w + x1]; o.z = src[y1 * w + x]; o.w = src[y1 * w + x1];
#ifdef MI_TILED_QUADS
    dst[(((unsigned)(y >> 3) * ((unsigned)(w + 7) >> 3) + (unsigned)(x >> 3)) << 6) | ((unsigned)(y & 7) << 3) | (unsigned)(x & 7)] = o;
#elif defined(MI_PAIR_RECORDS)
    reinterpret_cast<uint32_t*>(dst)[2 * (size_t)i] = o.x; reinterpret_cast<uint32_t*>(dst)[2 * (size_t)i + 1] = o.z;   /* (x, y), (x, y + 1) */
#else
    dst[(size_t)i * (MI_QUAD_WORDS / 4)] = o;
#if MI_QUAD_WORDS > 4
    dst[(size_t)i * (MI_QUAD_WORDS / 4) + 1] = o; dst[(size_t)i * (MI_QUAD_WORDS / 4) + 2] = o;
#endif
#endif
}

#if MI_FW == 5
/* The first follow-up list of a round from the masks its first launch left (OptArgs::follow_mask), in the order of the round's
 * list.  MI_FOLLOW_BLOCKS workgroups, each with a contiguous run of wavefront units; k_follow_count: the entries of every run;
 * k_follow_scatter: a run's offset = the sum of the runs before it, then the entries, ascending.  lpp: lanes per patch (a set
 * bit i of a mask = patch i / lpp of the unit), ppw: patches per wavefront. */
#define MI_FOLLOW_BLOCKS 1024
struct FollowArgs {
    const unsigned long long* mask; const unsigned* n_work_ptr; unsigned min_work, max_work;
    unsigned ppw, lpp; unsigned* blk_sum; unsigned* out; unsigned* out_n;
};
__device__ __forceinline__ bool follow_run(const FollowArgs& a, unsigned& first, unsigned& last) {
    const unsigned n = *a.n_work_ptr;
    if (n < a.min_work || n >= a.max_work) return false;
    const unsigned n_units = (n + a.ppw - 1) / a.ppw;
    const unsigned per = ((n_units + MI_FOLLOW_BLOCKS - 1) / MI_FOLLOW_BLOCKS + 255u) & ~255u;      /* whole tiles of 256 units */
    first = blockIdx.x * per; last = first + per < n_units ? first + per : n_units;
    return true;
}
__global__ __launch_bounds__(256) void k_follow_count(FollowArgs a) {
    __shared__ unsigned s_sum;
    unsigned first = 0, last = 0;
    const bool on = follow_run(a, first, last);
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
    unsigned c = 0;
    if (on) for (unsigned u = first + threadIdx.x; u < last; u += 256) c += (unsigned)__popcll(a.mask[u]);
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_sum, c);
    __syncthreads();
    if (threadIdx.x == 0) a.blk_sum[blockIdx.x] = s_sum;
}
__global__ __launch_bounds__(256) void k_follow_scatter(FollowArgs a) {
    __shared__ unsigned s_part[4], s_wave[4], s_base;
    unsigned first = 0, last = 0;
    const bool on = follow_run(a, first, last);
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    /* this run's offset: the entries of the runs before it (1024 words: four per thread) */
    unsigned before = 0, total = 0;
    for (unsigned b = (unsigned)tid; b < MI_FOLLOW_BLOCKS; b += 256) { const unsigned v = a.blk_sum[b]; total += v; if (b < blockIdx.x) before += v; }
    for (int off = 32; off > 0; off >>= 1) { before += __shfl_down(before, off); total += __shfl_down(total, off); }
    if (lane == 0) { s_part[wave] = before; s_wave[wave] = total; }
    __syncthreads();
    if (tid == 0) {
        s_base = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        if (blockIdx.x == 0) *a.out_n = on ? s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3] : 0u;
    }
    __syncthreads();
    if (!on) return;
    unsigned run = s_base;
    for (unsigned u0 = first; u0 < last; u0 += 256) {
        const unsigned u = u0 + (unsigned)tid;
        const unsigned long long m = u < last ? a.mask[u] : 0ull;
        const unsigned c = (unsigned)__popcll(m);
        /* exclusive prefix of c over the tile: inside the wavefront, then across the four */
        unsigned inc = c;
        for (int off = 1; off < 64; off <<= 1) { const unsigned v = __shfl_up(inc, off); if (lane >= off) inc += v; }
        __syncthreads();                                   /* (s_wave of the tile before has been read) */
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        unsigned pre = inc - c;
        for (int w = 0; w < wave; ++w) pre += s_wave[w];
        unsigned o = run + pre;
        for (unsigned long long r = m; r; r &= r - 1ull) a.out[o++] = u * a.ppw + (unsigned)(__ffsll((long long)r) - 1) / a.lpp;
        run += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    }
}

/* A one-lane kernel that does nothing: profiles are cut at its dispatches (mi_dmrecon_debug_region_mark: bench.py brackets every
 * timed region with one, tools/trace_regions.py keeps the dispatches between the marks of a rocprofv3 kernel trace). */
__global__ void k_region_mark(unsigned tag) { (void)tag; }
#endif

}  /* namespace MI_FWNS */
using namespace MI_FWNS;

/* ------------------------------------------------------------------------- */
/* Host-callable launchers (declared in dmrecon_device.h).                     */

#if MI_FW == 5
unsigned long long* mi_debug_tbuf = nullptr;   /* the debug buffer of MI_PROBE builds (mi_dmrecon_debug_buffer) */
#endif

static void launch_optimize(hipStream_t s, int lanes_per_view, unsigned grid_blocks, const DevJob* jobs, const DevView* views,
                        const float* lut, const DevSettings& st, const DevEntry* work, const DevHyp* hyp,
                        DevResult* results, const unsigned* n_work_ptr, unsigned n_work, unsigned min_work,
                        unsigned max_work, int round, DevCounters* counters, const unsigned* follow_in,
                        const unsigned* follow_in_n, unsigned* follow_out, unsigned* follow_out_n, unsigned follow_seg,
                        unsigned follow_seg_in, unsigned long long* follow_mask) {
    if (grid_blocks == 0) return;
    grid_blocks = (grid_blocks + MI_XCDS - 1) / MI_XCDS * MI_XCDS;      /* (XcdRange: every XCD the same number of workgroups) */
    OptArgs a;
    a.follow_seg = follow_seg; a.follow_seg_in = follow_seg_in; a.follow_mask = follow_mask;
    { const char* e = getenv("MI_DMRECON_DEBUG_SCRAMBLE"); a.scramble = (e && follow_in == nullptr && hyp == nullptr) ? (unsigned)atoi(e) : 0u; }
    a.jobs = jobs; a.views = views; a.lut = lut; a.st = st; a.work = work; a.hyp = hyp; a.results = results;
    a.n_work_ptr = n_work_ptr; a.n_work = n_work; a.min_work = min_work; a.max_work = max_work;
    a.round = round; a.counters = counters; a.tbuf = mi_debug_tbuf;
    a.max_attempts = (follow_out || follow_mask) ? 1 : 4; a.follow_in = follow_in; a.follow_in_n = follow_in_n;
    a.follow_out = follow_out; a.follow_out_n = follow_out_n;
    /* lanes_per_view: 1 = throughput layout, 2 = throughput layout and the FAST kernel for a launch that CONTINUES a follow-up list
     * (below), anything else = latency layout; st.K > 4: the eight-slot layouts */
    const bool lat = lanes_per_view != 1 && lanes_per_view != 2, eight = st.K > 4;
    if (st.K > 8) {
        /* sixteen view slots: the general kernel of the throughput layout is all there is (Lay<1, 16>) -- the entries'
         * attempts in a row, or one each where the caller keeps follow-up lists or gives explicit hypotheses */
        if (lat) return;
        const unsigned ncc16 = (unsigned)(Lay<1, 16>::PATCHES * st.ncc_stride * sizeof(float));
        const bool single16 = hyp != nullptr || follow_out != nullptr;
        if (hyp != nullptr && st.seed_reopt) hipLaunchKernelGGL((k_optimize<Lay<1, 16>, false, true, true>), dim3(grid_blocks), dim3(WAVE), ncc16, s, a);
        else if (single16) hipLaunchKernelGGL((k_optimize<Lay<1, 16>, false, true>), dim3(grid_blocks), dim3(WAVE), ncc16, s, a);
        else hipLaunchKernelGGL((k_optimize<Lay<1, 16>, false>), dim3(grid_blocks), dim3(WAVE), ncc16, s, a);
        return;
    }
    /* the first launch of a bulk round (one attempt per entry, a follow-up list for the rest) runs the FAST kernel: no
     * view selection code in it -- a patch that needs one goes to the follow-up launch, which is the general kernel */
    /* (lanes_per_view 2: second attempts in the FAST kernel as well -- same arithmetic, a patch that needs a view selection is
     * abandoned once more and goes on to the next list, which the general kernel takes) */
    const bool fast = !lat && (follow_out != nullptr || follow_mask != nullptr) && (follow_in == nullptr || lanes_per_view == 2) && hyp == nullptr && st.K <= 8;
    /* ... a launch that continues a follow-up list AND leaves one runs one attempt per entry as well, in the general kernel
     * (process_entry_single), and so do the seeds (one hypothesis each); only a propagation launch without a follow-up list
     * of its own runs an entry's attempts in a row */
    const bool single = !lat && !fast && (hyp != nullptr || (follow_out != nullptr && follow_in != nullptr));
    /* the throughput kernels with a view selection in them: its NCC table in dynamic shared memory (lds_ncc) */
    const unsigned ncc4 = (unsigned)(Lay<1, 4>::PATCHES * st.ncc_stride * sizeof(float)), ncc8 = (unsigned)(Lay<1, 8>::PATCHES * st.ncc_stride * sizeof(float));
    if (lat) {
        if (eight) hipLaunchKernelGGL((k_optimize<Lay<8, 8>, false>), dim3(grid_blocks), dim3(WAVE), 0, s, a);
        else hipLaunchKernelGGL((k_optimize<Lay<16, 4>, false>), dim3(grid_blocks), dim3(WAVE), 0, s, a);
    } else if (fast) {
        if (eight) hipLaunchKernelGGL((k_optimize<Lay<1, 8>, true>), dim3(grid_blocks), dim3(WAVE), 0, s, a);
        else hipLaunchKernelGGL((k_optimize<Lay<1, 4>, true>), dim3(grid_blocks), dim3(WAVE), 0, s, a);
    } else if (single && hyp != nullptr && st.seed_reopt) {
        /* the seeds, each re-optimised once from its own result (the reference's seed semantics): a kernel of its own, so that
         * the single-attempt kernel of the propagation rounds stays what it is */
        if (eight) hipLaunchKernelGGL((k_optimize<Lay<1, 8>, false, true, true>), dim3(grid_blocks), dim3(WAVE), ncc8, s, a);
        else hipLaunchKernelGGL((k_optimize<Lay<1, 4>, false, true, true>), dim3(grid_blocks), dim3(WAVE), ncc4, s, a);
    } else if (single) {
        if (eight) hipLaunchKernelGGL((k_optimize<Lay<1, 8>, false, true>), dim3(grid_blocks), dim3(WAVE), ncc8, s, a);
        else hipLaunchKernelGGL((k_optimize<Lay<1, 4>, false, true>), dim3(grid_blocks), dim3(WAVE), ncc4, s, a);
    } else {
        if (eight) hipLaunchKernelGGL((k_optimize<Lay<1, 8>, false>), dim3(grid_blocks), dim3(WAVE), ncc8, s, a);
        else hipLaunchKernelGGL((k_optimize<Lay<1, 4>, false>), dim3(grid_blocks), dim3(WAVE), ncc4, s, a);
    }
}

static void launch_optimize_spec(hipStream_t s, unsigned grid_blocks, const DevJob* jobs, const DevView* views, const float* lut,
                                 const DevSettings& st, const DevEntry* work, DevSpec* spec, const unsigned* items, const unsigned* n_items,
                                 const unsigned* n_work_ptr, unsigned n_work,
                                 unsigned min_work, unsigned max_work, int round, DevCounters* counters) {
    if (grid_blocks == 0) return;
    grid_blocks = (grid_blocks + MI_XCDS - 1) / MI_XCDS * MI_XCDS;
    SpecArgs t;
    t.o.jobs = jobs; t.o.views = views; t.o.lut = lut; t.o.st = st; t.o.work = work; t.o.hyp = nullptr; t.o.results = nullptr;
    t.o.n_work_ptr = n_work_ptr; t.o.n_work = n_work; t.o.min_work = min_work; t.o.max_work = max_work; t.o.round = round;
    t.o.counters = counters; t.o.tbuf = mi_debug_tbuf;
    t.o.max_attempts = 1; t.o.follow_in = nullptr; t.o.follow_in_n = nullptr; t.o.follow_out = nullptr; t.o.follow_out_n = nullptr; t.o.follow_seg = 0; t.o.follow_seg_in = 0; t.o.follow_mask = nullptr; t.o.scramble = 0;
    t.spec = spec; t.items = items; t.n_items = n_items;
    if (st.K > 4) hipLaunchKernelGGL((k_optimize_spec<Lay<1, 8> >), dim3(grid_blocks), dim3(WAVE), (unsigned)(Lay<1, 8>::PATCHES * st.ncc_stride * sizeof(float)), s, t);
    else hipLaunchKernelGGL((k_optimize_spec<Lay<1, 4> >), dim3(grid_blocks), dim3(WAVE), (unsigned)(Lay<1, 4>::PATCHES * st.ncc_stride * sizeof(float)), s, t);
}

static void launch_patch_eval(hipStream_t s, const DevJob* job, const DevView* views, const float* lut,
                          const DevSettings& st, int x, int y, float depth, float dzI, float dzJ,
                          float* master, float* ncc, int32_t* ok, float* col, float* deriv, int32_t* level) {
    EvalArgs a;
    a.job = job; a.views = views; a.lut = lut; a.st = st; a.x = x; a.y = y; a.depth = depth; a.dzI = dzI; a.dzJ = dzJ;
    a.master = master; a.ncc = ncc; a.ok = ok; a.col = col; a.deriv = deriv; a.level = level;
    hipLaunchKernelGGL(k_patch_eval, dim3(1), dim3(WAVE), 0, s, a);
}

static void launch_generate(hipStream_t s, const DevJob* jobs, int n_jobs, int max_tiles, DevEntry* work, DevEntry* work_lat,
                        unsigned* round_work, unsigned* round_work_lat, unsigned* view_count, unsigned* view_mode,
                        unsigned handover, int round, unsigned* items, unsigned* round_items, int self_round) {
    SweepArgs a;
    a.self_round = self_round;
    a.jobs = jobs; a.work = work; a.work_lat = work_lat; a.round_work = round_work; a.round_work_lat = round_work_lat;
    a.view_count = view_count; a.view_mode = view_mode; a.handover = handover; a.n_jobs = n_jobs; a.round = round;
    a.items = items; a.round_items = round_items;
    hipLaunchKernelGGL(k_generate, dim3(max_tiles, n_jobs), dim3(256), 0, s, a);
}

#if MI_FW == 5
void mi_launch_apply(hipStream_t s, unsigned grid_blocks, const DevJob* jobs, const DevEntry* work, const DevResult* results,
                     const unsigned* n_work_ptr, unsigned n_work, unsigned min_work, unsigned max_work, int round, DevCounters* counters) {
    if (grid_blocks == 0) return;
    ApplyArgs a;
    a.jobs = jobs; a.work = work; a.results = results; a.spec = nullptr; a.items = nullptr; a.n_items = nullptr; a.n_work_ptr = n_work_ptr; a.n_work = n_work;
    a.min_work = min_work; a.max_work = max_work; a.round = round;
    a.counters = counters; a.seed_keys = nullptr; a.key_off = nullptr; a.phase = 0; a.seed_reopt = 0; a.seed_count = nullptr;
    hipLaunchKernelGGL(k_apply, dim3(grid_blocks), dim3(256), 0, s, a);
}
void mi_launch_apply_spec(hipStream_t s, unsigned grid_blocks, const DevJob* jobs, const DevEntry* work, const DevSpec* spec,
                          const unsigned* items, const unsigned* n_items,
                          const unsigned* n_work_ptr, unsigned n_work, unsigned min_work, unsigned max_work, int round, DevCounters* counters) {
    if (grid_blocks == 0) return;
    ApplyArgs a;
    a.jobs = jobs; a.work = work; a.results = nullptr; a.spec = spec; a.items = items; a.n_items = n_items; a.n_work_ptr = n_work_ptr; a.n_work = n_work;
    a.min_work = min_work; a.max_work = max_work; a.round = round;
    a.counters = counters; a.seed_keys = nullptr; a.key_off = nullptr; a.phase = 0; a.seed_reopt = 0; a.seed_count = nullptr;
    hipLaunchKernelGGL(k_apply_spec, dim3(grid_blocks), dim3(256), 0, s, a);
}

#endif

static void launch_tail(hipStream_t s, unsigned grid_blocks, const DevJob* jobs, const DevView* views, const float* lut,
                    const DevSettings& st, const DevEntry* prev_work, const DevResult* prev_results, DevEntry* work,
                    DevResult* results, unsigned* round_work, int round, DevCounters* counters, bool speculative) {
    TailArgs t;
    t.o.jobs = jobs; t.o.views = views; t.o.lut = lut; t.o.st = st; t.o.work = work; t.o.hyp = nullptr; t.o.results = results;
    t.o.n_work_ptr = nullptr; t.o.n_work = 0; t.o.min_work = 0; t.o.max_work = 0xFFFFFFFFu; t.o.round = round;
    t.o.counters = counters; t.o.tbuf = mi_debug_tbuf;
    t.o.max_attempts = 4; t.o.follow_in = nullptr; t.o.follow_in_n = nullptr; t.o.follow_out = nullptr; t.o.follow_out_n = nullptr; t.o.follow_seg = 0; t.o.follow_seg_in = 0; t.o.follow_mask = nullptr; t.o.scramble = 0;
    t.prev_work = prev_work; t.prev_results = prev_results; t.round_work = round_work;
    if (st.K > 4) {
        if (speculative) hipLaunchKernelGGL((k_tail<true, 8>), dim3(grid_blocks), dim3(MI_TAIL_WAVES * WAVE), 0, s, t);
        else hipLaunchKernelGGL((k_tail<false, 8>), dim3(grid_blocks), dim3(WAVE), 0, s, t);
    } else {
        if (speculative) hipLaunchKernelGGL((k_tail<true, 4>), dim3(grid_blocks), dim3(MI_TAIL_WAVES * WAVE), 0, s, t);
        else hipLaunchKernelGGL((k_tail<false, 4>), dim3(grid_blocks), dim3(WAVE), 0, s, t);
    }
}

static void launch_front(hipStream_t s, int n_jobs, const DevJob* jobs, const DevView* views, const float* lut, const DevSettings& st,
                         const DevEntry* list, const DevResult* list_results, const unsigned* list_n,
                         DevEntry* work0, DevResult* results0, DevEntry* work1, DevResult* results1,
                         const unsigned* job_off, unsigned* job_count, unsigned* job_stats, int first_round, int max_rounds,
                         DevCounters* counters, int team, unsigned long long* mail, unsigned* team_flags,
                         const unsigned long long* job_start, unsigned long long* job_resume, unsigned* team_filled, unsigned spin_ticks,
                         int fault, int n_xcd, unsigned* host_done, const unsigned* block_map, unsigned grid_blocks, const unsigned* job_order) {
    static_assert(MI_FRONT_MAIL_WORDS == 2 * MI_FRONT_QCAP * 4 * MI_FRONT_GRAN, "mailbox size");
    if (n_jobs <= 0) return;
    if (!job_start) {
        FrontSplitArgs sp;
        sp.work = list; sp.results = list_results; sp.n_ptr = list_n; sp.owork = work0; sp.oresults = results0;
        sp.job_off = job_off; sp.job_count = job_count;
        hipLaunchKernelGGL(k_front_split, dim3(16), dim3(256), 0, s, sp);
    }
    FrontArgs t;
    t.o.jobs = jobs; t.o.views = views; t.o.lut = lut; t.o.st = st; t.o.work = nullptr; t.o.hyp = nullptr; t.o.results = nullptr;
    t.o.n_work_ptr = nullptr; t.o.n_work = 0; t.o.min_work = 0; t.o.max_work = 0xFFFFFFFFu; t.o.round = first_round;
    t.o.counters = counters; t.o.tbuf = mi_debug_tbuf;
    t.o.max_attempts = 4; t.o.follow_in = nullptr; t.o.follow_in_n = nullptr; t.o.follow_out = nullptr; t.o.follow_out_n = nullptr; t.o.follow_seg = 0; t.o.follow_seg_in = 0; t.o.follow_mask = nullptr; t.o.scramble = 0;
    t.work[0] = work0; t.work[1] = work1; t.results[0] = results0; t.results[1] = results1;
    t.job_off = job_off; t.job_count = job_count; t.job_start = job_start; t.job_resume = job_resume;
    t.job_stats = job_stats; t.max_rounds = max_rounds;
    t.team = 1; t.mail = nullptr; t.team_flags = nullptr; t.team_filled = team_filled; t.spin_ticks = spin_ticks;
    t.fault_member = fault < 0 ? -1 : (fault & 0xFF); t.fault_round = fault < 0 ? 0 : ((fault >> 8) & 0xFFFF);
    t.force_write_through = fault >= 0 && ((fault >> 24) & 1) != 0;
    if (fault >= 0 && (fault & 0xFF) == 0xFF) t.fault_member = -1;          /* (write-through forced, nobody vanishes) */
    t.n_jobs = n_jobs; t.n_xcd = n_xcd < 1 ? 1 : n_xcd;
    t.l2_exchange = (fault >= 0 && ((fault >> 25) & 1)) ? 0 : 1;
    t.host_done = host_done;
    t.block_map = nullptr; t.job_order = job_order;
    if (team > 1 && mail && team_flags && team_filled && block_map && grid_blocks > 0) {
        t.team = team > MI_FRONT_TEAM_MAX ? MI_FRONT_TEAM_MAX : team; t.mail = mail; t.team_flags = team_flags;
        t.block_map = block_map;
        const unsigned grid = grid_blocks;
        if (st.K > 4) hipLaunchKernelGGL((k_front<8, true>), dim3(grid), dim3(MI_FRONT_WAVES * WAVE), 0, s, t);
        else hipLaunchKernelGGL((k_front<4, true>), dim3(grid), dim3(MI_FRONT_WAVES * WAVE), 0, s, t);
        hipLaunchKernelGGL(k_front_commit, dim3((unsigned)(n_jobs + 255) / 256), dim3(256), 0, s, jobs, team_filled, counters, n_jobs);
    }
    else if (st.K > 4) hipLaunchKernelGGL((k_front<8, false>), dim3((unsigned)n_jobs), dim3(MI_FRONT_WAVES * WAVE), 0, s, t);
    else hipLaunchKernelGGL((k_front<4, false>), dim3((unsigned)n_jobs), dim3(MI_FRONT_WAVES * WAVE), 0, s, t);
}

#if MI_FW == 5
void mi_launch_flatten(hipStream_t s, float* maps, uint32_t* imaps, size_t total_px, bool eight_views, size_t first, size_t count) {
    if (total_px == 0 || count == 0) return;
    FlattenArgs a;
    /* (the pixels [first, first + count) of the batch: one view's, or all) */
    a.depth = maps + first; a.conf = maps + total_px + first; a.dz = maps + 2 * total_px + 2 * first; a.normal = maps + 4 * total_px + 3 * first;
    float* m1 = maps + 7 * total_px;
    a.depth1 = m1 + first; a.conf1 = m1 + total_px + first; a.dz1 = m1 + 2 * total_px + 2 * first; a.normal1 = m1 + 4 * total_px + 3 * first;
    a.views = imaps + first; a.upd = (int32_t*)(imaps + total_px + first);
    a.views1 = imaps + 2 * total_px + first; a.upd1 = (const int32_t*)(imaps + 3 * total_px + first);
    a.views_hi = eight_views ? imaps + 4 * total_px + first : nullptr; a.views1_hi = eight_views ? imaps + 5 * total_px + first : nullptr;
    a.n = (unsigned)count;
    hipLaunchKernelGGL(k_flatten, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, a);
}

void mi_launch_emit_changed(hipStream_t s, const float* maps, const uint32_t* imaps, size_t total_px, size_t first, size_t count,
                            int r0, unsigned view, unsigned stride, unsigned* d_count, unsigned cap, uint32_t* out) {
    if (total_px == 0 || count == 0) return;
    EmitArgs a;
    a.depth = maps + first; a.conf = maps + total_px + first; a.dz = maps + 2 * total_px + 2 * first; a.normal = maps + 4 * total_px + 3 * first;
    const float* m1 = maps + 7 * total_px;
    a.depth1 = m1 + first; a.conf1 = m1 + total_px + first; a.dz1 = m1 + 2 * total_px + 2 * first; a.normal1 = m1 + 4 * total_px + 3 * first;
    a.upd = (const int32_t*)(imaps + total_px + first); a.upd1 = (const int32_t*)(imaps + 3 * total_px + first);
    a.n = (unsigned)count; a.r0 = r0; a.view = view; a.stride = stride; a.count = d_count; a.cap = cap; a.out = out;
    hipLaunchKernelGGL(k_emit_changed, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, a);
}

void mi_launch_apply_seeds(hipStream_t s, const DevJob* jobs, const DevEntry* work, const DevResult* results,
                           unsigned n_work, DevCounters* counters, unsigned long long* seed_keys,
                           const unsigned* key_off, int seed_reopt, unsigned* seed_count) {
    if (n_work == 0) return;
    ApplyArgs a;
    a.seed_reopt = seed_reopt; a.seed_count = seed_count;
    a.jobs = jobs; a.work = work; a.results = results; a.spec = nullptr; a.items = nullptr; a.n_items = nullptr; a.n_work_ptr = nullptr; a.n_work = n_work; a.round = 0;
    a.min_work = 0; a.max_work = 0xFFFFFFFFu;
    a.counters = counters; a.seed_keys = seed_keys; a.key_off = key_off;
    a.phase = 0;
    hipLaunchKernelGGL(k_apply_seeds, dim3((n_work + 255) / 256), dim3(256), 0, s, a);
    a.phase = 1;
    hipLaunchKernelGGL(k_apply_seeds, dim3((n_work + 255) / 256), dim3(256), 0, s, a);
}

void mi_launch_round_report(hipStream_t s, const unsigned* a, int n_a, const unsigned* b, int n_b, const DevCounters* counters,
                            const DevJob* jobs, int n_jobs, unsigned* out_rw, DevCounters* out_hc, void* out_dyn, const unsigned* view_count) {
    hipLaunchKernelGGL(k_round_report, dim3(1), dim3(256), 0, s, a, n_a, b, n_b, counters, jobs, n_jobs, out_rw,
                       reinterpret_cast<unsigned*>(out_hc), static_cast<unsigned*>(out_dyn), view_count);
}
void mi_launch_follow_compact(hipStream_t s, const unsigned long long* mask, const unsigned* n_work_ptr, unsigned min_work, unsigned max_work,
                              unsigned ppw, unsigned lpp, unsigned* blk_sum, unsigned* out, unsigned* out_n) {
    FollowArgs a;
    a.mask = mask; a.n_work_ptr = n_work_ptr; a.min_work = min_work; a.max_work = max_work; a.ppw = ppw; a.lpp = lpp;
    a.blk_sum = blk_sum; a.out = out; a.out_n = out_n;
    hipLaunchKernelGGL(k_follow_count, dim3(MI_FOLLOW_BLOCKS), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_follow_scatter, dim3(MI_FOLLOW_BLOCKS), dim3(256), 0, s, a);
}
void mi_launch_region_mark(hipStream_t s, unsigned tag) { hipLaunchKernelGGL(k_region_mark, dim3(1), dim3(1), 0, s, tag); }
void mi_launch_unpack_jobs(hipStream_t s, const uint32_t* packed, unsigned words_per_job, DevJob* jobs, int n_jobs) {
    if (n_jobs <= 0 || words_per_job == 0) return;
    static_assert(sizeof(DevJob) % 4 == 0, "job records are copied word by word");
    hipLaunchKernelGGL(k_unpack_jobs, dim3(words_per_job > 768u ? 4u : (words_per_job + 255u) / 256u, (unsigned)n_jobs), dim3(256), 0, s,
                       packed, words_per_job, reinterpret_cast<uint32_t*>(jobs), (unsigned)(sizeof(DevJob) / 4));
}
unsigned mi_quad_words(void) { return MI_QUAD_WORDS; }      /* words per footprint record of this build (the host sizes the views by it) */
void mi_launch_quadify(hipStream_t s, const uint32_t* src, uint32_t* dst, int w, int h) {
    hipLaunchKernelGGL(k_quadify, dim3((w * h + 255) / 256), dim3(256), 0, s, src, (u32x4*)dst, w, h);
}

void mi_launch_pack_rgba(hipStream_t s, const uint8_t* src, uint32_t* dst, int n, int channels) {
    hipLaunchKernelGGL(k_pack_rgba, dim3((n + 255) / 256), dim3(256), 0, s, src, dst, n, channels);
}
void mi_launch_unpack_rgb(hipStream_t s, const uint32_t* src, uint8_t* dst, int n) {
    hipLaunchKernelGGL(k_unpack_rgb, dim3((n + 255) / 256), dim3(256), 0, s, src, dst, n);
}
void mi_launch_pyramid(hipStream_t s, const uint32_t* src, uint32_t* dst, int iw, int ih, int ow, int oh,
                       float w1, float w2, float w3) {
    hipLaunchKernelGGL(k_pyramid, dim3((ow + 63) / 64, (oh + 3) / 4), dim3(256), 0, s, src, dst, iw, ih, ow, oh, w1, w2, w3);
}
#endif

/* the launchers of this filter width (dmrecon_device.h: mi_device_api); host side only */
#if !defined(__HIP_DEVICE_COMPILE__)
extern const MiDeviceApi MI_CAT(mi_device_api_fw, MI_FW) = {MI_FW, launch_optimize, launch_patch_eval, launch_generate, launch_tail, launch_front, launch_optimize_spec};
#endif
