// Micro-benchmark (round 4): issue cost per wave-instruction of the instruction kinds the patch kernels are made of,
// on gfx950, at 1..3 wavefronts per SIMD.  Inline asm, eight independent instructions per loop trip.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate2.hip -o valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters, float a, float b) {
    float x[8]; double d[8]; int n[8]; long long q[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i + 0.25f; d[i] = x[i]; n[i] = threadIdx.x * 37 + i; q[i] = n[i]; }
    __shared__ float lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = i;
    __syncthreads();
    unsigned sh = 2;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            REP8(S)
#undef S
        } else if (MODE == 1) {
#define S(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d[i]) : "v"(d[(i + 1) & 7]), "v"(d[(i + 2) & 7]));
            REP8(S)
#undef S
        } else if (MODE == 2) {
#define S(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
            REP8(S)
#undef S
        } else if (MODE == 3) {
#define S(i) asm volatile("v_mad_i64_i32 %0, s[8:9], %1, %2, %0" : "+v"(q[i]) : "v"(n[i]), "v"(n[(i + 1) & 7]) : "s8", "s9");
            REP8(S)
#undef S
        } else if (MODE == 4) {
#define S(i) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(n[i]) : "v"(sh));
            REP8(S)
#undef S
        } else if (MODE == 5) {
#define S(i) asm volatile("v_floor_f32 %0, %0" : "+v"(x[i]));
            REP8(S)
#undef S
        } else if (MODE == 6) {
#define S(i) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(n[i]) : "v"(x[i]));
            REP8(S)
#undef S
        } else if (MODE == 7) {
#define S(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(n[i]) : "v"(n[(i + 1) & 7]), "v"(n[(i + 2) & 7]));
            REP8(S)
#undef S
        } else if (MODE == 8) {
#define S(i) asm volatile("v_lshl_add_u64 %0, %0, 4, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
            REP8(S)
#undef S
        } else if (MODE == 9) {
#define S(i) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
            REP8(S)
#undef S
        } else if (MODE == 10) {
#define S(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            REP8(S)
#undef S
        } else if (MODE == 11) {   // LUT-like LDS reads: address from data (1 KB table), 8 in flight, one wait
#define S(i) asm volatile("ds_read_b32 %0, %1" : "=v"(x[i]) : "v"((n[i] & 255) << 2));
            REP8(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (MODE == 12) {
#define S(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_and_b64 s[10:11], s[10:11], vcc" :: "v"(x[i]), "v"(a) : "vcc", "s10", "s11");
            REP8(S)
#undef S
        } else if (MODE == 13) {
#define S(i) asm volatile("v_fract_f32 %0, %0" : "+v"(x[i]));
            REP8(S)
#undef S
        } else if (MODE == 14) {
#define S(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            REP8(S)
#undef S
        } else if (MODE == 15) {
#define S(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            REP8(S)
#undef S
        } else if (MODE == 16) {
#define S(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(x[(i + 1) & 7]));
            REP8(S)
#undef S
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + (float)d[i] + (float)n[i] + (float)q[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int MODE>
double run(int waves_per_simd, int iters, float* d) {
    int grid = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, 10, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

#define RUN(M, NAME) { printf("[%s]\n", NAME); fflush(stdout); double ms = run<M>(w, iters, d); double r = 8.0 * iters * w / (ms * 1e-3); \
    printf("waves/SIMD=%d %-34s %8.3f ms  %6.2f cycles per wave-instruction at 2.4 GHz\n", w, NAME, ms, 2.4e9 / r); fflush(stdout); }

int main() {
    float* d; hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
    const int iters = 100000;
    for (int w = 1; w <= 3; ++w) {
        RUN(0, "v_fma_f32") RUN(1, "v_pk_fma_f32") RUN(15, "v_pk_mul_f32") RUN(2, "v_rcp_f32") RUN(3, "v_mad_i64_i32")
        RUN(4, "v_lshlrev_b32_sdwa (byte select)") RUN(5, "v_floor_f32") RUN(13, "v_fract_f32") RUN(6, "v_cvt_i32_f32")
        RUN(7, "v_mad_u32_u24") RUN(8, "v_lshl_add_u64") RUN(9, "v_add_f32_dpp quad_perm") RUN(16, "v_mov_b32_dpp row_shr:1")
        RUN(10, "v_med3_f32") RUN(14, "v_add_f64")
        RUN(11, "ds_read_b32 x8 + wait (1 KB table)")
    }
    return 0;
}
