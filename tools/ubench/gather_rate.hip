/*
 * gather_rate.hip -- how many 16-byte gathers a gfx950 compute unit's vector L1 serves, as a function of the number of
 * distinct 128-byte lines the 64 lanes of a gather touch and of where the lines are found (L1 / L2).  The measured ceiling
 * for the access pattern of the bulk patch kernels (k_optimize<Lay<1,4>, FAST>: one global_load_dwordx4 per lane and sample
 * from an 8-byte aligned address, five in flight per wavefront, three wavefronts per SIMD), next to which bench.py's figures
 * of that kernel -- 64 L1 accesses per gather instruction, profiles/r6_pmc.md -- can be put.
 *
 *   hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_rate tools/ubench/gather_rate.hip && /tmp/gather_rate
 *
 * One wavefront per workgroup, 12 per CU (3 per SIMD), each with a window of W lines of its own; a gather: lane l reads 16
 * bytes from line (base + (l % LINES) * 3) % W of the window at byte offset 8 * (l / LINES) -- 64 / LINES lanes share a line,
 * as the samples of neighbouring patches do; `base` moves pseudo-randomly from gather to gather.  12 windows of W lines per CU
 * against an L1 of 32 KB = 256 lines: W = 16 is served by the L1, W = 64 mostly by the L2 (4 MB per XCD: 384 windows of 8 KB).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
typedef const __attribute__((address_space(1))) u32x4_a4* gptr_t;

template <int LINES, int NV>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_gather(const uint32_t* __restrict__ buf, unsigned* __restrict__ sink, unsigned rounds, unsigned W) {
    const unsigned lane = threadIdx.x;
    const uint32_t* win = buf + (size_t)blockIdx.x * W * 32u;          /* this wavefront's window: W lines of 32 words */
    const unsigned myline = (lane % LINES) * 3u, myoff = ((lane / LINES) * 2u) % 28u;   /* words: 8-byte steps, 16 bytes stay in the line */
    unsigned acc = 0, h = blockIdx.x * 2654435761u + 12345u;
    for (unsigned r = 0; r < rounds; ++r) {
        u32x4 v[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {                                   /* five gathers in flight, as a row of the 5 x 5 window */
            h = h * 1664525u + 1013904223u;
            const unsigned line = ((h >> 8) + myline) % W;
            v[k] = *(gptr_t)(win + line * 32u + myoff);
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            /* NV dependent VALU instructions per gather that consume its value (the bulk kernel: ~107 per sample) */
            float f = __uint_as_float((v[k].x ^ v[k].y ^ v[k].z ^ v[k].w) & 0x3FFFFFFFu);
#pragma unroll
            for (int q = 0; q < NV; ++q) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f));
            acc ^= __float_as_uint(f);
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int LINES, int NV = 0>
static double run(const uint32_t* buf, unsigned* sink, unsigned grid, unsigned rounds, unsigned W) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k_gather<LINES, NV>), dim3(grid), dim3(64), 0, 0, buf, sink, rounds / 8, W);      /* warm */
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL((k_gather<LINES, NV>), dim3(grid), dim3(64), 0, 0, buf, sink, rounds, W);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return ms;
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const unsigned cus = (unsigned)p.multiProcessorCount, grid = cus * 12u, rounds = 4000;
    const unsigned Wmax = 256;
    uint32_t* buf; unsigned* sink;
    const size_t words = (size_t)grid * Wmax * 32u + 64;
    if (hipMalloc((void**)&buf, words * 4) != hipSuccess || hipMalloc((void**)&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 1, words * 4); (void)hipMemset(sink, 0, 64);
    const double ghz = 2.3;                                              /* shader clock under load (bench.py: shader_clock_mhz_measured) */
    printf("%u CUs, %u wavefronts (12 per CU), %u rounds x 5 gathers; cycles at %.1f GHz\n", cus, grid, rounds, ghz);
    printf("window lines/wave | distinct lines per gather | gathers per us per CU | shader cycles per gather per CU | lane accesses per cycle per CU\n");
    const unsigned Ws[] = {16, 32, 64, 128, 256};
    for (unsigned wi = 0; wi < 5; ++wi) {
        const unsigned W = Ws[wi];
        double ms[5];
        ms[0] = run<4>(buf, sink, grid, rounds, W); ms[1] = run<8>(buf, sink, grid, rounds, W); ms[2] = run<16>(buf, sink, grid, rounds, W);
        ms[3] = run<32>(buf, sink, grid, rounds, W); ms[4] = run<64>(buf, sink, grid, rounds, W);
        const int L[] = {4, 8, 16, 32, 64};
        for (int k = 0; k < 5; ++k) {
            const double gathers_per_cu = 12.0 * rounds * 5.0, us = ms[k] * 1000.0;
            const double per_us = gathers_per_cu / us, cyc = ghz * 1000.0 / per_us;
            printf("%17u | %25d | %21.2f | %31.1f | %.3f\n", W, L[k], per_us, cyc, 64.0 / cyc);
        }
    }
    /* the same gathers (16 distinct lines, windows that the L1 / L2 serve) with VALU work behind every gather: do the two overlap? */
    printf("\nVALU instructions per gather (dependent v_fma_f32, consuming the gathered value) | window 32 lines: cycles per gather per CU | window 256 lines | VALU floor (3 wavefronts x 4 cycles x n / 3 gathers... = 4 n cycles per gather per SIMD, / 4 SIMDs x 12 = n x 12 / 4... )\n");
    {
        const double t0a = run<16, 0>(buf, sink, grid, rounds, 32), t0b = run<16, 0>(buf, sink, grid, rounds, 256);
        const double t1a = run<16, 50>(buf, sink, grid, rounds, 32), t1b = run<16, 50>(buf, sink, grid, rounds, 256);
        const double t2a = run<16, 100>(buf, sink, grid, rounds, 32), t2b = run<16, 100>(buf, sink, grid, rounds, 256);
        const double t3a = run<16, 150>(buf, sink, grid, rounds, 32), t3b = run<16, 150>(buf, sink, grid, rounds, 256);
        const double t4a = run<16, 200>(buf, sink, grid, rounds, 32), t4b = run<16, 200>(buf, sink, grid, rounds, 256);
        const double ta[] = {t0a, t1a, t2a, t3a, t4a}, tb[] = {t0b, t1b, t2b, t3b, t4b};
        const int nv[] = {0, 50, 100, 150, 200};
        for (int k = 0; k < 5; ++k) {
            const double g = 12.0 * rounds * 5.0;
            /* VALU floor: 12 wavefronts x nv x 4 cycles on 4 SIMDs per 12 gathers = nv x 4 x 3 / 12 x ... = nv cycles per gather per CU */
            printf("%5d | %8.1f | %8.1f | VALU floor %d cycles per gather per CU\n", nv[k], ghz * 1e6 * ta[k] / g, ghz * 1e6 * tb[k] / g, nv[k]);
        }
    }
    return 0;
}
