// Micro-benchmark: VALU issue rate of gfx950 per SIMD for scalar f32 FMA, packed FMA, rcp, cvt_ubyte, LDS reads,
// at 1..4 wavefronts per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    __shared__ float lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = i;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {           // 8 independent scalar FMAs
            x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
            x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
        } else if (MODE == 1) {    // one dependent chain of 8 FMAs
            x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b);
            x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b);
        } else if (MODE == 2) {    // 8 rcp
            x0 = __frcp_rn(x0) + a; x1 = __frcp_rn(x1) + a; x2 = __frcp_rn(x2) + a; x3 = __frcp_rn(x3) + a;
            x4 = __frcp_rn(x4) + a; x5 = __frcp_rn(x5) + a; x6 = __frcp_rn(x6) + a; x7 = __frcp_rn(x7) + a;
        } else if (MODE == 3) {    // 8 LDS reads (data-dependent address) + 8 adds
            x0 += lds[((int)x1) & 1023]; x1 += lds[((int)x2) & 1023]; x2 += lds[((int)x3) & 1023]; x3 += lds[((int)x4) & 1023];
            x4 += lds[((int)x5) & 1023]; x5 += lds[((int)x6) & 1023]; x6 += lds[((int)x7) & 1023]; x7 += lds[((int)x0) & 1023];
        } else if (MODE == 4) {    // 8 independent mul + add (unfused)
            x0 = __fadd_rn(__fmul_rn(x0, a), b); x1 = __fadd_rn(__fmul_rn(x1, a), b); x2 = __fadd_rn(__fmul_rn(x2, a), b); x3 = __fadd_rn(__fmul_rn(x3, a), b);
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
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

int main() {
    float* d; hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
    const int iters = 200000;
    const char* names[] = {"8 indep fma", "8 dep fma", "8 rcp+add", "8 lds+add", "4 mul+add unfused"};
    const int ops[] = {8, 8, 16, 16, 8};
    for (int w = 1; w <= 4; ++w) {
        double ms[5] = {run<0>(w, iters, d), run<1>(w, iters, d), run<2>(w, iters / 4, d) * 4, run<3>(w, iters / 4, d) * 4, run<4>(w, iters, d)};
        for (int m = 0; m < 5; ++m) {
            double inst_per_simd_per_s = (double)ops[m] * iters * w / (ms[m] * 1e-3);
            printf("waves/SIMD=%d %-18s %.3f ms  %.2f G wave-instr/s per SIMD  (%.2f cycles/instr at 2.4 GHz)\n", w, names[m], ms[m],
                   inst_per_simd_per_s / 1e9, 2.4e9 / inst_per_simd_per_s);
        }
    }
    return 0;
}
