// Micro-benchmark of the sampler's gather pattern against the gfx950 L1 (TCP): a wavefront = 16 neighbouring
// patches x 4 views (quad layout lane = patch*4 + view, or view-major lane = view*16 + patch); per sample every lane
// reads the 2x2 texel block of its patch point (two row gathers); 25 samples (5x5) per pass, several passes over the
// same window.  Variants: 8-byte row gathers as the kernel issues them (4-byte aligned), 8-byte gathers forced to
// 8-byte alignment, four dword gathers.  Reported: ns per wave-level sample and (under rocprofv3 --pmc
// TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum) requests per sample.
// Build: hipcc --offload-arch=gfx950 -O3 gather_l1.hip -o gather_l1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32;
struct __attribute__((aligned(4))) u32x2a4 { u32 x, y; };
struct __attribute__((aligned(8))) u32x2a8 { u32 x, y; };

template <int MODE, int VIEWMAJOR>
__global__ __launch_bounds__(64) void k(const u32* img, int W, int H, int n_views, int passes, int step_fp8, int scatter, u32* out) {
    const int lane = threadIdx.x;
    const int p = VIEWMAJOR ? (lane & 15) : (lane >> 2), v = VIEWMAJOR ? (lane >> 4) : (lane & 3);
    const u32* base = img + (size_t)(v % n_views) * W * H;
    // window origin of this wavefront; `scatter` spreads the 16 patches of a wave over the image instead of a row
    unsigned h = blockIdx.x * 2654435761u;
    int x0 = 8 + (h % (unsigned)(W - 200)), y0 = 8 + ((h >> 12) % (unsigned)(H - 40));
    if (scatter) { unsigned g = (blockIdx.x * 16 + p) * 2246822519u; x0 = 8 + (g % (unsigned)(W - 200)); y0 = 8 + ((g >> 12) % (unsigned)(H - 40)); }
    u32 acc = 0;
    for (int pass = 0; pass < passes; ++pass)
        for (int i = 0; i < 25; ++i) {
            const int di = i % 5, dj = i / 5;
            const int x = x0 + (((scatter ? 0 : p) + di) * step_fp8 >> 8), y = y0 + ((dj * step_fp8) >> 8);
            const u32* r0 = base + (size_t)y * W + x;
            if (MODE == 0) {           // as the kernel: dwordx2, 4-byte aligned
                const u32x2a4 a = *(const u32x2a4*)r0, b = *(const u32x2a4*)(r0 + W);
                acc += a.x + a.y + b.x + b.y;
            } else if (MODE == 1) {    // dwordx2 forced to 8-byte alignment (reads the aligned pair containing x)
                const u32* q = (const u32*)((size_t)r0 & ~(size_t)7);
                const u32x2a8 a = *(const u32x2a8*)q, b = *(const u32x2a8*)(q + W + (W & 1));
                acc += a.x + a.y + b.x + b.y;
            } else {                   // four dword gathers
                const u32* r1 = r0 + 1;
                asm volatile("" : "+v"(r1));
                acc += r0[0] + r1[0] + r0[W] + r1[W];
            }
        }
    out[blockIdx.x * 64 + lane] = acc;
}

template <int MODE, int VM>
float run(const u32* img, int W, int H, int nv, int grid, int passes, int step, int scatter, u32* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, VM>), dim3(grid), dim3(64), 0, 0, img, W, H, nv, 1, step, scatter, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, VM>), dim3(grid), dim3(64), 0, 0, img, W, H, nv, passes, step, scatter, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main(int argc, char** argv) {
    const int W = 960, H = 540, NV = 20;
    u32* img; hipMalloc(&img, (size_t)NV * W * H * 4); hipMemset(img, 1, (size_t)NV * W * H * 4);
    u32* out; hipMalloc(&out, 65536 * 64 * 4);
    const int passes = 24;
    for (int scatter = 0; scatter < 2; ++scatter)
        for (int step = 256; step <= 512; step += 256)
            for (int wps = 1; wps <= 3; wps += 2) {
                const int grid = 256 * 4 * wps * 4;     // 4 generations of resident wavefronts
                float t[6] = { run<0, 0>(img, W, H, NV, grid, passes, step, scatter, out), run<1, 0>(img, W, H, NV, grid, passes, step, scatter, out),
                               run<2, 0>(img, W, H, NV, grid, passes, step, scatter, out), run<0, 1>(img, W, H, NV, grid, passes, step, scatter, out),
                               run<1, 1>(img, W, H, NV, grid, passes, step, scatter, out), run<2, 1>(img, W, H, NV, grid, passes, step, scatter, out) };
                const double samples = (double)grid * passes * 25;
                printf("scatter=%d step=%.1f texel waves/SIMD~%d : ns per wave-sample (per SIMD slot)  quad: x2 %.1f  x2-aligned %.1f  4xdword %.1f | view-major: x2 %.1f  x2-aligned %.1f  4xdword %.1f\n",
                       scatter, step / 256.0, wps, t[0] * 1e6 / samples * (1024 * wps), t[1] * 1e6 / samples * (1024 * wps), t[2] * 1e6 / samples * (1024 * wps),
                       t[3] * 1e6 / samples * (1024 * wps), t[4] * 1e6 / samples * (1024 * wps), t[5] * 1e6 / samples * (1024 * wps));
            }
    return 0;
}
