/*
 * Harness helper (NOT on the product path, never linked into libmi_dmrecon.so): the synthetic scenes of mve_amd/synth.py
 * rendered on the GPU -- the same per-pixel ray / height-field intersection and band-limited texture as
 * synth_render.cc (the OpenMP renderer the golden fixtures were made with), one lane per pixel, in double precision.
 * The device's sin / cos differ from the host libm's in the last bits, so a byte of an image may differ by one from the
 * CPU renderer's: scenes rendered here are used where GPU path and checker read the SAME images anyway (the large
 * configurations of bench.py and tests/test_gpu_fullsize.py, the distinct-scenes variant of the bench) -- a 100-view
 * 4032 x 3024 scene takes 87 s on the box's 16-CPU quota and well under a second here.
 * Built as mve_amd/csrc/libmi_synth_gpu.so; loaded by mve_amd.synth via ctypes.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

struct mi_synth_params {        /* = synth_render.cc */
    double cam_pos[3];
    double rot[9];
    double ax, ay, cx, cy;
    double bump_amp;
    int32_t n_waves;
    int32_t width, height;
};
#define MI_SYNTH_MAX_WAVES 64
struct SynthTex { double fx[MI_SYNTH_MAX_WAVES], fy[MI_SYNTH_MAX_WAVES], mix[2 * MI_SYNTH_MAX_WAVES * 3]; };

__global__ __launch_bounds__(256) void k_synth_render(mi_synth_params p, const SynthTex* __restrict__ tex, uint8_t* __restrict__ out_rgb) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.width || y >= p.height) return;
    const double dxc = (x + 0.5 - p.cx) / p.ax, dyc = (y + 0.5 - p.cy) / p.ay;
    const double dx = p.rot[0] * dxc + p.rot[3] * dyc + p.rot[6];
    const double dy = p.rot[1] * dxc + p.rot[4] * dyc + p.rot[7];
    const double dz = p.rot[2] * dxc + p.rot[5] * dyc + p.rot[8];
    double t = (0.0 - p.cam_pos[2]) / dz;
    for (int it = 0; it < 12; ++it) {
        const double sx = p.cam_pos[0] + t * dx, sy = p.cam_pos[1] + t * dy;
        t = (p.bump_amp * sin(1.7 * sx + 0.3) * sin(2.3 * sy - 0.2) - p.cam_pos[2]) / dz;
    }
    const double sx = p.cam_pos[0] + t * dx, sy = p.cam_pos[1] + t * dy;
    double s[3] = {0.0, 0.0, 0.0};
    const int NW = p.n_waves;
    for (int k = 0; k < NW; ++k) {
        const double a = sx * tex->fx[k] + sy * tex->fy[k];
        double sn, cs;
        sincos(a, &sn, &cs);
        for (int c = 0; c < 3; ++c) s[c] += sn * tex->mix[k * 3 + c] + cs * tex->mix[(NW + k) * 3 + c];
    }
    for (int c = 0; c < 3; ++c) {
        const double v = s[c] < -1.0 ? -1.0 : (s[c] > 1.0 ? 1.0 : s[c]);
        out_rgb[((size_t)y * p.width + x) * 3 + c] = (uint8_t)floor(127.5 + 87.5 * v + 0.5);
    }
}

extern "C" {
/* out_rgb: height * width * 3 bytes of HOST memory (page-locked or not).  Returns 0, or a HIP error code. */
int mi_synth_render_gpu(const mi_synth_params* p, const double* fx, const double* fy, const double* mix, uint8_t* out_rgb) {
    if (!p || !out_rgb || p->n_waves < 1 || p->n_waves > MI_SYNTH_MAX_WAVES) return -1;
    static SynthTex* d_tex = nullptr;
    static uint8_t* d_img = nullptr;
    static size_t cap = 0;
    const size_t bytes = (size_t)p->width * p->height * 3;
    hipError_t e;
    if (!d_tex && (e = hipMalloc((void**)&d_tex, sizeof(SynthTex))) != hipSuccess) return (int)e;
    if (cap < bytes) {
        if (d_img) (void)hipFree(d_img);
        d_img = nullptr; cap = 0;
        if ((e = hipMalloc((void**)&d_img, bytes)) != hipSuccess) return (int)e;
        cap = bytes;
    }
    SynthTex h;
    for (int k = 0; k < p->n_waves; ++k) { h.fx[k] = fx[k]; h.fy[k] = fy[k]; }
    for (int k = 0; k < 2 * p->n_waves * 3; ++k) h.mix[k] = mix[k];
    if ((e = hipMemcpy(d_tex, &h, sizeof(h), hipMemcpyHostToDevice)) != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_synth_render, dim3((p->width + 63) / 64, (p->height + 3) / 4), dim3(256), 0, 0, *p, d_tex, d_img);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    if ((e = hipMemcpy(out_rgb, d_img, bytes, hipMemcpyDeviceToHost)) != hipSuccess) return (int)e;
    return 0;
}
}
