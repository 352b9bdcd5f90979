/*
 * pmc_calib.hip -- known byte counts for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 in the access
 * patterns of the dmrecon kernels (MI355X_MICROARCH.md: "FETCH_SIZE reports 1/2 of a wide coalesced read; other
 * widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
 *
 *   k_stream16   every lane reads consecutive 16-byte records (coalesced dwordx4): the guide's case
 *   k_gather16   every lane reads ONE 16-byte record at a pseudo-random position (the footprint gathers of
 *                k_optimize / k_tail: global_load_dwordx4 at scattered addresses)
 *   k_gather16r5 every lane reads FIVE consecutive 16-byte records at a random position (a row of the 5x5 window)
 *   k_write4     every lane writes 4 bytes at a random position (the state-map writes of k_apply / k_tail)
 *   k_write16s   coalesced 16-byte stores
 * The buffer (1 GiB) is four times the Infinity Cache, every record is touched at most once per kernel, so the
 * bytes below really come from / go to HBM.  Build + run:  tools/pmc_calib.sh  (prints bytes per kernel; the
 * counters come from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE around the same binary).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t mix(uint64_t x) {           /* bijective on [0, 2^k): odd multiplier + xorshift */
    x *= 0x9E3779B97F4A7C15ull;
    return x;
}

__global__ void k_stream16(const u32x4* __restrict__ src, unsigned* __restrict__ sink, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32x4 v = src[i];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) sink[0] = 1;
}

__global__ void k_gather16(const u32x4* __restrict__ src, unsigned* __restrict__ sink, uint64_t n, unsigned bits) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t j = (mix(i) >> (64 - bits));                  /* a permutation of the 2^bits records as i runs over them */
    u32x4 v = src[j];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) sink[0] = 1;
}

__global__ void k_gather16r5(const u32x4* __restrict__ src, unsigned* __restrict__ sink, uint64_t n, unsigned bits) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t j = (mix(i) >> (64 - bits)) * 8;              /* groups of 8 records (128 B), 5 of them read */
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) { u32x4 v = src[j + k]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = 1;
}

__global__ void k_write4(unsigned* __restrict__ dst, uint64_t n, unsigned bits) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dst[(mix(i) >> (64 - bits)) * 4] = (unsigned)i;              /* one dword per 16-byte record */
}

__global__ void k_write16s(u32x4* __restrict__ dst, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32x4 v; v.x = v.y = v.z = v.w = (unsigned)i;
    dst[i] = v;
}

int main() {
    const unsigned bits = 26;                                    /* 2^26 records x 16 B = 1 GiB */
    const uint64_t nrec = 1ull << bits;
    u32x4* buf; unsigned* sink;
    if (hipMalloc((void**)&buf, nrec * 16) != hipSuccess || hipMalloc((void**)&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 1, nrec * 16); (void)hipMemset(sink, 0, 64);
    (void)hipDeviceSynchronize();
    const int T = 256;
    const uint64_t n_g = nrec / 4;                               /* gathers: a quarter of the records, each once */
    const uint64_t n_r5 = nrec / 8;                              /* row gathers: every 128-byte group once */
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_stream16, dim3((unsigned)((nrec + T - 1) / T)), dim3(T), 0, 0, buf, sink, nrec);
        hipLaunchKernelGGL(k_gather16, dim3((unsigned)((n_g + T - 1) / T)), dim3(T), 0, 0, buf, sink, n_g, bits);
        hipLaunchKernelGGL(k_gather16r5, dim3((unsigned)((n_r5 + T - 1) / T)), dim3(T), 0, 0, buf, sink, n_r5, bits - 3);
        hipLaunchKernelGGL(k_write4, dim3((unsigned)((n_g + T - 1) / T)), dim3(T), 0, 0, (unsigned*)buf, n_g, bits);
        hipLaunchKernelGGL(k_write16s, dim3((unsigned)((nrec + T - 1) / T)), dim3(T), 0, 0, buf, nrec);
        (void)hipDeviceSynchronize();
    }
    printf("known bytes per launch (useful): k_stream16 read %llu ; k_gather16 read %llu (64-B sectors touched: %llu, 128-B lines: %llu) ; "
           "k_gather16r5 read %llu (128-B lines: %llu) ; k_write4 written %llu (64-B sectors dirtied: up to %llu) ; k_write16s written %llu\n",
           (unsigned long long)(nrec * 16), (unsigned long long)(n_g * 16), (unsigned long long)(n_g * 64), (unsigned long long)(n_g * 128),
           (unsigned long long)(n_r5 * 80), (unsigned long long)(n_r5 * 128), (unsigned long long)(n_g * 4), (unsigned long long)(n_g * 64),
           (unsigned long long)(nrec * 16));
    return 0;
}
