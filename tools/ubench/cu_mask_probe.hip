// Which compute units does a CU-masked stream use?  (hipExtStreamCreateWithCUMask: "bit i = CU i" says nothing about
// how the bits map onto the eight XCDs of an MI355X.)  Every workgroup of a grid that oversubscribes the GPU records
// the XCD, shader engine and CU it ran on; the host prints the set per mask.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/cu_mask_probe.hip -o build/cu_mask_probe && build/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
#include <map>

__global__ void k_where(uint32_t* out) {
    const unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | 4);           // HW_REG_HW_ID, 16 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;   // HW_REG_XCC_ID
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) { }                                  // 20 us: keep the CU busy so that the grid spreads
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 16) | (hw & 0xFFFFu);
}

static void run(const char* name, hipStream_t s, uint32_t* d, std::vector<uint32_t>& h) {
    hipLaunchKernelGGL(k_where, dim3((unsigned)h.size()), dim3(256), 0, s, d);
    hipMemcpyAsync(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    std::map<unsigned, std::set<unsigned> > per_xcc;
    for (uint32_t v : h) {
        const unsigned xcc = v >> 16, cu = (v >> 8) & 15u, sh = (v >> 12) & 1u, se = (v >> 13) & 7u;
        per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
    }
    size_t total = 0;
    printf("%s:", name);
    for (auto& kv : per_xcc) { printf(" xcc%u=%zu", kv.first, kv.second.size()); total += kv.second.size(); }
    printf("  total %zu CUs\n", total);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount, words = (ncu + 31) / 32;
    printf("multiProcessorCount %d\n", ncu);
    std::vector<uint32_t> h(8192);
    uint32_t* d; hipMalloc(&d, h.size() * 4);
    hipStream_t s0; hipStreamCreate(&s0);
    run("no mask", s0, d, h);
    struct { const char* name; int lo, hi, step; } cases[] = {
        {"bits 0..31 cleared", 0, 32, 1}, {"bits 0..63 cleared", 0, 64, 1}, {"every 8th bit cleared", 0, ncu, 8},
        {"bits 224..255 cleared", 224, 256, 1}, {"only bits 0..31 set", -1, 32, 1}};
    for (auto& c : cases) {
        std::vector<uint32_t> m(words, c.lo < 0 ? 0u : 0xFFFFFFFFu);
        if (c.lo < 0) for (int i = 0; i < c.hi; ++i) m[i / 32] |= 1u << (i % 32);
        else for (int i = c.lo; i < c.hi; i += c.step) m[i / 32] &= ~(1u << (i % 32));
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, m.data());
        if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", c.name, hipGetErrorString(e)); continue; }
        run(c.name, s, d, h);
        hipStreamDestroy(s);
    }
    return 0;
}
