// Lab: which CUs does a stream created with hipExtStreamCreateWithCUMask use?  A kernel of 2048 short workgroups records (XCC id, HW id) of each;
// prints the number of distinct (SE, CU) per XCC for a few masks.   hipcc --offload-arch=gfx950 -O2 tools/cu_mask_lab.hip -o tools/bin/cu_mask_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <map>
#include <vector>
__global__ void k_probe(uint32_t* out) {
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) __builtin_amdgcn_s_sleep(8);          // 20 us: keep the CUs busy so that the grid spreads
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = xcc; out[blockIdx.x * 2 + 1] = hw; }
}
static void run(const char* what, const uint32_t* mask, int words) {
    hipStream_t st;
    hipError_t e = mask ? hipExtStreamCreateWithCUMask(&st, words, mask) : hipStreamCreate(&st);
    if (e != hipSuccess) { printf("%s: stream creation failed: %s\n", what, hipGetErrorString(e)); return; }
    const int n = 4096;
    uint32_t* d; hipMalloc(&d, n * 8);
    hipLaunchKernelGGL(k_probe, dim3(n), dim3(256), 0, st, d);
    hipStreamSynchronize(st);
    std::vector<uint32_t> h(n * 2); hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    std::map<int, std::set<int>> per;
    for (int i = 0; i < n; ++i) { const uint32_t hw = h[i * 2 + 1]; per[h[i * 2] & 0xf].insert(((hw >> 13) & 7) * 16 + ((hw >> 8) & 15)); }   // SE_ID [15:13], CU_ID [11:8]
    int tot = 0; printf("%-28s", what);
    for (auto& kv : per) { printf(" xcc%d:%2zu", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
    printf("  total %d CUs\n", tot);
    hipFree(d); hipStreamDestroy(st);
}
int main() {
    run("no mask", nullptr, 0);
    uint32_t a[8], b[8], c[8], dd[8];
    for (int i = 0; i < 8; ++i) { a[i] = 0x00FF00FFu; b[i] = 0xFF00FF00u; c[i] = i < 4 ? 0xFFFFFFFFu : 0u; dd[i] = 0x55555555u; }
    run("0x00FF00FF x 8", a, 8);
    run("0xFF00FF00 x 8", b, 8);
    run("low 128 bits", c, 8);
    run("0x55555555 x 8", dd, 8);
    uint32_t e4[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    run("4 words all ones", e4, 4);
    return 0;
}
