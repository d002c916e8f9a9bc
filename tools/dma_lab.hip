// Micro-benchmark: how fast can one CU pull L2-resident data, by instruction form and by waves per CU?
//   mode 0: global_load_lds_dwordx4 (LDS-DMA)   mode 1: global_load_dwordx4 -> VGPR   mode 2: half-line pattern (16 rows x 64 B) LDS-DMA
// Every block of XCD x (blockIdx % 8) streams the same `region` bytes over and over (L2 hits after the first pass).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void k(const char* src, size_t region, int iters, int ld, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = src + (size_t)(blockIdx.x % 8) * region;
    // block-level stream position: each "tile" = 16 KB (4 waves x 4 instr x 1 KB); blocks of one XCD start at different tiles
    size_t pos = ((size_t)(blockIdx.x / 8) * 16384 * 7) & (region - 1);
    uint4 acc = make_uint4(0, 0, 0, 0);
    if (MODE == 1) {        // plain loads to VGPRs, DEPTH loads in flight per wave (compiler-scheduled waits)
        constexpr int NB = DEPTH / 4;
        for (int it = 0; it < iters; it += NB) {
            uint4 t[NB * 4];
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const size_t off = (pos + (size_t)j * 16384 + (size_t)(wave * 4 + i) * 1024 + lane * 16) & (region - 1);
                    t[j * 4 + i] = *reinterpret_cast<const uint4*>(base + off);
                }
#pragma unroll
            for (int j = 0; j < NB * 4; ++j) { acc.x ^= t[j].x; acc.y ^= t[j].y; acc.z ^= t[j].z; acc.w ^= t[j].w; }
            pos = (pos + (size_t)NB * 16384) & (region - 1);
        }
    } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            size_t off;
            if (MODE == 2) {   // 16 rows x 64 B per instruction, rows ld bytes apart
                const int q = wave * 4 + i;
                off = (pos + (size_t)(q * 16 + (lane >> 2)) * ld + (lane & 3) * 16) & (region - 1);
            } else {
                off = (pos + (size_t)(wave * 4 + i) * 1024 + lane * 16) & (region - 1);
            }
            const char* p = base + off;
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(smem + ((it & 3) * 16384) + (wave * 4 + i) * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");
        }
        pos = (pos + 16384) & (region - 1);
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1.f;
}

template <int MODE, int DEPTH>
void run(const char* name, const char* src, size_t region, int blocks, int ld, float* sink) {
    const int iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(blocks), dim3(256), 65536, 0, src, region, 200, ld, sink);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(blocks), dim3(256), 65536, 0, src, region, iters, ld, sink);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * iters * 16384;
    const int cus = blocks < 256 ? blocks : 256;
    printf("%-44s blocks=%4d region=%6zu KB depth=%2d : %8.2f TB/s  %6.1f B/clk/CU(@2.4GHz)\n", name, blocks, region >> 10, DEPTH,
           bytes / ms / 1e9, bytes / (ms * 1e-3) / cus / 2.4e9);
}

int main() {
    const size_t total = (size_t)8 * (64 << 20);
    char* src; hipMalloc(&src, total); hipMemset(src, 1, total);
    float* sink; hipMalloc(&sink, 4);
    for (size_t region : {(size_t)1 << 20, (size_t)4 << 20, (size_t)32 << 20}) {
        for (int blocks : {256, 512}) {
            run<0, 12>("LDS-DMA full lines", src, region, blocks, 0, sink);
            run<0, 28>("LDS-DMA full lines", src, region, blocks, 0, sink);
            run<2, 12>("LDS-DMA 16 rows x 64 B (ld 1536)", src, region, blocks, 1536, sink);
            run<1, 12>("global_load_dwordx4 -> VGPR", src, region, blocks, 0, sink);
            run<1, 24>("global_load_dwordx4 -> VGPR", src, region, blocks, 0, sink);
        }
    }
    run<0, 12>("LDS-DMA full lines, 64 blocks", src, 1 << 20, 64, 0, sink);
    run<1, 12>("VGPR loads, 64 blocks", src, 1 << 20, 64, 0, sink);
    run<0, 12>("LDS-DMA full lines, 768 blocks", src, 1 << 20, 768, 0, sink);
    return 0;
}
