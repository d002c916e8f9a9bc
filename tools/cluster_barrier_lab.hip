// Micro-benchmark for the decode loop's "cluster" question: could the self- and the cross-attention sublayer of a decoder
// layer be ONE kernel, with a flag barrier among the 8 workgroups (heads) of an utterance pair instead of a kernel boundary?
// 256 workgroups (one per CU) in 32 clusters of 8.  Per iteration every workgroup writes its partial slab (2 rows x 768 bf16 = 3 KB, the
// size k_df_attn writes), the cluster synchronises through a counter in device memory (release / acquire at agent scope), and every member
// reads the cluster's 8 slabs (what the next sublayer's LayerNorm prologue does).  Two placements:
//   cross-XCD  members = workgroups c*8 .. c*8+7: eight different XCDs -- the decode kernels' layout (head h on XCD h keeps the head's
//              590 KB weight slice in ONE L2);
//   same-XCD   members = workgroups c, c+32, ...: one XCD (each XCD's L2 would then have to hold all 8 heads' slices: 4.7 MB > 4 MB).
// Compare with the cost of the kernel boundary inside the frame's hipGraph (~1.5 us between kernels + ~2 us until a kernel's first loads
// have landed; profiles/NOTES_r02.md).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/cluster_barrier_lab.hip -o tools/bin/cluster_barrier_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ inline void cluster_barrier(unsigned* counter, unsigned& gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        gen += 8;
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) { __builtin_amdgcn_s_sleep(1); }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <bool SAME_XCD, bool BARRIER>
__global__ __launch_bounds__(512) void k(unsigned* counters, uint2* slabs, int iters, float* out) {
    // cluster / member of this workgroup
    const int wg = blockIdx.x;
    const int cl = SAME_XCD ? ((wg & 7) * 4 + ((wg >> 3) & 3)) : (wg >> 3);         // same-XCD: cluster = (xcd, slot): members share wg & 7
    const int mem = SAME_XCD ? (wg >> 5) : (wg & 7);
    unsigned gen = 0, gen2 = 0;
    float acc = 0.f;
    uint2* mine = slabs + ((size_t)cl * 8 + mem) * 384;       // 3 KB = 384 x 8 bytes
    for (int i = 0; i < iters; ++i) {
        if (threadIdx.x < 384) mine[threadIdx.x] = make_uint2((unsigned)i, (unsigned)wg);
        if (BARRIER) cluster_barrier(counters + cl * 64, gen);
        if (threadIdx.x < 384) {
            unsigned s = 0;
#pragma unroll
            for (int m = 0; m < 8; ++m) s += __builtin_nontemporal_load(&slabs[((size_t)cl * 8 + m) * 384 + threadIdx.x].x);
            acc += (float)s;
        }
        if (BARRIER) cluster_barrier(counters + cl * 64 + 32, gen2);      // (second barrier: nobody overwrites a slab that is still being read)
    }
    if (threadIdx.x == 0) out[wg] = acc;
}

int main() {
    unsigned* counters; uint2* slabs; float* out;
    hipMalloc(&counters, 32 * 64 * 4); hipMalloc(&slabs, 256 * 3072); hipMalloc(&out, 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    auto run = [&](const char* name, const void* fn) {
        hipMemset(counters, 0, 32 * 64 * 4);
        int it = iters;
        void* args[] = {&counters, &slabs, &it, &out};
        hipEventRecord(e0, 0);
        hipError_t e = hipLaunchCooperativeKernel(fn, dim3(256), dim3(512), args, 0, 0);
        hipEventRecord(e1, 0);
        hipError_t s = hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        float h[256]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-44s launch %s sync %s  %.3f us per iteration  (check %.0f; exact when synchronised: %.0f)\n", name, hipGetErrorString(e), hipGetErrorString(s),
               ms * 1e3 / iters, h[5], 384.0 * 8.0 * iters * (iters - 1) / 2);
    };
    run("no barrier (write + read only)", (const void*)k<false, false>);
    run("8-workgroup barrier, members on 8 XCDs", (const void*)k<false, true>);
    run("8-workgroup barrier, members on one XCD", (const void*)k<true, true>);
    return 0;
}
