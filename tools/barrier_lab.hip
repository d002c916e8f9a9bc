// Micro-benchmark: cost of a device-wide barrier among co-resident workgroups (one per CU) -- the price of a phase
// boundary inside a persistent decode kernel, to be compared with ~7 us per kernel node in a hipGraph.
#include <hip/hip_runtime.h>
#include <cstdio>
#ifndef SLEEP
#define SLEEP
#endif

__device__ inline void grid_barrier(unsigned* counter, unsigned& gen, unsigned nblocks) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        gen += nblocks;
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) { SLEEP }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

__global__ __launch_bounds__(512) void k(unsigned* counter, float* data, int iters) {
    unsigned gen = 0;
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        // a little phase work: every block writes one value, everybody reads a neighbour's after the barrier
        if (threadIdx.x == 0) data[blockIdx.x] = (float)i;
        grid_barrier(counter, gen, gridDim.x);
        acc += data[(blockIdx.x + 1) % gridDim.x];
    }
    if (threadIdx.x == 0 && acc == -1.f) data[0] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) data[gridDim.x] = acc;
}

int main() {
    unsigned* counter; float* data;
    hipMalloc(&counter, 4); hipMalloc(&data, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nb : {64, 128, 256}) {
        const int iters = 2000;
        hipMemset(counter, 0, 4);
        void* args[] = {&counter, &data, (void*)&iters};
        hipEventRecord(e0, 0);
        hipError_t e = hipLaunchCooperativeKernel((const void*)k, dim3(nb), dim3(512), args, 0, 0);
        hipEventRecord(e1, 0);
        hipError_t s = hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        float h; hipMemcpy(&h, data + nb, 4, hipMemcpyDeviceToHost);
        printf("blocks %3d: launch %s sync %s  %.3f us per barrier  (check %.0f == %.0f)\n", nb, hipGetErrorString(e), hipGetErrorString(s),
               ms * 1e3 / iters, h, (float)iters * (iters - 1) / 2);
    }
    return 0;
}
