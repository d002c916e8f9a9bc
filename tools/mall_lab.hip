// Does a buffer that was touched a moment ago stream faster the second time (memory-side Infinity Cache, 256 MB), and is a sparse touch
// (4 bytes per 128-byte line) enough to bring it there?  Decides whether the decode loop's next-layer K/V cache can be pulled in under
// the latency-bound kernels that precede its self-attention.
//   hipcc --offload-arch=gfx950 -O3 tools/mall_lab.hip -o tools/bin/mall_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(512) void k_stream(const uint4* __restrict__ p, long n16, float* sink) {       // full read, 16 B per lane
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 512 + threadIdx.x; i < n16; i += (long)gridDim.x * 512) { uint4 v = p[i]; acc += __uint_as_float(v.x ^ v.y ^ v.z ^ v.w); }
    if (acc == 123.456f) sink[0] = acc;
}
template <int U>
__global__ __launch_bounds__(512) void k_stream_u(const uint4* __restrict__ p, long n16, float* sink) {     // U independent 16-byte loads in flight per lane
    float acc = 0.f;
    const long stride = (long)gridDim.x * 512;
    long i = (long)blockIdx.x * 512 + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += __uint_as_float(v[u].x ^ v[u].y ^ v[u].z ^ v[u].w);
    }
    for (; i < n16; i += stride) { uint4 v = p[i]; acc += __uint_as_float(v.x ^ v.y ^ v.z ^ v.w); }
    if (acc == 123.456f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_touch(const unsigned* __restrict__ p, long nlines, float* sink) {  // 4 B per 128-byte line
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nlines; i += (long)gridDim.x * 256) acc += __uint_as_float(p[i * 32]);
    if (acc == 123.456f) sink[0] = acc;
}
int main() {
    const long MB = 1 << 20;
    float* sink; CK(hipMalloc(&sink, 64));
    char* big; CK(hipMalloc(&big, 1536 * MB)); CK(hipMemset(big, 1, 1536 * MB));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_stream = [&](char* p, long bytes, int wgs) { float ms; (void)hipEventRecord(e0); hipLaunchKernelGGL(k_stream, dim3(wgs), dim3(512), 0, 0, (const uint4*)p, bytes / 16, sink); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f; };
    auto time_touch = [&](char* p, long bytes, int wgs) { float ms; (void)hipEventRecord(e0); hipLaunchKernelGGL(k_touch, dim3(wgs), dim3(256), 0, 0, (const unsigned*)p, bytes / 128, sink); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f; };
    auto time_u = [&](char* p, long bytes, int wgs, int u) { float ms; (void)hipEventRecord(e0);
        if (u == 4) hipLaunchKernelGGL(k_stream_u<4>, dim3(wgs), dim3(512), 0, 0, (const uint4*)p, bytes / 16, sink);
        else hipLaunchKernelGGL(k_stream_u<8>, dim3(wgs), dim3(512), 0, 0, (const uint4*)p, bytes / 16, sink);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f; };
    for (int u : {4, 8}) for (int wgs : {256, 512, 1024}) {
        char* buf = big; char* flush = big + 512 * MB; long bytes = 98 * MB; float cold = 0, warm = 0;
        for (int rep = 0; rep < 3; ++rep) { time_stream(flush, 1024 * MB, 1024); cold = time_u(buf, bytes, wgs, u); warm = time_u(buf, bytes, wgs, u); }
        printf("98 MB, %d loads in flight per lane, %4d workgroups: cold %.1f us (%.2f TB/s) | again %.1f us (%.2f TB/s)\n", u, wgs, cold, bytes / cold / 1e6, warm, bytes / warm / 1e6);
    }
    for (long sz : {32L, 98L, 196L}) {
        char* buf = big;                       // the first sz MB; the flush streams the 1 GB behind it
        char* flush = big + 512 * MB;
        long bytes = sz * MB;
        float cold = 0, warm = 0, pre = 0, after_pre = 0, pre_small = 0, after_small = 0;
        for (int rep = 0; rep < 3; ++rep) {
            time_stream(flush, 1024 * MB, 1024);
            cold = time_stream(buf, bytes, 256);
            warm = time_stream(buf, bytes, 256);
            time_stream(flush, 1024 * MB, 1024);
            pre = time_touch(buf, bytes, 1024);
            after_pre = time_stream(buf, bytes, 256);
            time_stream(flush, 1024 * MB, 1024);
            pre_small = time_touch(buf, bytes, 64);          // a few spare waves' worth
            after_small = time_stream(buf, bytes, 256);
        }
        printf("%4ld MB: cold %.1f us (%.2f TB/s) | again %.1f us (%.2f TB/s) | touch(1024 wg) %.1f us then stream %.1f us (%.2f TB/s) | touch(64 wg) %.1f us then stream %.1f us\n",
               sz, cold, bytes / cold / 1e6, warm, bytes / warm / 1e6, pre, after_pre, bytes / after_pre / 1e6, pre_small, after_small);
    }
    return 0;
}
