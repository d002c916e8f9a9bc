// What does the LayerNorm backward's traffic mix cost as a plain element-wise kernel?  3 input streams (bf16 dY 12.5 MB, fp32 x 25 MB,
// fp32 residual gradient 25 MB) and 2 output streams (fp32 dx 25 MB, bf16 dY' 12.5 MB) at M = 8148, D = 768: 100 MB per launch.
//   hipcc --offload-arch=gfx950 -O3 tools/mix_lab.hip -o tools/bin/mix_lab
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
template <int U>
__global__ __launch_bounds__(256) void k_mix(const u2* __restrict__ dy, const f4* __restrict__ x, const f4* __restrict__ dr, f4* __restrict__ dx, u2* __restrict__ dy2, long n4) {
    const long stride = (long)gridDim.x * 256;
    // (launch only with gridDim.x * 256 * U <= n4: the remainder loop is omitted)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n4; i += U * stride) {
        u2 a[U]; f4 b[U], c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { a[u] = dy[i + u * stride]; b[u] = x[i + u * stride]; c[u] = dr[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f4 o = b[u] * 0.5f + c[u]; o[0] += __uint_as_float(a[u][0] << 16); o[2] += __uint_as_float(a[u][1] << 16);
            dx[i + u * stride] = o;
            u2 w; w[0] = __float_as_uint(o[0]) >> 16 | (__float_as_uint(o[1]) & 0xffff0000u); w[1] = __float_as_uint(o[2]) >> 16 | (__float_as_uint(o[3]) & 0xffff0000u);
            dy2[i + u * stride] = w;
        }
    }
}
int main() {
    const long M = 8148, D = 768, n4 = M * D / 4;
    void *dy, *x, *dr, *dx, *dy2, *flush;
    hipMalloc(&dy, n4 * 8); hipMalloc(&x, n4 * 16); hipMalloc(&dr, n4 * 16); hipMalloc(&dx, n4 * 16); hipMalloc(&dy2, n4 * 8); hipMalloc(&flush, 1L << 30);
    hipMemset(dy, 0, n4 * 8); hipMemset(x, 0, n4 * 16); hipMemset(dr, 0, n4 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {512, 1024, 2048, 4096}) for (int u : {1, 2, 4}) {
        if ((long)wgs * 256 * u * 2 > n4) continue;                     // at least two full passes per thread
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipMemsetAsync(flush, rep, 1L << 30, 0);
            hipEventRecord(e0);
            if (u == 1) hipLaunchKernelGGL(k_mix<1>, dim3(wgs), dim3(256), 0, 0, (const u2*)dy, (const f4*)x, (const f4*)dr, (f4*)dx, (u2*)dy2, n4);
            else if (u == 2) hipLaunchKernelGGL(k_mix<2>, dim3(wgs), dim3(256), 0, 0, (const u2*)dy, (const f4*)x, (const f4*)dr, (f4*)dx, (u2*)dy2, n4);
            else hipLaunchKernelGGL(k_mix<4>, dim3(wgs), dim3(256), 0, 0, (const u2*)dy, (const f4*)x, (const f4*)dr, (f4*)dx, (u2*)dy2, n4);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%4d workgroups, %d iterations in flight: %.1f us = %.2f TB/s\n", wgs, u, best * 1e3, 100.1e6 / (best * 1e-3) / 1e12);
    }
    return 0;
}
