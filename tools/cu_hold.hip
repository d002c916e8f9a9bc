// Lab helper for the data-parallel CU-loss table (profiles/NOTES_r03.md): a kernel whose workgroups each take a whole CU's LDS (160 KB)
// and spin for a given wall-clock time -- what a communication library's persistent channel kernels do to the training step, without
// the communication.  Launched first, its n workgroups settle on n CUs; no kernel that needs LDS (GEMM, attention, LayerNorm backward,
// grouped weight gradients) can start a workgroup there until it ends.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/cu_hold.hip -o tools/bin/libcuhold.so
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ __launch_bounds__(1024) void k_hold(uint64_t ticks, int* sink) {
    extern __shared__ int lds[];
    const uint64_t t0 = wall_clock64();                  // constant 100 MHz counter
    lds[threadIdx.x] = (int)t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (ticks == 0) sink[0] = lds[(threadIdx.x + 1) & 1023];
}

extern "C" int cu_hold(int n_cus, float ms, void* stream) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_hold), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 1;
        attr = true;
    }
    static int* sink = nullptr;
    if (!sink && hipMalloc(&sink, 64) != hipSuccess) return 2;
    if (n_cus <= 0) return 0;
    hipLaunchKernelGGL(k_hold, dim3(n_cus), dim3(1024), 160 * 1024, (hipStream_t)stream, (uint64_t)(ms * 1e5), sink);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
