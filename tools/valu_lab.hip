// Micro-benchmark: issue cost of the VALU instructions the attention kernels' dropout hash and softmax are made of (one wave per SIMD and four
// waves per SIMD; 8 independent chains per lane so that dependent-issue latency does not show).  Not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 128
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, long long* cyc) {
    uint32_t x[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { x[i] = seed + threadIdx.x * 8 + i; f[i] = (float)x[i] * 1e-9f; }
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int ii = 0; ii < 64; ++ii) {
            const int i = ii & 7;
            if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(0x7feb352du));
            if (OP == 1) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(0xfeb352du));
            if (OP == 2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(0x7feb352du));
            if (OP == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
            if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(1.0001f));
            if (OP == 5) asm volatile("v_lshrrev_b32 %0, 15, %0" : "+v"(x[i]));
            if (OP == 6) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x[i]) : "v"(0xfeb352du));
            if (OP == 7) asm volatile("v_cmp_ge_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, 0, %0, vcc" : "+v"(x[i]) : "v"(0x1feb352du) : "vcc");
            if (OP == 8) asm volatile("v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(x[i]));
            if (OP == 9) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(1.0001f));
            if (OP == 10) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&f[i & 6])) : "v"(1.0001));
            if (OP == 11) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(f[i]));
            if (OP == 12) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(0x7feb352du));
            if (OP == 13) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(0x7feb352du));
            if (OP == 14) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(x[i]) : "v"(0x19991999u));
            if (OP == 15) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(1.0001f));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    uint32_t acc = 0;
    for (int i = 0; i < 8; ++i) acc += x[i] + __float_as_uint(f[i]);
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> void run(const char* name, uint32_t* out, long long* cyc) {
    for (int wg = 1; wg <= 4; wg *= 2) {                          // 256 threads = 1 wave per SIMD; 4 workgroups per CU = 4 waves per SIMD
        hipLaunchKernelGGL(k<OP>, dim3(256 * wg), dim3(256), 0, 0, out, 1u, cyc);
        hipDeviceSynchronize();
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-22s %d wave(s)/SIMD: %.2f memtime ticks per instruction per wave\n", name, wg, (double)c / (REP * 64));
    }
}
int main() {
    uint32_t* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4 * 4); hipMalloc(&cyc, 8);
    run<0>("v_mul_lo_u32", out, cyc); run<12>("v_mul_hi_u32", out, cyc); run<1>("v_mul_u32_u24", out, cyc); run<6>("v_mad_u32_u24", out, cyc);
    run<2>("v_xor_b32", out, cyc); run<13>("v_add_u32", out, cyc); run<5>("v_lshrrev_b32", out, cyc); run<8>("v_xor_b32_sdwa", out, cyc);
    run<7>("v_cmp + v_cndmask", out, cyc); run<14>("v_pk_sub_u16 clamp", out, cyc);
    run<3>("v_exp_f32", out, cyc); run<4>("v_fma_f32", out, cyc); run<15>("v_add_f32", out, cyc); run<9>("v_max3_f32", out, cyc);
    run<10>("v_pk_mul_f32", out, cyc); run<11>("v_cvt_pk_bf16_f32", out, cyc);
    return 0;
}
