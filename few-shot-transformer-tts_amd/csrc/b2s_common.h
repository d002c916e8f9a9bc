// Common device/host helpers for the byte2speech MI355X (gfx950) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define B2S_WAVE 64

// ------------------------------------------------------------------ error handling (host)
extern thread_local char g_b2s_err[512];
int b2s_fail(const char* file, int line, const char* fmt, ...);
#define B2S_CHECK(cond, ...) do { if (!(cond)) return b2s_fail(__FILE__, __LINE__, __VA_ARGS__); } while (0)
#define B2S_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return b2s_fail(__FILE__, __LINE__, "HIP error %d (%s) in %s", (int)e_, hipGetErrorString(e_), #expr); } while (0)
#define B2S_LAUNCH_CHECK() B2S_HIP(hipGetLastError())
#define B2S_TRY(expr) do { int r_ = (expr); if (r_) return r_; } while (0)

// ------------------------------------------------------------------ bf16 conversion
__device__ __host__ inline float bf2f(bf16_t x) {
    union { uint32_t u; float f; } c; c.u = ((uint32_t)x) << 16; return c.f;
}
// packed conversion on the device: one v_cvt_pk_bf16_f32 (round to nearest even, NaN -> quiet NaN) per pair
__device__ inline uint32_t f2bf2(float lo, float hi) {
    typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const f2_t v = {lo, hi};
    const bf2_t w = __builtin_convertvector(v, bf2_t);
    return __builtin_bit_cast(uint32_t, w);
}
__device__ __host__ inline bf16_t f2bf(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (bf16_t)(f2bf2(f, 0.f) & 0xffffu);
#endif
    union { uint32_t u; float f; } c; c.f = f;
    uint32_t u = c.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)0x7fc0;   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);                               // round to nearest even
    return (bf16_t)(u >> 16);
}

template <typename T> struct TT;
template <> struct TT<float> {
    static constexpr int VE = 4;      // elements per 16-byte vector
    __device__ static inline float ld(const float* p) { return *p; }
    __device__ static inline void st(float* p, float v) { *p = v; }
};
template <> struct TT<bf16_t> {
    static constexpr int VE = 8;
    __device__ static inline float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static inline void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// ------------------------------------------------------------------ counter-based dropout RNG
// keep(idx) is a pure function of (key, idx) so backward kernels regenerate the forward mask.
// lowbias32 integer hash; keep-rate is tested statistically (tests/test_gpu_ops.py).
__device__ __host__ inline uint32_t b2s_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
struct DropCfg {
    uint32_t key;      // per-op key (seed mixed with an op id)
    uint32_t thresh;   // drop if hash < thresh ; thresh = p * 2^32 ; 0 => no dropout
    float scale;       // 1 / (1 - p)
};
__device__ __host__ inline bool b2s_keep(const DropCfg& d, uint32_t idx) {
    return b2s_hash32(idx * 0x9E3779B1u + d.key) >= d.thresh;
}
// attention-weight dropout of the training kernels.  A weight row r (rows numbered (b H + h) Lq + q) has a seed, drawn with the full hash once
// per row; the four keys 4 kq .. 4 kq + 3 of the row share ONE mixing step and take their 16-bit fields from two multiplies of it:
//      seed(r)     = hash32(r * golden + key)
//      y(r, kq)    = x ^ (x >> 16),  x = seed(r) + kq * golden
//      word_j      = y * (j ? 0xC2B2AE35 : 0x85EBCA6B)                  j = (k >> 1) & 1
//      field(k)    = (k & 1) ? word_j >> 16 : word_j & 0xffff
//      keep(k)    <=> (int16) field(k) >= (thresh >> 16) - 32768       (the field read as a SIGNED 16-bit number: p = 0.1 drops 6553 of 65536 values)
// A lane of the attention kernels owns runs of four adjacent keys of one row (32x32 and 16x16 accumulator layouts alike): 4 integer operations per
// four weights instead of 4 x 8, and the signed compare is one saturating packed subtract + shift on the bf16 pair the weights are packed into.
// Statistics (keep rate, neighbour / row / diagonal correlations, row and column sums): tests/test_host_logic.py.
constexpr uint32_t B2S_WC0 = 0x85EBCA6Bu, B2S_WC1 = 0xC2B2AE35u;
__device__ __host__ inline uint32_t b2s_wseed(const DropCfg& d, uint32_t row) { return b2s_hash32(row * 0x9E3779B1u + d.key); }
__device__ __host__ inline uint32_t b2s_wmix(uint32_t seed, uint32_t kq) { const uint32_t x = seed + kq * 0x9E3779B1u; return x ^ (x >> 16); }
__device__ __host__ inline int b2s_wthresh(const DropCfg& d) { return (int)(d.thresh >> 16) - 32768; }
__device__ __host__ inline bool b2s_keep_w(const DropCfg& d, uint32_t row, uint32_t k) {
    const uint32_t w = b2s_wmix(b2s_wseed(d, row), k >> 2) * ((k & 2u) ? B2S_WC1 : B2S_WC0);
    return (int)(int16_t)(uint16_t)(w >> ((k & 1u) * 16u)) >= b2s_wthresh(d);
}
inline DropCfg make_drop(float p, uint64_t seed, uint32_t op_id) {
    DropCfg d;
    if (p <= 0.f) { d.key = 0; d.thresh = 0; d.scale = 1.f; return d; }
    d.key = b2s_hash32((uint32_t)seed ^ b2s_hash32((uint32_t)(seed >> 32) + 0x51ed270bU) ^ (op_id * 0x85ebca6bU + 0x1234567U));
    double t = (double)p * 4294967296.0;
    d.thresh = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
    d.scale = 1.f / (1.f - p);
    return d;
}

// ------------------------------------------------------------------ wave reductions
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
