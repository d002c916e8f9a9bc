// bf16 MFMA GEMM, second-generation main loop for gfx950: operands go HBM -> LDS directly with
// global_load_lds_dwordx4 (LDS-DMA, no staging VGPRs, no ds_write pass), LDS tiles are dense and XOR-swizzled so
// that both the K-contiguous fragment reads (ds_read_b128) and the reduction-major fragment reads
// (ds_read_b64_tr_b16) are bank-conflict free, and the next K tile's DMA is in flight while the current tile's
// MFMAs run (one s_barrier per K tile).  Same contract / epilogues as gemm_kernel in gemm.hip.
//
// The LDS-DMA writes lane-linearly (wave base + lane*16 B), so the swizzle is applied on the SOURCE side: lane L
// of a DMA instruction owns LDS slot L and fetches the global chunk whose swizzled position is L.
//   N tile  [128 rows][32 k]   (64 B rows, 4 chunks):   physical chunk = chunk ^ ((-(row >> 2)) & 3)
//   T tile  [32 k][128 cols]   (256 B rows, 16 chunks): physical chunk = chunk ^ (((k & 3) | ((k >> 1) & 4)) << 1)
// K is walked in BK = 32 steps through an NSTAGE-deep LDS ring: NSTAGE-1 tiles are in flight, each wave waits only for
// its own DMA of the tile it is about to use (counted s_waitcnt vmcnt, never 0 in steady state) and one raw s_barrier
// per tile publishes it to the other waves and frees the oldest slot.
// (The round-1 ablation switches that attributed the main loop's time -- no DMA / no wait / no barrier / no LDS reads / L1-hot
// DMA / whole-line fetch / register loads, results in profiles/README.md -- lived here as #ifdefs; they were removed from the
// production kernel and can be re-created from commit a6102c4 with tools/gemm_lab.hip.)
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include "gemm.h"
#include "gemm_epi.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int TILE = 128 * 32;            // elements per operand tile (8 KB)
constexpr int NSTAGE = 4;

typedef __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) bf16x4_t* lds_b64_ptr;

__device__ inline int swz_t(int k) { return ((k & 3) | ((k >> 1) & 4)) << 1; }
__device__ inline int swz_n(int r) { return (-(r >> 2)) & 3; }

// global address of the 16-byte chunk (stored row r, stored col c) of an operand, or the zero page
__device__ inline const bf16_t* chunk_src(const GemmOperand& o, const bf16_t* base, int r, int c, const bf16_t* zero) {
    if (r >= o.R || c >= o.C) return zero;
    if (o.g_cin > 0) {
        int j = c / o.g_cin, ci = c - j * o.g_cin;
        int b = r / o.g_T, t = r - b * o.g_T;
        int ts = t + j - 2;
        int lim = o.g_len ? min(o.g_len[b], o.g_T) : o.g_T;
        if (ts < 0 || ts >= lim) return zero;
        return base + (long)(b * o.g_T + ts) * o.ld + ci;
    }
    return base + (long)r * o.ld + c;
}

template <bool TA, bool TB, bool GATHER>
__global__ __launch_bounds__(256, 2) void gemm_glds_kernel(GemmArgs g, const bf16_t* zero, float* splitk_ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);      // [buf][A tile | B tile]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int wrow = (wave >> 1) * 64, wcol = (wave & 1) * 64;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int z = blockIdx.z / g.splitk, ksplit = blockIdx.z - z * g.splitk;
    const int zo = z / g.batch_inner, zi = z - zo * g.batch_inner;
    const bf16_t* Ab = reinterpret_cast<const bf16_t*>(g.A.p) + zo * g.A.bs_o + zi * g.A.bs_i;
    const bf16_t* Bb = reinterpret_cast<const bf16_t*>(g.B.p) + zo * g.B.bs_o + zi * g.B.bs_i;

    // DMA issue: wave w owns instructions q = w*2 + i (i = 0..1) of each operand tile; instruction q fills LDS bytes
    // [q*1024, q*1024 + 1024) of the tile.  For plain (non-gather) operands the source address of a lane is affine
    // in the K-tile index, so it is computed once: pointer at tile 0, a per-tile step and a validity bound.
    const bf16_t* pa[2]; const bf16_t* pb[2];
    int ka[2], kb_[2];            // reduction index (relative to the tile start) this lane's chunk covers; -1 = never valid
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = wave * 2 + i;
        if (TA) { int kr = q * 4 + (lane >> 4), pc = lane & 15; int c = m0 + ((pc ^ swz_t(kr)) << 3);
                  ka[i] = c < g.A.C ? kr : -1; pa[i] = Ab + (long)kr * g.A.ld + c; }
        else    { int r = q * 16 + (lane >> 2), pc = lane & 3; int c = (pc ^ swz_n(r)) << 3;
                  ka[i] = (m0 + r) < g.A.R ? c : -1; pa[i] = Ab + (long)(m0 + r) * g.A.ld + c; }
        if (TB) { int kr = q * 4 + (lane >> 4), pc = lane & 15; int c = n0 + ((pc ^ swz_t(kr)) << 3);
                  kb_[i] = c < g.B.C ? kr : -1; pb[i] = Bb + (long)kr * g.B.ld + c; }
        else    { int r = q * 16 + (lane >> 2), pc = lane & 3; int c = (pc ^ swz_n(r)) << 3;
                  kb_[i] = (n0 + r) < g.B.R ? c : -1; pb[i] = Bb + (long)(n0 + r) * g.B.ld + c; }
    }
    const long stepA = TA ? (long)BK * g.A.ld : BK, stepB = TB ? (long)BK * g.B.ld : BK;
    const int limA = TA ? g.A.R : g.A.C, limB = TB ? g.B.R : g.B.C;      // bound of the reduction index

    const int nk_all = (g.K + BK - 1) / BK;
    const int per = (nk_all + g.splitk - 1) / g.splitk;
    const int kt0 = ksplit * per;
    const int kt_end = min(nk_all, kt0 + per);
    const int nk = kt_end - kt0;
    if (nk <= 0) return;

    // Always issues exactly 4 DMA instructions per wave (tiles past the end fetch the zero page into a slot nobody
    // reads again), so the in-flight count is a compile-time constant and the waits below never drain the queue.
    auto issue = [&](int kt, int buf) {
        const int kb = kt < kt_end ? kt * BK : (1 << 28);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = wave * 2 + i;
            const bf16_t *sa, *sb;
            if (GATHER) {
                if (TA) { int kr = q * 4 + (lane >> 4), pc = lane & 15; int c = pc ^ swz_t(kr); sa = chunk_src(g.A, Ab, kb + kr, m0 + c * 8, zero); }
                else    { int r = q * 16 + (lane >> 2), pc = lane & 3; int c = pc ^ swz_n(r);   sa = chunk_src(g.A, Ab, m0 + r, kb + c * 8, zero); }
                if (TB) { int kr = q * 4 + (lane >> 4), pc = lane & 15; int c = pc ^ swz_t(kr); sb = chunk_src(g.B, Bb, kb + kr, n0 + c * 8, zero); }
                else    { int r = q * 16 + (lane >> 2), pc = lane & 3; int c = pc ^ swz_n(r);   sb = chunk_src(g.B, Bb, n0 + r, kb + c * 8, zero); }
            } else {
                sa = (ka[i] >= 0 && kb + ka[i] < limA) ? pa[i] + kt * stepA : zero;
                sb = (kb_[i] >= 0 && kb + kb_[i] < limB) ? pb[i] + kt * stepB : zero;
            }
            __builtin_amdgcn_global_load_lds((gptr_t)sa, (lptr_t)(smem + buf * 2 * TILE + q * 512), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)sb, (lptr_t)(smem + buf * 2 * TILE + TILE + q * 512), 16, 0, 0);
        }
    };
    // Fragment reads are issued as inline-asm DS instructions so that the compiler cannot sink them next to their
    // consumers (which exposes the LDS latency) nor wait for them early: all reads of tile kt+1 are issued before the
    // MFMAs of tile kt and waited for (one lgkmcnt(0)) after them.  Per-lane byte offsets inside a tile:
    unsigned offA[4], offB[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (TA) { const int k = lg * 8 + (li >> 2), col = wrow + t * 16 + (li & 3) * 4, sw = ((li >> 2) | ((lg & 1) << 2)) << 1;
                  offA[t] = 2u * (k * 128 + (((col >> 3) ^ sw) << 3) + (col & 7)); }
        else    { const int r = wrow + t * 16 + li; offA[t] = 2u * (r * 32 + ((lg ^ swz_n(r)) << 3)); }
        if (TB) { const int k = lg * 8 + (li >> 2), col = wcol + t * 16 + (li & 3) * 4, sw = ((li >> 2) | ((lg & 1) << 2)) << 1;
                  offB[t] = 2u * (TILE + k * 128 + (((col >> 3) ^ sw) << 3) + (col & 7)); }
        else    { const int r = wcol + t * 16 + li; offB[t] = 2u * (TILE + r * 32 + ((lg ^ swz_n(r)) << 3)); }
    }
    const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)smem;
    auto frag_issue = [&](bf16x8_t& dst, bool trans, unsigned addr) {
        if (!trans) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
        } else {
            bf16x4_t lo, hi;
            asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:1024"        // rows k and k + 4 (4 x 256 B)
                         : "=&v"(lo), "=&v"(hi) : "v"(addr) : "memory");
            dst[0] = lo[0]; dst[1] = lo[1]; dst[2] = lo[2]; dst[3] = lo[3];
            dst[4] = hi[0]; dst[5] = hi[1]; dst[6] = hi[2]; dst[7] = hi[3];
        }
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // Software pipeline: the DMA ring keeps NSTAGE-1 tiles in flight; the fragments of tile kt+1 are read from LDS while
    // the 16 MFMAs of tile kt execute (register double buffer), so neither HBM nor LDS latency is exposed.
#define B2S_MMA16(CA, CB)                                                                                         \
    _Pragma("unroll") for (int a = 0; a < 4; ++a) _Pragma("unroll") for (int b = 0; b < 4; ++b)                   \
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(CA[a], CB[b], acc[a][b], 0, 0, 0);
#define B2S_STEP(KT, CA, CB, NA, NB)                                                                              \
    {                                                                                                              \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NSTAGE - 3)) : "memory");   /* tile KT+1 landed (own DMAs) */    \
        __builtin_amdgcn_s_barrier();             /* ... for every wave; tile KT-1 is fully consumed */             \
        issue(kt0 + (KT) + NSTAGE - 1, ((KT) + NSTAGE - 1) % NSTAGE);                                              \
        const unsigned sb_ = lds_base + (unsigned)((((KT) + 1) % NSTAGE) * 2 * TILE * 2);                         \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                            \
            frag_issue(NA[t], TA, sb_ + offA[t]);                                                                  \
            frag_issue(NB[t], TB, sb_ + offB[t]);                                                                  \
        }                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        B2S_MMA16(CA, CB)                                                                                          \
        __builtin_amdgcn_sched_barrier(0);        /* keep the MFMAs ABOVE the wait (they are register-only) */      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* tile KT+1 fragments are in registers */              \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
    }
#pragma unroll
    for (int p = 0; p < NSTAGE - 1; ++p) issue(kt0 + p, p);
    bf16x8_t fa0[4], fb0[4], fa1[4], fb1[4];
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NSTAGE - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int t = 0; t < 4; ++t) { frag_issue(fa0[t], TA, lds_base + offA[t]); frag_issue(fb0[t], TB, lds_base + offB[t]); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    int kt = 0;
    for (; kt + 2 < nk; kt += 2) {
        B2S_STEP(kt, fa0, fb0, fa1, fb1)
        B2S_STEP(kt + 1, fa1, fb1, fa0, fb0)
    }
    if (kt + 1 < nk) {          // two tiles left
        B2S_STEP(kt, fa0, fb0, fa1, fb1)
        B2S_MMA16(fa1, fb1)
    } else {                    // one tile left
        B2S_MMA16(fa0, fb0)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the trailing zero-page DMAs before the LDS goes away
#undef B2S_STEP
#undef B2S_MMA16

    // ---------------- epilogue (gemm_epi.h): accumulators -> per-wave LDS staging -> vectorised, fused stores
    gemm_wave_epilogue<4>(g, acc, reinterpret_cast<float*>(smem_raw) + wave * (64 * 64), m0 + wrow, n0 + wcol, lane, z, zo, zi, ksplit, splitk_ws);
}

bf16_t* g_zero_page = nullptr;         // immutable after ensure_globals(): see b2s_gemm_zero_page()

// dst[m*ldc + col(n)] += alpha-scaled sum over splits of ws[s][m][n]
__global__ void splitk_reduce_kernel(const float* ws, float* dst, int M, int N, int ldc, int splitk, int conv_dw_cin) {
    const long total = (long)M * N;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < total; i += (long)gridDim.x * blockDim.x * 4) {
        float4 acc = *reinterpret_cast<const float4*>(ws + i);
        for (int s = 1; s < splitk; ++s) {
            const float4 v = *reinterpret_cast<const float4*>(ws + (long)s * total + i);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        const int m = (int)(i / N), n = (int)(i - (long)m * N);
        const float a4[4] = {acc.x, acc.y, acc.z, acc.w};
        if (conv_dw_cin > 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { int nn = n + j, jj = nn / conv_dw_cin; dst[(long)m * ldc + (nn - jj * conv_dw_cin) * 5 + jj] += a4[j]; }
        } else {
            float4* d = reinterpret_cast<float4*>(dst + (long)m * ldc + n);
            float4 c = *d;
            c.x += acc.x; c.y += acc.y; c.z += acc.z; c.w += acc.w;
            *d = c;
        }
    }
}

// conv weight gradient: ws[s][m][j*cin + ci] -> dst[m*ldc + ci*5 + j] (the [Cout][Cin][5] master layout).  One thread owns the five taps of
// one (m, ci): its slab reads are coalesced over ci and its read-modify-write of dst is 20 contiguous bytes next to its neighbours'
// (the element-wise form above scatters 4-byte accesses at a 20-byte stride: 31 -> 36 us per 512 x 2560 x 6-slab reduction).
__global__ __launch_bounds__(256) void splitk_reduce_conv_kernel(const float* __restrict__ ws, float* __restrict__ dst, int M, int N, int ldc, int splitk, int cin) {
    const long total = (long)M * N, pairs = (long)M * cin;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / cin), ci = (int)(i - (long)m * cin);
        float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        const float* src = ws + (long)m * N + ci;
        for (int s = 0; s < splitk; ++s) {
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[j] += src[(long)s * total + (long)j * cin];
        }
        float* d = dst + (long)m * ldc + (long)ci * 5;
#pragma unroll
        for (int j = 0; j < 5; ++j) d[j] += acc[j];
    }
}

template <bool TA, bool TB, bool GATHER>
int launch_t(const GemmArgs& g_in, hipStream_t stream) {
    GemmArgs g = g_in;
    if (g.splitk > 1) {                     // every split must own at least one K tile (empty splits would leave slabs unwritten)
        const int nk_all = cdiv(g.K, BK), per = cdiv(nk_all, g.splitk);
        g.splitk = cdiv(nk_all, per);
    }
    constexpr size_t smem = NSTAGE * 2 * (size_t)TILE * sizeof(bf16_t);     // 16 KB per stage
    static std::once_flag attr_once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(attr_once, [] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_glds_kernel<TA, TB, GATHER>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    });
    B2S_HIP(attr_err);
    dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM), g.batch * g.splitk);
    float* ws = nullptr;
    if (g.splitk > 1 && g.batch == 1 && g.c_fp32 && g.epi.accumulate && (g.N & 3) == 0 && (g.ldc & 3) == 0 &&
        g.ws && (size_t)g.splitk * g.M * g.N <= g.ws_floats)
        ws = g.ws;
    hipLaunchKernelGGL((gemm_glds_kernel<TA, TB, GATHER>), grid, dim3(256), smem, stream, g, (const bf16_t*)g_zero_page, ws);
    B2S_LAUNCH_CHECK();
    if (ws) B2S_TRY(b2s_splitk_reduce_launch(ws, (float*)g.C, g.M, g.N, g.ldc, g.splitk, g.epi.conv_dw_cin, stream));
    return 0;
}

}  // namespace

int b2s_splitk_reduce_launch(const float* ws, float* dst, int M, int N, int ldc, int splitk, int conv_dw_cin, hipStream_t stream) {
    constexpr bool conv_v1 = false;        // A/B switch
    if (conv_dw_cin > 0 && N == 5 * conv_dw_cin && !conv_v1) {
        const long pairs = (long)M * conv_dw_cin;
        hipLaunchKernelGGL(splitk_reduce_conv_kernel, dim3((int)std::min<long>((pairs + 255) / 256, 2048)), dim3(256), 0, stream, ws, dst, M, N, ldc, splitk,
                           conv_dw_cin);
        B2S_LAUNCH_CHECK();
        return 0;
    }
    const long total4 = (long)M * N / 4;
    int blocks = (int)std::min<long>((total4 + 255) / 256, 2048);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, ws, dst, M, N, ldc, splitk, conv_dw_cin);
    B2S_LAUNCH_CHECK();
    return 0;
}

static int ensure_globals() {
    static std::once_flag once;
    static hipError_t err = hipSuccess;
    std::call_once(once, [] {
        err = hipMalloc(&g_zero_page, 256);
        if (err == hipSuccess) err = hipMemset(g_zero_page, 0, 256);
    });
    B2S_HIP(err);
    return 0;
}
const bf16_t* b2s_gemm_zero_page() { return ensure_globals() ? nullptr : g_zero_page; }

int b2s_gemm_glds_launch(const GemmArgs& g, bool ta, bool tb, hipStream_t stream) {
    B2S_TRY(ensure_globals());
    // tile shape: 256-row tiles with 64-deep K steps (gemm_glds256.hip) for every non-batched problem taller than one
    // 128-row tile -- measured faster in the training step down to the M = 1596 encoder shapes (half the barriers per
    // FLOP); the 128x128 kernel keeps the batched (per-head) and short problems.  B2S_GEMM256_MIN_M overrides.
    static const long min_m = getenv("B2S_GEMM256_MIN_M") ? atol(getenv("B2S_GEMM256_MIN_M")) : 129;
    if (g.batch == 1 && g.M >= min_m)
        return b2s_gemm_glds256_launch(g, ta, tb, g_zero_page, stream);
    const bool gather = g.A.g_cin > 0 || g.B.g_cin > 0;
    if (gather) {       // conv1d forms: forward / backward-data (NT, gather on A) and weight gradient (TN, gather on B)
        if (!ta && !tb) return launch_t<false, false, true>(g, stream);
        if (ta && tb) return launch_t<true, true, true>(g, stream);
        return b2s_fail(__FILE__, __LINE__, "conv gather is supported for the NT and TN forms only");
    }
    if (!ta && !tb) return launch_t<false, false, false>(g, stream);
    if (!ta && tb) return launch_t<false, true, false>(g, stream);
    if (ta && !tb) return launch_t<true, false, false>(g, stream);
    return launch_t<true, true, false>(g, stream);
}
