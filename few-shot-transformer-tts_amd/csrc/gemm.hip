// MFMA GEMM for gfx950 (MI355X): C = epilogue(op(A) * op(B)), batched, with an optional conv1d
// (k=5, pad=2) row-gather on either operand.  One kernel template serves every dense contraction
// of the Transformer-TTS path (reference: every nn.Linear / torch.matmul / nn.Conv1d call in
// transformer/attention.py:43-47,83,91, transformer/modules.py:11-13, transformer/tacotron.py:50-52,
// 78,104-105, and their autograd backward forms).
//
//   * 128x128 block tile, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16 tiles.
//   * bf16: v_mfma_f32_16x16x32_bf16, BK = 64, fp32 accumulate.   fp32: v_mfma_f32_16x16x4_f32
//     (exact fp32 fma chain, used by the parity mode), BK = 16.
//   * operands are staged HBM -> registers (16-byte loads) -> LDS, double-buffered, one barrier per
//     K tile; the next tile's global loads are issued before the MFMAs of the current tile.
//   * a K-contiguous operand is kept [rows][BK+pad] in LDS and read with ds_read_b128 (bf16) /
//     ds_read_b32 (fp32); an operand whose reduction index is the slow dimension (the "NN"/"TN"
//     backward forms) is kept [BK][rows+pad] and read with ds_read_b64_tr_b16 (bf16), so no
//     transposed copy of any activation or weight is ever written to HBM.
#include <mutex>
#include "gemm.h"
#include "gemm_epi.h"

namespace {

constexpr int BM = 128, BN = 128;

template <typename T> struct Cfg;
template <> struct Cfg<float>  { static constexpr int BK = 16, KSTEP = 4,  VE = 4; };
template <> struct Cfg<bf16_t> { static constexpr int BK = 64, KSTEP = 32, VE = 8; };

template <typename T> struct FragT;
template <> struct FragT<float>  { typedef float type; };
template <> struct FragT<bf16_t> { typedef bf16x8_t type; };

typedef __attribute__((address_space(3))) bf16x4_t* lds_b64_ptr;

// ---- fragment loads.  lane: i = lane & 15 (row/col inside the 16-tile), g = lane >> 4 (k group)
// K-contiguous tile [rows][LD]: element (row0 + i, k0 + g*KPL ..)
__device__ inline float frag_n(const float* tile, int LD, int row0, int k0, int i, int g) {
    return tile[(row0 + i) * LD + k0 + g];
}
__device__ inline bf16x8_t frag_n(const bf16_t* tile, int LD, int row0, int k0, int i, int g) {
    return *reinterpret_cast<const bf16x8_t*>(tile + (row0 + i) * LD + k0 + g * 8);
}
// reduction-major tile [BK][LD] (LD = rows + pad): same logical fragment, gathered by transpose reads
__device__ inline float frag_t(const float* tile, int LD, int row0, int k0, int i, int g) {
    return tile[(k0 + g) * LD + row0 + i];
}
__device__ inline bf16x8_t frag_t(const bf16_t* tile, int LD, int row0, int k0, int i, int g) {
    // ds_read_b64_tr_b16: within a 16-lane group, lane i receives element (i & 3) of the 8-byte
    // rows supplied by lanes 4j + (i >> 2), j = 0..3.  Supplying row (k0 + g*8 + (i >> 2)), columns
    // row0 + (i & 3)*4 .. +3 therefore returns [k0 + g*8 + j][row0 + i], j = 0..3.
    const bf16_t* p0 = tile + (k0 + g * 8 + (i >> 2)) * LD + row0 + (i & 3) * 4;
    bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64_ptr)(p0));
    bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64_ptr)(p0 + 4 * LD));
    bf16x8_t r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
__device__ inline f32x4_t mma(float a, float b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ inline f32x4_t mma(bf16x8_t a, bf16x8_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ---- one 16-byte chunk of an operand: stored row r, stored cols c .. c+VE-1 (zero if out of range)
template <typename T>
__device__ inline uint4 load_chunk(const GemmOperand& o, const T* base, int r, int c) {
    uint4 z = make_uint4(0, 0, 0, 0);
    if (o.g_cin > 0) {
        // conv gather: r = token (b*T + t), c = j*cin + ci
        if (r >= o.R || c >= o.C) return z;
        int j = c / o.g_cin, ci = c - j * o.g_cin;
        int b = r / o.g_T, t = r - b * o.g_T;
        int ts = t + j - 2;
        int lim = o.g_len ? min(o.g_len[b], o.g_T) : o.g_T;
        if (ts < 0 || ts >= lim) return z;
        return *reinterpret_cast<const uint4*>(base + (long)(b * o.g_T + ts) * o.ld + ci);
    }
    if (r >= o.R || c >= o.C) return z;
    return *reinterpret_cast<const uint4*>(base + (long)r * o.ld + c);
}

template <typename T, bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    constexpr int BK = Cfg<T>::BK, KSTEP = Cfg<T>::KSTEP, VE = Cfg<T>::VE;
    constexpr int LDN = BK + VE;          // K-contiguous tile row stride (elements)
    constexpr int LDT = 128 + VE;         // reduction-major tile row stride
    constexpr int TILE_A = TA ? BK * LDT : BM * LDN;
    constexpr int TILE_B = TB ? BK * LDT : BN * LDN;
    constexpr int NV = 128 * BK / VE / 256;     // 16-byte vectors per thread per operand tile
    constexpr int VPR_N = BK / VE, VPR_T = 128 / VE;
    typedef typename FragT<T>::type frag_t_;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // NOTE: LDS addresses are formed as (shared base + integer offset) everywhere: keeping tile pointers in an
    // array makes the compiler lose the LDS address space and emit FLAT loads/stores for the fragments.
    T* smem = reinterpret_cast<T*>(smem_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int wrow = (wave >> 1) * 64, wcol = (wave & 1) * 64;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int z = blockIdx.z / g.splitk, ksplit = blockIdx.z - z * g.splitk;
    const int zo = z / g.batch_inner, zi = z - zo * g.batch_inner;
    const T* Ab = reinterpret_cast<const T*>(g.A.p) + zo * g.A.bs_o + zi * g.A.bs_i;
    const T* Bb = reinterpret_cast<const T*>(g.B.p) + zo * g.B.bs_o + zi * g.B.bs_i;

    uint4 ra[NV], rb[NV];
    auto gload = [&](int kt) {
        const int kb = kt * BK;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int v = tid + i * 256;
            if (TA) { int kr = v / VPR_T, c = (v % VPR_T) * VE; ra[i] = load_chunk<T>(g.A, Ab, kb + kr, m0 + c); }
            else    { int r = v / VPR_N, c = (v % VPR_N) * VE;  ra[i] = load_chunk<T>(g.A, Ab, m0 + r, kb + c); }
            if (TB) { int kr = v / VPR_T, c = (v % VPR_T) * VE; rb[i] = load_chunk<T>(g.B, Bb, kb + kr, n0 + c); }
            else    { int r = v / VPR_N, c = (v % VPR_N) * VE;  rb[i] = load_chunk<T>(g.B, Bb, n0 + r, kb + c); }
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int v = tid + i * 256;
            const int oa = buf * TILE_A, ob = 2 * TILE_A + buf * TILE_B;
            if (TA) { int kr = v / VPR_T, c = (v % VPR_T) * VE; *reinterpret_cast<uint4*>(smem + oa + kr * LDT + c) = ra[i]; }
            else    { int r = v / VPR_N, c = (v % VPR_N) * VE;  *reinterpret_cast<uint4*>(smem + oa + r * LDN + c) = ra[i]; }
            if (TB) { int kr = v / VPR_T, c = (v % VPR_T) * VE; *reinterpret_cast<uint4*>(smem + ob + kr * LDT + c) = rb[i]; }
            else    { int r = v / VPR_N, c = (v % VPR_N) * VE;  *reinterpret_cast<uint4*>(smem + ob + r * LDN + c) = rb[i]; }
        }
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk_all = (g.K + BK - 1) / BK;
    const int per = (nk_all + g.splitk - 1) / g.splitk;
    const int kt0 = ksplit * per;
    const int nk = min(nk_all, kt0 + per) - kt0;
    if (nk <= 0) return;
    gload(kt0);
    sstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload(kt0 + kt + 1);
        const T* tA = smem + cur * TILE_A;
        const T* tB = smem + 2 * TILE_A + cur * TILE_B;
#pragma unroll
        for (int ks = 0; ks < BK / KSTEP; ++ks) {
            frag_t_ fa[4], fb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fa[t] = TA ? frag_t(tA, LDT, wrow + t * 16, ks * KSTEP, li, lg)
                           : frag_n(tA, LDN, wrow + t * 16, ks * KSTEP, li, lg);
                fb[t] = TB ? frag_t(tB, LDT, wcol + t * 16, ks * KSTEP, li, lg)
                           : frag_n(tB, LDN, wcol + t * 16, ks * KSTEP, li, lg);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = mma(fa[a], fb[b], acc[a][b]);
        }
        if (kt + 1 < nk) sstore(cur ^ 1);
        __syncthreads();
    }

    // ---------------- epilogue.  C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4)*4 + r
    const GemmEpilogue& e = g.epi;
    if (e.colstat && g.splitk == 1) epi_colstat<4>(acc, e.colstat, e.alpha, g.M, g.N, m0 + wrow, n0 + wcol, lane);
    const long cbase = zo * g.cs_o + zi * g.cs_i;
    float* Cf = reinterpret_cast<float*>(g.C);
    T* Ct = reinterpret_cast<T*>(g.C);
    const T* aux = reinterpret_cast<const T*>(e.relu_aux);
    int ncol[4], nst[4];
    float bv[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int n = n0 + wcol + b * 16 + li;
        ncol[b] = n < g.N ? n : -1;
        bv[b] = (e.bias && n < g.N) ? e.bias[n] : 0.f;
        nst[b] = n;
        if (e.conv_dw_cin > 0) { int j = n / e.conv_dw_cin; nst[b] = (n - j * e.conv_dw_cin) * 5 + j; }
    }
    const bool plain = !e.relu && !aux && !e.drop.thresh && !e.residual && !e.row_len;
    DropCfg dcfg = e.drop;
    if (e.drop.thresh && e.drop_salt) dcfg.key ^= b2s_hash32((uint32_t)(*e.drop_salt) * 2246822519u + 3266489917u);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wrow + a * 16 + lg * 4 + r;
            if (m >= g.M) continue;
            const long rowoff = cbase + (long)m * g.ldc;
            if (plain) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (ncol[b] < 0) continue;
                    const float v = acc[a][b][r] * e.alpha + bv[b];
                    const long off = rowoff + nst[b];
                    if (g.c_fp32) { if (g.splitk > 1) atomicAdd(Cf + off, v); else if (e.accumulate) Cf[off] += v; else Cf[off] = v; }
                    else TT<T>::st(Ct + off, v);
                }
                continue;
            }
            bool rowzero = false;
            if (e.row_len) {
                int bb = m / e.rows_per_batch, t = m - bb * e.rows_per_batch;
                rowzero = t >= e.row_len[bb];
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int n = ncol[b];
                if (n < 0) continue;
                float v = acc[a][b][r] * e.alpha + bv[b];
                if (e.relu) v = fmaxf(v, 0.f);
                if (aux) v = TT<T>::ld(aux + (long)m * e.ld_aux + n) > 0.f ? v * e.aux_scale : 0.f;
                if (e.drop.thresh) {
                    uint32_t idx = (uint32_t)(((long)z * g.M + m) * g.N + n);
                    v = b2s_keep(dcfg, idx) ? v * dcfg.scale : 0.f;
                }
                if (e.residual) v += e.residual[(long)m * e.ldr + n];
                if (rowzero) v = 0.f;
                const long off = rowoff + nst[b];
                if (g.c_fp32) { if (e.accumulate) Cf[off] += v; else Cf[off] = v; }
                else TT<T>::st(Ct + off, v);
            }
        }
    }
}

template <typename T, bool TA, bool TB>
int launch_t(const GemmArgs& g, hipStream_t stream) {
    constexpr int BK = Cfg<T>::BK, VE = Cfg<T>::VE;
    constexpr int LDN = BK + VE, LDT = 128 + VE;
    constexpr int TILE_A = TA ? BK * LDT : BM * LDN;
    constexpr int TILE_B = TB ? BK * LDT : BN * LDN;
    constexpr size_t smem = 2 * (size_t)(TILE_A + TILE_B) * sizeof(T);
    static std::once_flag attr_once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(attr_once, [] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<T, TA, TB>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    });
    B2S_HIP(attr_err);
    dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM), g.batch * g.splitk);
    hipLaunchKernelGGL((gemm_kernel<T, TA, TB>), grid, dim3(256), smem, stream, g);
    B2S_LAUNCH_CHECK();
    return 0;
}

template <typename T>
int launch_d(const GemmArgs& g, bool ta, bool tb, hipStream_t s) {
    if (!ta && !tb) return launch_t<T, false, false>(g, s);
    if (!ta && tb) return launch_t<T, false, true>(g, s);
    if (ta && !tb) return launch_t<T, true, false>(g, s);
    return launch_t<T, true, true>(g, s);
}

}  // namespace

// ---- optional per-launch timing with HIP events on the launch stream (bench.py roofline leg)
#include <atomic>
#include <vector>
#include <cstdlib>
namespace {
struct ProfRec { hipEvent_t a, b; int variant; double flops; int M, N, K, batch, splitk; };
// measurement state (b2s_prof_*): process-wide by design -- one timing log for every model in the process -- and guarded
std::atomic<bool> g_prof_on{false};
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
}  // namespace
extern "C" void b2s_prof_enable(int on) { g_prof_on = on != 0; }
// out[v*3 + {0,1,2}] = total flops, total milliseconds, launches of GEMM variant v = dtype*8 + trans_a*4 + trans_b*2 + conv_gather;
// v = 16: grouped bf16 weight-gradient launches
extern "C" int b2s_prof_collect(double* out, int n_variants) {
    for (int i = 0; i < n_variants * 3; ++i) out[i] = 0.0;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    for (auto& r : g_prof) {
        B2S_HIP(hipEventSynchronize(r.b));
        float ms = 0.f;
        B2S_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        if (r.variant < n_variants) { out[r.variant * 3] += r.flops; out[r.variant * 3 + 1] += ms; out[r.variant * 3 + 2] += 1.0; }
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    g_prof.clear();
    return 0;
}

static int gemm_launch_inner(const GemmArgs& g, int dtype, bool ta, bool tb, hipStream_t stream);
int b2s_gemm_launch(const GemmArgs& g, int dtype, bool ta, bool tb, hipStream_t stream) {
    if (!g_prof_on) return gemm_launch_inner(g, dtype, ta, tb, stream);
    ProfRec r;
    B2S_HIP(hipEventCreate(&r.a)); B2S_HIP(hipEventCreate(&r.b));
    r.variant = dtype * 8 + (ta ? 4 : 0) + (tb ? 2 : 0) + ((g.A.g_cin > 0 || g.B.g_cin > 0) ? 1 : 0);
    r.flops = 2.0 * g.M * g.N * (double)g.K * g.batch;
    r.M = g.M; r.N = g.N; r.K = g.K; r.batch = g.batch; r.splitk = g.splitk;
    B2S_HIP(hipEventRecord(r.a, stream));
    int rc = gemm_launch_inner(g, dtype, ta, tb, stream);
    B2S_HIP(hipEventRecord(r.b, stream));
    { std::lock_guard<std::mutex> lock(g_prof_mu); g_prof.push_back(r); }
    return rc;
}
// grouped bf16 weight-gradient launch (gemm_glds256.hip), with the same optional timing record (variant 16)
int b2s_gemm_grouped_launch(const GemmArgs* probs, int n, hipStream_t stream) {
    if (!g_prof_on) return b2s_gemm_glds256_grouped_launch(probs, n, b2s_gemm_zero_page(), stream);
    ProfRec r;
    B2S_HIP(hipEventCreate(&r.a)); B2S_HIP(hipEventCreate(&r.b));
    r.variant = 16; r.flops = 0.0;
    for (int i = 0; i < n; ++i) r.flops += 2.0 * probs[i].M * probs[i].N * (double)probs[i].K;
    r.M = probs[0].M; r.N = probs[0].N; r.K = probs[0].K; r.batch = n; r.splitk = 1;
    B2S_HIP(hipEventRecord(r.a, stream));
    int rc = b2s_gemm_glds256_grouped_launch(probs, n, b2s_gemm_zero_page(), stream);
    B2S_HIP(hipEventRecord(r.b, stream));
    { std::lock_guard<std::mutex> lock(g_prof_mu); g_prof.push_back(r); }
    return rc;
}
static int gemm_launch_inner(const GemmArgs& g, int dtype, bool ta, bool tb, hipStream_t stream) {
    const int ve = dtype ? 8 : 4;
    B2S_CHECK(g.M > 0 && g.N > 0 && g.K > 0 && g.batch > 0 && g.batch_inner > 0, "gemm: bad shape M=%d N=%d K=%d batch=%d",
              g.M, g.N, g.K, g.batch);
    B2S_CHECK(g.A.p && g.B.p && g.C, "gemm: null pointer");
    B2S_CHECK(g.A.ld % ve == 0 && g.B.ld % ve == 0, "gemm: leading dimensions (%d, %d) must be multiples of %d", g.A.ld,
              g.B.ld, ve);
    B2S_CHECK(((uintptr_t)g.A.p % 16 == 0) && ((uintptr_t)g.B.p % 16 == 0), "gemm: operands must be 16-byte aligned");
    B2S_CHECK(g.A.bs_o % ve == 0 && g.A.bs_i % ve == 0 && g.B.bs_o % ve == 0 && g.B.bs_i % ve == 0,
              "gemm: batch strides must be multiples of %d elements", ve);
    B2S_CHECK(g.batch * g.splitk <= 65535, "gemm: batch too large");
    B2S_CHECK(g.splitk >= 1 && (g.splitk == 1 || (g.c_fp32 && g.epi.accumulate && !g.epi.relu && !g.epi.bias && !g.epi.residual &&
                                                   !g.epi.relu_aux && !g.epi.drop.thresh)),
              "gemm: split-K needs a linear fp32 accumulate epilogue");
    B2S_CHECK(!g.epi.kv_k, "gemm: the cache-append fusion exists in the decode-step kernel only (M <= 64, K %% 32 == 0)");
    constexpr bool use_v1 = false;         // A/B switch: register-staged bf16 main loop
    if (dtype && !use_v1) return b2s_gemm_glds_launch(g, ta, tb, stream);
    return dtype ? launch_d<bf16_t>(g, ta, tb, stream) : launch_d<float>(g, ta, tb, stream);
}
