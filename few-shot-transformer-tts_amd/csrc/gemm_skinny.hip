// Weight-streaming GEMM for the autoregressive decode step: out[M <= 64, N] = X[M, K] * W[N, K]^T with the usual
// epilogues.  The step is HBM-bound on W (every weight is read once per frame), so the launch is shaped for
// bandwidth, not MFMA occupancy: one workgroup per 16 output columns (N/16 workgroups stream disjoint 16-row
// slabs of W), its 8 waves split K eight ways and meet in LDS, each issuing the loads of 4 K steps at a time; X (<= 64 rows) comes from L2.
#include <cstdlib>
#include "gemm.h"

namespace {

template <typename T> struct SK;
template <> struct SK<float>  { static constexpr int KS = 4;  typedef float frag; };
template <> struct SK<bf16_t> { static constexpr int KS = 32; typedef bf16x8_t frag; };

__device__ inline float ldfrag(const float* p, int lg) { return p[lg]; }
__device__ inline bf16x8_t ldfrag(const bf16_t* p, int lg) { return *reinterpret_cast<const bf16x8_t*>(p + lg * 8); }
__device__ inline f32x4_t mma(float a, float b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ inline f32x4_t mma(bf16x8_t a, bf16x8_t b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

constexpr int NW = 8;          // waves per workgroup: K is split NW ways
constexpr int UN = 4;          // K steps whose loads are issued together (memory-level parallelism per wave)

template <typename T>
__global__ __launch_bounds__(NW * 64) void skinny_kernel(GemmArgs g) {
    constexpr int KS = SK<T>::KS;
    __shared__ float red[NW][64][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const T* X = reinterpret_cast<const T*>(g.A.p);
    const T* W = reinterpret_cast<const T*>(g.B.p);
    // gridDim.y > 1: K is also split over workgroups (the 768 x 3072 ffn-out GEMM has only 48 column slabs and a 12-step
    // K walk per wave); partial results are added atomically into the fp32 output, which already holds the residual
    const int nsteps_all = (g.K + KS - 1) / KS, per_wg = (nsteps_all + gridDim.y - 1) / gridDim.y;
    const int w0 = blockIdx.y * per_wg, nsteps = min(nsteps_all, w0 + per_wg);
    const int per = (nsteps - w0 + NW - 1) / NW;
    const int s0 = w0 + wave * per, s1 = min(nsteps, s0 + per);
    const T* wrow = W + (long)min(n0 + li, g.N - 1) * g.B.ld;
    const T* xrow[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) xrow[mt] = X + (long)min(mt * 16 + li, g.M - 1) * g.A.ld;
    f32x4_t acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int mtiles = (g.M + 15) / 16;
    // epilogue operands of this thread's (at most 2) output elements, fetched ahead of the K walk instead of after it
    float rpre[2] = {0.f, 0.f}, bpre[2] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int i = tid + t * NW * 64, m = i >> 4, n = n0 + (i & 15);
        if (i < 64 * 16 && m < g.M && n < g.N) {
            if (g.epi.residual && gridDim.y == 1) rpre[t] = g.epi.residual[(long)m * g.epi.ldr + n];
            if (g.epi.bias && blockIdx.y == 0) bpre[t] = g.epi.bias[n];
        }
    }
    for (int s = s0; s < s1; s += UN) {
        typename SK<T>::frag wb[UN], xa[UN][4];
#pragma unroll
        for (int u = 0; u < UN; ++u) {                     // issue every load of the group before the first MFMA
            const int k = min(s + u, s1 - 1) * KS;         // (tail steps re-read the last valid step; their MFMAs are skipped)
            wb[u] = ldfrag(wrow + k, lg);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                if (mt < mtiles) xa[u][mt] = ldfrag(xrow[mt] + k, lg);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
            if (s + u < s1) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    if (mt < mtiles) acc[mt] = mma(xa[u][mt], wb[u], acc[mt]);
            }
    }
    // D layout: col = n (li), row = m (lg*4 + r) within the m tile
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][mt * 16 + lg * 4 + r][li] = acc[mt][r];
    __syncthreads();
    const GemmEpilogue& e = g.epi;
    DropCfg dcfg = e.drop;
    if (e.drop.thresh && e.drop_salt) dcfg.key ^= b2s_hash32((uint32_t)(*e.drop_salt) * 2246822519u + 3266489917u);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int i = tid + t * NW * 64;
        const int m = i >> 4, nl = i & 15, n = n0 + nl;
        if (i >= 64 * 16 || m >= g.M || n >= g.N) continue;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][m][nl];
        v *= e.alpha;
        v += bpre[t];
        if (e.relu) v = fmaxf(v, 0.f);
        if (e.drop.thresh) v = b2s_keep(dcfg, (uint32_t)((long)m * g.N + n)) ? v * dcfg.scale : 0.f;
        if (gridDim.y > 1) { atomicAdd(reinterpret_cast<float*>(g.C) + (long)m * g.ldc + n, v); continue; }    // (launcher: C == residual, fp32, linear epilogue)
        v += rpre[t];
        if (e.row_len) { int bb = m / e.rows_per_batch, t = m - bb * e.rows_per_batch; if (t >= e.row_len[bb]) v = 0.f; }
        const long off = (long)m * g.ldc + n;
        if (g.c_fp32) { float* C = reinterpret_cast<float*>(g.C); if (e.accumulate) C[off] += v; else C[off] = v; }
        else TT<T>::st(reinterpret_cast<T*>(g.C) + off, v);
        if (e.kv_k && n >= e.kv_D) {          // this frame's k / v columns also go to the caches at position *kv_t
            const int which = n >= 2 * e.kv_D, c = n - (1 + which) * e.kv_D;
            T* cache = reinterpret_cast<T*>(which ? e.kv_v : e.kv_k);
            const int hh = c / e.kv_dh, d = c - hh * e.kv_dh;
            TT<T>::st(cache + (((long)m * (e.kv_D / e.kv_dh) + hh) * e.kv_maxT + *e.kv_t) * e.kv_dh + d, v);
        }
    }
}

}  // namespace

// NT form, M <= 64, no batching / gather / split-K / relu_aux.  Returns -1 if the problem does not fit this kernel.
int b2s_gemm_skinny_launch(const GemmArgs& g, int dtype, hipStream_t stream) {
    const int ks = dtype ? 32 : 4;
    if (g.M > 64 || g.batch != 1 || g.splitk != 1 || g.A.g_cin || g.B.g_cin || g.epi.relu_aux || g.epi.conv_dw_cin || (g.K % ks) != 0)
        return -1;
    if (g.epi.kv_k && (!g.epi.kv_v || !g.epi.kv_t || g.N != 3 * g.epi.kv_D || g.epi.kv_dh <= 0 || g.epi.kv_D % g.epi.kv_dh)) return -1;
    // K split over workgroups for the few-column, deep-K problems of the bf16 decode step (fp32 keeps one deterministic
    // summation order): the epilogue must be linear in the accumulator and the output must already hold the residual
    int ksplit = 1;
    constexpr bool no_split = false;
    if (dtype && !no_split && g.K >= 2048 && g.N <= 1024 && g.c_fp32 && g.epi.residual == g.C && g.epi.ldr == g.ldc && !g.epi.relu &&
        !g.epi.accumulate && !g.epi.row_len && !g.epi.kv_k)
        ksplit = 4;
    dim3 grid(cdiv(g.N, 16), ksplit);
    if (dtype) hipLaunchKernelGGL((skinny_kernel<bf16_t>), grid, dim3(NW * 64), 0, stream, g);
    else hipLaunchKernelGGL((skinny_kernel<float>), grid, dim3(NW * 64), 0, stream, g);
    B2S_LAUNCH_CHECK();
    return 0;
}
