// Fused multi-head attention for gfx950 (reference: transformer/attention.py:72-92 dot_product_attention and its
// autograd backward).  Flash-style: the [Lq, Lk] logits / weights are never written to HBM.
//
//   forward   one workgroup = 64 query rows of one (batch, head): 4 waves x 16 rows.  K and V tiles of 64 keys are
//             staged in LDS; S^T = K Q^T and O^T += V^T P^T run on MFMA (bf16: 16x16x32, fp32 parity mode: 16x16x4);
//             the swapped products leave every lane with 16 logits of ONE query row per tile, so the online softmax
//             needs only two wavefront shuffles (across the four 16-lane groups) per tile.  Masks come from the
//             lengths / causal structure, dropout from the counter RNG (same element index as the backward).
//   backward  two kernels that recompute P from the saved log-sum-exp: dQ (per 64 query rows, loops over key tiles)
//             and dK/dV (per 64 keys, loops over query tiles).  D = rowsum(dO * O) comes from a tiny prep kernel.
//
// Operand layouts: q [B, Lq, H*dh] with leading dimension ldq (heads interleaved, exactly as the fused QKV / KV
// projection GEMMs write them), k, v likewise; ctx [B, Lq, H*dh].
#include <cstdlib>
#include <mutex>
#include "b2s_common.h"
#include "attention.h"

namespace {

typedef __attribute__((address_space(3))) bf16x4_t* lds_b64_ptr;

template <typename T> struct AT;
template <> struct AT<float> {
    static constexpr int PAD = 4, VE = 4;
    typedef float frag;                      // one fp32 per lane per 16x16x4 MFMA operand
};
template <> struct AT<bf16_t> {
    static constexpr int PAD = 8, VE = 8;
    typedef bf16x8_t frag;                   // 8 bf16 per lane per 16x16x32 MFMA operand
};

__device__ inline f32x4_t mma(float a, float b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ inline f32x4_t mma(bf16x8_t a, bf16x8_t b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// cooperative load of a [64][DH] tile (rows row0.. of a [rows, ld] matrix, zero beyond nrows) into LDS [64][DH+PAD], split
// in two halves so that the global loads of the NEXT tile are in flight while the current tile is being consumed
// (register double buffer; a single fused loop compiled to load -> wait -> store round trips, 3 per tile, and was the
// whole cost of these kernels).
template <typename T, int DH> struct TileRegs { uint4 v[64 * (DH / AT<T>::VE) / 256]; };
template <typename T, int DH>
__device__ inline void tile_fetch(TileRegs<T, DH>& r, const T* src, long ld, int row0, int nrows, int tid) {
    constexpr int VE = AT<T>::VE, VPR = DH / VE, NV = 64 * VPR / 256;
    static_assert(64 * VPR % 256 == 0, "tile must split evenly over 256 threads");
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * 256, rr = v / VPR, c = (v - rr * VPR) * VE;
        r.v[i] = make_uint4(0, 0, 0, 0);
        if (row0 + rr < nrows) r.v[i] = *reinterpret_cast<const uint4*>(src + (long)(row0 + rr) * ld + c);
    }
}
template <typename T, int DH>
__device__ inline void tile_store(T* lds, const TileRegs<T, DH>& r, int tid) {
    constexpr int VE = AT<T>::VE, LD = DH + AT<T>::PAD, VPR = DH / VE, NV = 64 * VPR / 256;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * 256, rr = v / VPR, c = (v - rr * VPR) * VE;
        *reinterpret_cast<uint4*>(lds + rr * LD + c) = r.v[i];
    }
}

// ---- fragment helpers (lane: li = lane & 15, lg = lane >> 4)
// row-operand fragment: element rows (row0 + li), k columns of step ks            (K-contiguous read)
template <int LD> __device__ inline float frag_row(const float* t, int row0, int ks, int li, int lg) { return t[(row0 + li) * LD + ks * 4 + lg]; }
template <int LD> __device__ inline bf16x8_t frag_row(const bf16_t* t, int row0, int ks, int li, int lg) {
    return *reinterpret_cast<const bf16x8_t*>(t + (row0 + li) * LD + ks * 32 + lg * 8);
}
// same fragment taken straight from global memory (the wave's own 16 rows; rows beyond nrows read row `clamp`)
template <typename T> __device__ inline typename AT<T>::frag frag_global(const T* base, long ld, int row, int ks, int lg);
template <> __device__ inline float frag_global<float>(const float* base, long ld, int row, int ks, int lg) { return base[row * ld + ks * 4 + lg]; }
template <> __device__ inline bf16x8_t frag_global<bf16_t>(const bf16_t* base, long ld, int row, int ks, int lg) {
    return *reinterpret_cast<const bf16x8_t*>(base + row * ld + ks * 32 + lg * 8);
}
// transposed fragment for the "second" products (reduction over the tile's 64 rows, output dim = tile columns):
// bf16: rows {r0 + lg*4 + j, r1 + lg*4 + j}, column c0 + li, via two transpose reads; the matching P fragment packs
// p[t0][0..3], p[t1][0..3].
template <int LD> __device__ inline bf16x8_t frag_tr(const bf16_t* t, int r0, int r1, int c0, int li, int lg) {
    const bf16_t* p0 = t + (r0 + lg * 4 + (li >> 2)) * LD + c0 + (li & 3) * 4;
    const bf16_t* p1 = t + (r1 + lg * 4 + (li >> 2)) * LD + c0 + (li & 3) * 4;
    bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64_ptr)p0);
    bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64_ptr)p1);
    bf16x8_t r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
__device__ inline bf16x8_t pack8(const f32x4_t& a, const f32x4_t& b) {
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
    const u32x4_t u = {f2bf2(a[0], a[1]), f2bf2(a[2], a[3]), f2bf2(b[0], b[1]), f2bf2(b[2], b[3])};
    return __builtin_bit_cast(bf16x8_t, u);
}

// acc[dt] (+)= sum over the tile's 64 rows of  tile[row][dt*16 + i] * w[row][j]     (w in the MFMA C layout of a
// [64 rows x 16] block: w[t][r] belongs to row t*16 + lg*4 + r, column li)          -> result col = li, row = dt*16+lg*4+r
template <typename T, int DH, int LD>
__device__ inline void second_product(f32x4_t (&acc)[DH / 16], const T* tile, const f32x4_t (&w)[4], int li, int lg);
template <int DH, int LD>
__device__ inline void second_product_f32(f32x4_t (&acc)[DH / 16], const float* tile, const f32x4_t (&w)[4], int li, int lg) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float b = w[t][r];
            const float* row = tile + (t * 16 + lg * 4 + r) * LD + li;
#pragma unroll
            for (int dt = 0; dt < DH / 16; ++dt) acc[dt] = mma(row[dt * 16], b, acc[dt]);
        }
}
template <int DH, int LD>
__device__ inline void second_product_bf16(f32x4_t (&acc)[DH / 16], const bf16_t* tile, const f32x4_t (&w)[4], int li, int lg) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const bf16x8_t b = pack8(w[2 * kb], w[2 * kb + 1]);
#pragma unroll
        for (int dt = 0; dt < DH / 16; ++dt)
            acc[dt] = mma(frag_tr<LD>(tile, kb * 32, kb * 32 + 16, dt * 16, li, lg), b, acc[dt]);
    }
}
template <typename T, int DH, int LD> struct SP;
template <int DH, int LD> struct SP<float, DH, LD> {
    __device__ static inline void run(f32x4_t (&acc)[DH / 16], const float* tile, const f32x4_t (&w)[4], int li, int lg) { second_product_f32<DH, LD>(acc, tile, w, li, lg); }
};
template <int DH, int LD> struct SP<bf16_t, DH, LD> {
    __device__ static inline void run(f32x4_t (&acc)[DH / 16], const bf16_t* tile, const f32x4_t (&w)[4], int li, int lg) { second_product_bf16<DH, LD>(acc, tile, w, li, lg); }
};

// first product: c[t] = sum_d tile[t*16 + i][d] * own[d][j]  -> c[t][r] = value(row t*16 + lg*4 + r of the tile, own row li)
template <typename T, int DH, int LD>
__device__ inline void first_product(f32x4_t (&c)[4], const T* tile, const typename AT<T>::frag (&own)[DH / (sizeof(T) == 2 ? 32 : 4)], int li, int lg) {
    constexpr int NKS = DH / (sizeof(T) == 2 ? 32 : 4);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        c[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) c[t] = mma(frag_row<LD>(tile, t * 16, ks, li, lg), own[ks], c[t]);
    }
}

__device__ inline float frag_dot(float x, float y) { return x * y; }
__device__ inline float frag_dot(bf16x8_t x, bf16x8_t y) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += bf2f((bf16_t)x[e]) * bf2f((bf16_t)y[e]);
    return s;
}
__device__ inline float group_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ inline float group_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

// XCD-aware (tile, head) assignment: workgroups go to the 8 XCDs round-robin by linear id and every XCD has its own L2, so
// by default the ~10 query tiles that share one head's K / V land on 8 different L2s and each of them fetches K / V from
// the fabric (measured: 107-139 MB fetched per launch for 37 MB of operands).  Give every XCD one contiguous range of
// the (head-major) tile list instead: the tiles of a head sit on one XCD and re-read K / V from its L2.
__device__ inline void xcd_block(int& tile, int& z) {
    const int nx = gridDim.x, nwg = nx * gridDim.y, orig = blockIdx.y * nx + blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    z = wg / nx; tile = wg - z * nx;
}

// v_exp_f32 / v_log_f32 directly (exp2(-inf) = 0); logits are kept in the log2 domain: p = 2^(s * scale * log2(e) - m)
__device__ inline float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
constexpr float B2S_LOG2E = 1.4426950408889634f, B2S_LN2 = 0.6931471805599453f;
// running-maximum slack of the forward pass: the reference m of a row only moves (and O, l are only rescaled) when a
// tile's maximum exceeds it by more than 2^8; weights stay <= 2^8 * e^(row spread) in between, exact in fp32/bf16 range
constexpr float B2S_LAZY = 8.f;

// guided-attention weight of (query q, key k) for an utterance with inverse lengths iq = 1/qlen, ik = 1/klen
__device__ inline float ga_w(int q, int k, float iq, float ik, float inv2s2) {
    const float d = (float)k * ik - (float)q * iq;
    return 1.f - __expf(-d * d * inv2s2);
}

// attention-weight dropout (b2s_common.h: b2s_keep_w) for a lane that owns a weight row (seed = b2s_wseed of the row): registers r = 0 .. 3 of tile t
// are the quad of keys k0 + t * 16 + lg * 4 + {0 .. 3}; kept elements are multiplied by mul
__device__ inline void drop_quads(f32x4_t (&w)[4], uint32_t seed, int k0, int lg, int ts, float mul) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint32_t y = b2s_wmix(seed, (uint32_t)((k0 >> 2) + t * 4 + lg)), w0 = y * B2S_WC0, w1 = y * B2S_WC1;
        w[t][0] = (int)(int16_t)(uint16_t)w0 >= ts ? w[t][0] * mul : 0.f;
        w[t][1] = (int)(int16_t)(uint16_t)(w0 >> 16) >= ts ? w[t][1] * mul : 0.f;
        w[t][2] = (int)(int16_t)(uint16_t)w1 >= ts ? w[t][2] * mul : 0.f;
        w[t][3] = (int)(int16_t)(uint16_t)(w1 >> 16) >= ts ? w[t][3] * mul : 0.f;
    }
}

// store a transposed accumulator (col = own row li, rows = feature dt*16 + lg*4 + r) as 4 consecutive features
template <typename T, int DH>
__device__ inline void store_rows(T* dst, const f32x4_t (&acc)[DH / 16], float mul, int lg) {
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) {
        T* p = dst + dt * 16 + lg * 4;
        if (sizeof(T) == 4) {
            *reinterpret_cast<float4*>(p) = make_float4(acc[dt][0] * mul, acc[dt][1] * mul, acc[dt][2] * mul, acc[dt][3] * mul);
        } else {
            uint2 u;
            u.x = f2bf2(acc[dt][0] * mul, acc[dt][1] * mul);
            u.y = f2bf2(acc[dt][2] * mul, acc[dt][3] * mul);
            *reinterpret_cast<uint2*>(p) = u;
        }
    }
}

// ================================================================================================ forward
template <typename T, int DH>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 3 : 1) void attn_fwd_kernel(AttnArgs a) {     // (3 workgroups per CU: the kernel is latency-bound, 41 -> 34 us)
    constexpr int LD = DH + AT<T>::PAD, NKS = DH / (sizeof(T) == 2 ? 32 : 4);
    __shared__ __attribute__((aligned(16))) T sK[64 * LD];
    __shared__ __attribute__((aligned(16))) T sV[64 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    int tile_, z;
    xcd_block(tile_, z);
    const int b = z / a.H, h = z - b * a.H;
    // this utterance's rows of the q-side / k-side tensors (AttnArgs::qoff / koff: ragged row offsets; default: the padded [B, L] layout)
    const int qrow0 = a.qoff ? a.qoff[b] : b * a.Lq, Lq = a.qoff ? a.qoff[b + 1] - qrow0 : a.Lq;
    const int krow0 = a.koff ? a.koff[b] : b * a.Lk, Lk = a.koff ? a.koff[b + 1] - krow0 : a.Lk;
    const int qb0 = tile_ * 64, q = qb0 + wave * 16 + li;
    const int qc = min(q, Lq - 1);
    if (a.qskip && qb0 >= a.qskip[b]) {              // a tile of padded query rows (workgroup-uniform)
        if (q < Lq) {
            f32x4_t zero[DH / 16];
#pragma unroll
            for (int dt = 0; dt < DH / 16; ++dt) zero[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            store_rows<T, DH>(reinterpret_cast<T*>(a.out) + ((long)qrow0 + q) * a.ldo + h * DH, zero, 1.f, lg);
            if (lg == 0 && a.lse) a.lse[(long)z * a.Lq + q] = 0.f;
            if (lg == 0 && a.ga_rows) a.ga_rows[(long)z * a.Lq + q] = 0.f;
        }
        return;
    }
    const T* Q = reinterpret_cast<const T*>(a.q) + (long)qrow0 * a.ldq + h * DH;
    const T* K = reinterpret_cast<const T*>(a.k) + (long)krow0 * a.ldk + h * DH;
    const T* V = reinterpret_cast<const T*>(a.v) + (long)krow0 * a.ldv + h * DH;
    typename AT<T>::frag qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = frag_global<T>(Q, a.ldq, qc, ks, lg);

    int kend = Lk;
    if (a.mask_mode & 1) kend = min(kend, a.klen[b]);
    int ktiles = (kend + 63) / 64;
    if (a.mask_mode & 2) ktiles = min(ktiles, (min(qb0 + 63, Lq - 1)) / 64 + 1);
    f32x4_t o[DH / 16];
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;                    // m: reference exponent (log2 domain), l: this lane's part of the row sum
    const bool ga = a.ga_rows != nullptr;
    float g = 0.f, ga_iq = 0.f, ga_ik = 0.f;
    if (ga) { ga_iq = 1.f / (float)max(min(a.qlen[b], Lq), 1); ga_ik = 1.f / (float)max(kend, 1); }
    const float sl2 = a.scale * B2S_LOG2E;
    const int qw0 = qb0 + wave * 16;                 // first query row of this wave
    const uint32_t dseed = b2s_wseed(a.drop, (uint32_t)((long)z * a.Lq + qc));
    const int dts = b2s_wthresh(a.drop);
    TileRegs<T, DH> rk, rv;
    if (ktiles > 0) { tile_fetch<T, DH>(rk, K, a.ldk, 0, Lk, tid); tile_fetch<T, DH>(rv, V, a.ldv, 0, Lk, tid); }
    for (int kt = 0; kt < ktiles; ++kt) {
        const int k0 = kt * 64;
        __syncthreads();
        tile_store<T, DH>(sK, rk, tid);
        tile_store<T, DH>(sV, rv, tid);
        __syncthreads();
        if (kt + 1 < ktiles) { tile_fetch<T, DH>(rk, K, a.ldk, k0 + 64, Lk, tid); tile_fetch<T, DH>(rv, V, a.ldv, k0 + 64, Lk, tid); }
        f32x4_t s[4];
        first_product<T, DH, LD>(s, sK, qf, li, lg);
        // every key of the tile visible to every row of this wave?  (wave-uniform; the common case skips all mask math)
        const bool interior = k0 + 64 <= kend && (!(a.mask_mode & 2) || k0 + 63 <= qw0);
        float mx = -INFINITY;
        if (interior) {
#pragma unroll
            for (int t = 0; t < 4; ++t) mx = fmaxf(fmaxf(mx, fmaxf(s[t][0], s[t][1])), fmaxf(s[t][2], s[t][3]));
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + t * 16 + lg * 4 + r;
                    const bool ok = key < kend && (!(a.mask_mode & 2) || key <= q);
                    s[t][r] = ok ? s[t][r] : -INFINITY;
                    mx = fmaxf(mx, s[t][r]);
                }
        }
        mx = group_max(mx) * sl2;
        const bool grow = mx > m + B2S_LAZY;             // also true for the first finite maximum (m = -inf)
        if (__any(grow)) {
            const float mn = grow ? mx : m;
            const float alpha = mn == m ? 1.f : fast_exp2(m - mn);      // m = -inf -> 0
            m = mn; l *= alpha; g *= alpha;
#pragma unroll
            for (int dt = 0; dt < DH / 16; ++dt) { o[dt][0] *= alpha; o[dt][1] *= alpha; o[dt][2] *= alpha; o[dt][3] *= alpha; }
        }
        const float mref = m == -INFINITY ? 0.f : m;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = fast_exp2(fmaf(s[t][r], sl2, -mref));   // masked: 2^-inf = 0
                l += p;
                s[t][r] = p;
            }
        if (ga) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) g += s[t][r] * ga_w(q, k0 + t * 16 + lg * 4 + r, ga_iq, ga_ik, a.ga_inv2s2);
        }
        if (a.drop.thresh) drop_quads(s, dseed, k0, lg, dts, a.drop.scale);
        SP<T, DH, LD>::run(o, sV, s, li, lg);
    }
    l = group_sum(l);
    if (ga) g = group_sum(g);
    if (q < Lq) {
        const float inv = 1.f / l;
        T* out = reinterpret_cast<T*>(a.out) + ((long)qrow0 + q) * a.ldo + h * DH;
        store_rows<T, DH>(out, o, inv, lg);
        if (lg == 0 && a.lse) a.lse[(long)z * a.Lq + q] = (m + __log2f(l)) * B2S_LN2;
        if (lg == 0 && ga) a.ga_rows[(long)z * a.Lq + q] = q < a.qlen[b] ? g * inv : 0.f;
    }
}

// ================================================================================================ backward
// D[z, q] = sum_d dO[q][d] * O[q][d]
template <typename T>
__global__ void attn_bwd_prep_kernel(const T* dO, const T* O, int ldo, float* D, int H, int Lq, int dh, long rows) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);           // r = (b*Lq + q)*H + h
    if (r >= rows) return;
    const int h = (int)(r % H);
    const long bq = r / H;
    const int q = (int)(bq % Lq), b = (int)(bq / Lq);
    float acc = 0.f;
    for (int d = lane; d < dh; d += 64) acc += TT<T>::ld(dO + bq * ldo + h * dh + d) * TT<T>::ld(O + bq * ldo + h * dh + d);
    acc = wave_sum(acc);
    if (lane == 0) D[((long)b * H + h) * Lq + q] = acc;
}

// workgroups per CU the backward kernels are compiled for (2: up to 256 registers per lane; 3: 168).  dQ fits 166 registers without spilling and is
// latency-bound like the forward kernel: 48.0 -> 43.1 us per launch at 3 (same box, rocprofv3), step 7.11 -> 7.08 ms; dK / dV spills 145 registers
// at 168 (45 -> 118 us) and stays at 2
#ifndef B2S_ATTN_DQ_WPC
#define B2S_ATTN_DQ_WPC 3
#endif
#ifndef B2S_ATTN_DKV_WPC
#define B2S_ATTN_DKV_WPC 2
#endif
// dQ: per workgroup 64 query rows; loops over key tiles
template <typename T, int DH>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? B2S_ATTN_DQ_WPC : 1) void attn_bwd_dq_kernel(AttnArgs a) {
    constexpr int LD = DH + AT<T>::PAD, NKS = DH / (sizeof(T) == 2 ? 32 : 4);
    __shared__ __attribute__((aligned(16))) T sK[64 * LD];
    __shared__ __attribute__((aligned(16))) T sV[64 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    int tile_, z;
    xcd_block(tile_, z);
    const int b = z / a.H, h = z - b * a.H;
    // this utterance's rows of the q-side / k-side tensors (AttnArgs::qoff / koff: ragged row offsets; default: the padded [B, L] layout)
    const int qrow0 = a.qoff ? a.qoff[b] : b * a.Lq, Lq = a.qoff ? a.qoff[b + 1] - qrow0 : a.Lq;
    const int krow0 = a.koff ? a.koff[b] : b * a.Lk, Lk = a.koff ? a.koff[b + 1] - krow0 : a.Lk;
    const int qb0 = tile_ * 64, q = qb0 + wave * 16 + li;
    const int qc = min(q, Lq - 1);
    if (a.qskip && qb0 >= a.qskip[b]) {              // padded query rows: d context is zero there, so is dQ (the dK/dV kernel never reads dsum)
        if (q < Lq) {
            f32x4_t zero[DH / 16];
#pragma unroll
            for (int dt = 0; dt < DH / 16; ++dt) zero[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            store_rows<T, DH>(reinterpret_cast<T*>(a.dq) + ((long)qrow0 + q) * a.lddq + h * DH, zero, 1.f, lg);
            if (lg == 0) a.dsum[(long)z * a.Lq + q] = 0.f;
        }
        return;
    }
    const T* Q = reinterpret_cast<const T*>(a.q) + (long)qrow0 * a.ldq + h * DH;
    const T* K = reinterpret_cast<const T*>(a.k) + (long)krow0 * a.ldk + h * DH;
    const T* V = reinterpret_cast<const T*>(a.v) + (long)krow0 * a.ldv + h * DH;
    const T* dO = reinterpret_cast<const T*>(a.dout) + (long)qrow0 * a.ldo + h * DH;
    typename AT<T>::frag qf[NKS], dof[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) { qf[ks] = frag_global<T>(Q, a.ldq, qc, ks, lg); dof[ks] = frag_global<T>(dO, a.ldo, qc, ks, lg); }
    const float lse = a.lse[(long)z * a.Lq + qc];
    // D[q] = sum_d dO[q][d] * O[q][d], from the same fragments (each lane holds 1/4 of the row; two shuffles finish it)
    float Dq = 0.f;
    {
        const T* O = reinterpret_cast<const T*>(a.oref) + (long)qrow0 * a.ldo + h * DH;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) Dq += frag_dot(frag_global<T>(O, a.ldo, qc, ks, lg), dof[ks]);
        Dq = group_sum(Dq);
    }
    int kend = Lk;
    if (a.mask_mode & 1) kend = min(kend, a.klen[b]);
    // guided attention: dP += c W, D += c * rowsum(P W) on valid query rows
    float gc = 0.f, ga_iq = 0.f, ga_ik = 0.f;
    if (a.ga_rows) {
        const int ql = min(a.qlen[b], Lq);
        if (q < ql) gc = *a.ga_scale;
        ga_iq = 1.f / (float)max(ql, 1); ga_ik = 1.f / (float)max(kend, 1);
        Dq += gc * a.ga_rows[(long)z * a.Lq + qc];
    }
    if (lg == 0 && q < Lq) a.dsum[(long)z * a.Lq + q] = Dq;              // the dK/dV kernel reads it
    int ktiles = (kend + 63) / 64;
    if (a.mask_mode & 2) ktiles = min(ktiles, (min(qb0 + 63, Lq - 1)) / 64 + 1);
    f32x4_t dq[DH / 16];
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) dq[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const uint32_t dseed = b2s_wseed(a.drop, (uint32_t)((long)z * a.Lq + qc));
    const int dts = b2s_wthresh(a.drop);
    const float sl2 = a.scale * B2S_LOG2E, lse2 = lse * B2S_LOG2E;
    const int qw0 = qb0 + wave * 16;
    TileRegs<T, DH> rk, rv;
    if (ktiles > 0) { tile_fetch<T, DH>(rk, K, a.ldk, 0, Lk, tid); tile_fetch<T, DH>(rv, V, a.ldv, 0, Lk, tid); }
    for (int kt = 0; kt < ktiles; ++kt) {
        const int k0 = kt * 64;
        __syncthreads();
        tile_store<T, DH>(sK, rk, tid);
        tile_store<T, DH>(sV, rv, tid);
        __syncthreads();
        if (kt + 1 < ktiles) { tile_fetch<T, DH>(rk, K, a.ldk, k0 + 64, Lk, tid); tile_fetch<T, DH>(rv, V, a.ldv, k0 + 64, Lk, tid); }
        f32x4_t s[4], dp[4];
        first_product<T, DH, LD>(s, sK, qf, li, lg);
        first_product<T, DH, LD>(dp, sV, dof, li, lg);
        const bool interior = k0 + 64 <= kend && (!(a.mask_mode & 2) || k0 + 63 <= qw0);
        if (interior) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[t][r] = fast_exp2(fmaf(s[t][r], sl2, -lse2));
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + t * 16 + lg * 4 + r;
                    const bool ok = key < kend && (!(a.mask_mode & 2) || key <= q);
                    s[t][r] = ok ? fast_exp2(fmaf(s[t][r], sl2, -lse2)) : 0.f;
                }
        }
        if (a.drop.thresh) drop_quads(dp, dseed, k0, lg, dts, a.drop.scale);
        if (__any(gc != 0.f)) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) dp[t][r] += gc * ga_w(q, k0 + t * 16 + lg * 4 + r, ga_iq, ga_ik, a.ga_inv2s2);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[t][r] = s[t][r] * (dp[t][r] - Dq) * a.scale;
        SP<T, DH, LD>::run(dq, sK, s, li, lg);
    }
    if (q < Lq) store_rows<T, DH>(reinterpret_cast<T*>(a.dq) + ((long)qrow0 + q) * a.lddq + h * DH, dq, 1.f, lg);
}

// dK, dV: per workgroup 64 keys; loops over query tiles
template <typename T, int DH>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? B2S_ATTN_DKV_WPC : 1) void attn_bwd_dkv_kernel(AttnArgs a) {
    constexpr int LD = DH + AT<T>::PAD, NKS = DH / (sizeof(T) == 2 ? 32 : 4);
    __shared__ __attribute__((aligned(16))) T sQ[64 * LD];
    __shared__ __attribute__((aligned(16))) T sO[64 * LD];
    __shared__ __attribute__((aligned(16))) float sL[64], sD[64];
    __shared__ __attribute__((aligned(16))) uint32_t sS[64];             // dropout seeds of the tile's query rows (b2s_common.h: b2s_keep_w)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    int tile_, z;
    xcd_block(tile_, z);
    const int b = z / a.H, h = z - b * a.H;
    // this utterance's rows of the q-side / k-side tensors (AttnArgs::qoff / koff: ragged row offsets; default: the padded [B, L] layout)
    const int qrow0 = a.qoff ? a.qoff[b] : b * a.Lq, Lq = a.qoff ? a.qoff[b + 1] - qrow0 : a.Lq;
    const int krow0 = a.koff ? a.koff[b] : b * a.Lk, Lk = a.koff ? a.koff[b + 1] - krow0 : a.Lk;
    const int kb0 = tile_ * 64, key = kb0 + wave * 16 + li;
    const int kc = min(key, Lk - 1);
    const T* Q = reinterpret_cast<const T*>(a.q) + (long)qrow0 * a.ldq + h * DH;
    const T* K = reinterpret_cast<const T*>(a.k) + (long)krow0 * a.ldk + h * DH;
    const T* V = reinterpret_cast<const T*>(a.v) + (long)krow0 * a.ldv + h * DH;
    const T* dO = reinterpret_cast<const T*>(a.dout) + (long)qrow0 * a.ldo + h * DH;
    typename AT<T>::frag kf[NKS], vf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) { kf[ks] = frag_global<T>(K, a.ldk, kc, ks, lg); vf[ks] = frag_global<T>(V, a.ldv, kc, ks, lg); }
    int kend = Lk;
    if (a.mask_mode & 1) kend = min(kend, a.klen[b]);
    const bool key_ok = key < kend;
    float gc = 0.f, ga_iq = 0.f, ga_ik = 0.f;
    int ga_ql = 0;
    if (a.ga_rows) {
        ga_ql = min(a.qlen[b], Lq);
        gc = *a.ga_scale;
        ga_iq = 1.f / (float)max(ga_ql, 1); ga_ik = 1.f / (float)max(kend, 1);
    }
    f32x4_t dk[DH / 16], dv[DH / 16];
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) { dk[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
    int qtiles = (Lq + 63) / 64;
    if (a.qskip) qtiles = min(qtiles, (a.qskip[b] + 63) / 64);             // tiles of padded query rows contribute nothing (d context = 0)
    const int qt0 = (a.mask_mode & 2) ? kb0 / 64 : 0;          // causal: queries before this key tile never see it
    TileRegs<T, DH> rq, ro;
    float r_l = 0.f, r_d = 0.f;
    uint32_t r_s = 0;
    const float sl2 = a.scale * B2S_LOG2E;
    const int kw0 = kb0 + wave * 16;                 // first key of this wave
    const uint32_t zq = (uint32_t)z * (uint32_t)a.Lq;
    // this key's part of the dropout rule: quad offset, multiplier, the shift that brings its field to the top 16 bits
    const uint32_t dxk = (uint32_t)(kc >> 2) * 0x9E3779B1u, dwc = (kc & 2) ? B2S_WC1 : B2S_WC0, dsh = (kc & 1) ? 0u : 16u;
    const int dts32 = (int)((uint32_t)b2s_wthresh(a.drop) << 16);
    if (qt0 < qtiles) {
        tile_fetch<T, DH>(rq, Q, a.ldq, qt0 * 64, Lq, tid); tile_fetch<T, DH>(ro, dO, a.ldo, qt0 * 64, Lq, tid);
        if (tid < 64) {
            const int qq = min(qt0 * 64 + tid, Lq - 1);
            r_l = a.lse[(long)z * a.Lq + qq] * B2S_LOG2E; r_d = a.dsum[(long)z * a.Lq + qq];
            if (a.drop.thresh) r_s = b2s_wseed(a.drop, zq + (uint32_t)qq);
        }
    }
    for (int qt = qt0; qt < qtiles; ++qt) {
        const int q0 = qt * 64;
        __syncthreads();
        tile_store<T, DH>(sQ, rq, tid);
        tile_store<T, DH>(sO, ro, tid);
        if (tid < 64) { sL[tid] = r_l; sD[tid] = r_d; sS[tid] = r_s; }
        __syncthreads();
        if (qt + 1 < qtiles) {
            tile_fetch<T, DH>(rq, Q, a.ldq, q0 + 64, Lq, tid); tile_fetch<T, DH>(ro, dO, a.ldo, q0 + 64, Lq, tid);
            if (tid < 64) {
                const int qq = min(q0 + 64 + tid, Lq - 1);
                r_l = a.lse[(long)z * a.Lq + qq] * B2S_LOG2E; r_d = a.dsum[(long)z * a.Lq + qq];
                if (a.drop.thresh) r_s = b2s_wseed(a.drop, zq + (uint32_t)qq);
            }
        }
        f32x4_t s[4], dp[4], pd[4];
        first_product<T, DH, LD>(s, sQ, kf, li, lg);          // s[t][r] = S[q = q0 + t*16 + lg*4 + r][key = own]
        first_product<T, DH, LD>(dp, sO, vf, li, lg);
        // all 16 keys of this wave valid and visible to all 64 queries of the tile?  (wave-uniform)
        const bool interior = kw0 + 16 <= kend && q0 + 64 <= Lq && (!(a.mask_mode & 2) || kw0 + 15 <= q0);
        if (interior) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4_t lq = *reinterpret_cast<const f32x4_t*>(sL + t * 16 + lg * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) s[t][r] = fast_exp2(fmaf(s[t][r], sl2, -lq[r]));
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4_t lq = *reinterpret_cast<const f32x4_t*>(sL + t * 16 + lg * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qq = q0 + t * 16 + lg * 4 + r;
                    const bool ok = key_ok && qq < Lq && (!(a.mask_mode & 2) || key <= qq);
                    s[t][r] = ok ? fast_exp2(fmaf(s[t][r], sl2, -lq[r])) : 0.f;
                }
            }
        }
        if (a.drop.thresh) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
                const u32x4_t sd4 = *reinterpret_cast<const u32x4_t*>(sS + t * 16 + lg * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t x = sd4[r] + dxk;                                // seed of query row q0 + t 16 + lg 4 + r, + this key's quad
                    const bool keep = (int)(((x ^ (x >> 16)) * dwc) << dsh) >= dts32;
                    dp[t][r] = keep ? dp[t][r] * a.drop.scale : 0.f;
                    pd[t][r] = keep ? s[t][r] * a.drop.scale : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) pd[t] = s[t];
        }
        if (gc != 0.f) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qq = q0 + t * 16 + lg * 4 + r;
                    dp[t][r] += qq < ga_ql ? gc * ga_w(qq, key, ga_iq, ga_ik, a.ga_inv2s2) : 0.f;
                }
        }
        SP<T, DH, LD>::run(dv, sO, pd, li, lg);                // dV^T[d][key] += sum_q dO[q][d] * Pd[q][key]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4_t dsum4 = *reinterpret_cast<const f32x4_t*>(sD + t * 16 + lg * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) s[t][r] = s[t][r] * (dp[t][r] - dsum4[r]) * a.scale;
        }
        SP<T, DH, LD>::run(dk, sQ, s, li, lg);                 // dK^T[d][key] += sum_q Q[q][d]  * dS[q][key]
    }
    if (key < Lk) {
        store_rows<T, DH>(reinterpret_cast<T*>(a.dk) + ((long)krow0 + key) * a.lddk + h * DH, dk, 1.f, lg);
        store_rows<T, DH>(reinterpret_cast<T*>(a.dv) + ((long)krow0 + key) * a.lddv + h * DH, dv, 1.f, lg);
    }
}

// ================================================================================================ resident keys (Lk <= 128, no causal mask)
// Encoder-decoder attention reads a SHORT key sequence (the utterance's <= 128 text bytes) against ~600 query frames.  In the generic
// kernels above a workgroup's whole life is then prologue + two key tiles + epilogue, and the 44 KB of K / V of a (batch, head) are staged
// through LDS once per 64-query tile -- ten times.  Here K and V of the (batch, head) stay in LDS (2 x 128 rows), a workgroup walks
// `tpw` consecutive query tiles, and with every key on chip the softmax is single-pass: no running maximum, no rescale, and NO barrier
// after the prologue -- a wave's 16 query rows depend on nothing another wave does, so the four waves drift apart and hide each other's
// latencies.  Forward and dQ; dK / dV keeps the generic kernel (it loops over queries, its keys were always resident).
// Same arithmetic as the generic kernels up to the summation order of the row sum (results agree to fp32 rounding; the saved
// log-sum-exp is the same quantity), same dropout / guided-attention / padded-tile semantics.
constexpr int RES_KEYS = 128;

// K, V rows [0, 128) of one (batch, head) -> LDS images [128][LD] (zero beyond Lk), then one barrier
template <typename T, int DH>
__device__ inline void res_load_kv(T* sK, T* sV, const T* K, const T* V, const AttnArgs& a, int Lk, int nkt, int tid) {
    constexpr int LD = DH + AT<T>::PAD;
    TileRegs<T, DH> rk, rv;
    for (int kt = 0; kt < nkt; ++kt) {
        tile_fetch<T, DH>(rk, K, a.ldk, kt * 64, Lk, tid); tile_fetch<T, DH>(rv, V, a.ldv, kt * 64, Lk, tid);
        tile_store<T, DH>(sK + kt * 64 * LD, rk, tid); tile_store<T, DH>(sV + kt * 64 * LD, rv, tid);
    }
    __syncthreads();
}

template <typename T, int DH>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 2 : 1) void attn_fwd_res_kernel(AttnArgs a, int tpw) {
    constexpr int LD = DH + AT<T>::PAD, NKS = DH / (sizeof(T) == 2 ? 32 : 4);
    extern __shared__ __attribute__((aligned(16))) unsigned char res_smem[];
    T* sK = reinterpret_cast<T*>(res_smem);
    T* sV = sK + RES_KEYS * LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    int chunk, z;
    xcd_block(chunk, z);
    const int b = z / a.H, h = z - b * a.H;
    // this utterance's rows of the q-side / k-side tensors (AttnArgs::qoff / koff: ragged row offsets; default: the padded [B, L] layout)
    const int qrow0 = a.qoff ? a.qoff[b] : b * a.Lq, Lq = a.qoff ? a.qoff[b + 1] - qrow0 : a.Lq;
    const int krow0 = a.koff ? a.koff[b] : b * a.Lk, Lk = a.koff ? a.koff[b + 1] - krow0 : a.Lk;
    const T* Q = reinterpret_cast<const T*>(a.q) + (long)qrow0 * a.ldq + h * DH;
    const T* K = reinterpret_cast<const T*>(a.k) + (long)krow0 * a.ldk + h * DH;
    const T* V = reinterpret_cast<const T*>(a.v) + (long)krow0 * a.ldv + h * DH;
    int kend = Lk;
    if (a.mask_mode & 1) kend = min(kend, a.klen[b]);
    const int nkt = (kend + 63) / 64;                              // 0, 1 or 2 key tiles carry valid keys
    const int ntiles = (Lq + 63) / 64;
    const int t_begin = chunk * tpw, t_end = min(t_begin + tpw, ntiles);
    // tiles of padded query rows need no keys: a chunk that holds nothing else skips the K / V load as well
    const int q_live = a.qskip ? min(a.qskip[b], Lq) : Lq;
    if (t_begin * 64 < q_live) res_load_kv<T, DH>(sK, sV, K, V, a, Lk, nkt, tid);
    const bool ga = a.ga_rows != nullptr;
    float ga_iq = 0.f, ga_ik = 0.f;
    if (ga) { ga_iq = 1.f / (float)max(min(a.qlen[b], Lq), 1); ga_ik = 1.f / (float)max(kend, 1); }
    const float sl2 = a.scale * B2S_LOG2E;
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int qb0 = tile * 64, q = qb0 + wave * 16 + li, qc = min(q, Lq - 1);
        f32x4_t o[DH / 16];
#pragma unroll
        for (int dt = 0; dt < DH / 16; ++dt) o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (qb0 >= q_live && a.qskip) {                            // a tile of padded query rows (workgroup-uniform)
            if (q < Lq) {
                store_rows<T, DH>(reinterpret_cast<T*>(a.out) + ((long)qrow0 + q) * a.ldo + h * DH, o, 1.f, lg);
                if (lg == 0 && a.lse) a.lse[(long)z * a.Lq + q] = 0.f;
                if (lg == 0 && ga) a.ga_rows[(long)z * a.Lq + q] = 0.f;
            }
            continue;
        }
        typename AT<T>::frag qf[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = frag_global<T>(Q, a.ldq, qc, ks, lg);
        f32x4_t s[2][4];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            if (kt >= nkt) {
#pragma unroll
                for (int t = 0; t < 4; ++t) s[kt][t] = (f32x4_t){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                continue;
            }
            first_product<T, DH, LD>(s[kt], sK + kt * 64 * LD, qf, li, lg);
            if (kt * 64 + 64 > kend) {                             // the boundary tile: keys >= kend are masked
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[kt][t][r] = (kt * 64 + t * 16 + lg * 4 + r) < kend ? s[kt][t][r] : -INFINITY;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) mx = fmaxf(fmaxf(mx, fmaxf(s[kt][t][0], s[kt][t][1])), fmaxf(s[kt][t][2], s[kt][t][3]));
        }
        mx = group_max(mx) * sl2;
        const float mref = mx == -INFINITY ? 0.f : mx;
        float l = 0.f, g = 0.f;
        const uint32_t dseed = b2s_wseed(a.drop, (uint32_t)((long)z * a.Lq + qc));
    const int dts = b2s_wthresh(a.drop);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            if (kt >= nkt) continue;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = fast_exp2(fmaf(s[kt][t][r], sl2, -mref));      // masked: 2^-inf = 0
                    l += p;
                    s[kt][t][r] = p;
                }
            if (ga) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) g += s[kt][t][r] * ga_w(q, kt * 64 + t * 16 + lg * 4 + r, ga_iq, ga_ik, a.ga_inv2s2);
            }
            if (a.drop.thresh) drop_quads(s[kt], dseed, kt * 64, lg, dts, a.drop.scale);
            SP<T, DH, LD>::run(o, sV + kt * 64 * LD, s[kt], li, lg);
        }
        l = group_sum(l);
        if (ga) g = group_sum(g);
        if (q < Lq) {
            const float inv = 1.f / l;
            store_rows<T, DH>(reinterpret_cast<T*>(a.out) + ((long)qrow0 + q) * a.ldo + h * DH, o, inv, lg);
            if (lg == 0 && a.lse) a.lse[(long)z * a.Lq + q] = (mref + __log2f(l)) * B2S_LN2;
            if (lg == 0 && ga) a.ga_rows[(long)z * a.Lq + q] = q < a.qlen[b] ? g * inv : 0.f;
        }
    }
}

template <typename T, int DH>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 2 : 1) void attn_bwd_dq_res_kernel(AttnArgs a, int tpw) {
    constexpr int LD = DH + AT<T>::PAD, NKS = DH / (sizeof(T) == 2 ? 32 : 4);
    extern __shared__ __attribute__((aligned(16))) unsigned char res_smem[];
    T* sK = reinterpret_cast<T*>(res_smem);
    T* sV = sK + RES_KEYS * LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    int chunk, z;
    xcd_block(chunk, z);
    const int b = z / a.H, h = z - b * a.H;
    // this utterance's rows of the q-side / k-side tensors (AttnArgs::qoff / koff: ragged row offsets; default: the padded [B, L] layout)
    const int qrow0 = a.qoff ? a.qoff[b] : b * a.Lq, Lq = a.qoff ? a.qoff[b + 1] - qrow0 : a.Lq;
    const int krow0 = a.koff ? a.koff[b] : b * a.Lk, Lk = a.koff ? a.koff[b + 1] - krow0 : a.Lk;
    const T* Q = reinterpret_cast<const T*>(a.q) + (long)qrow0 * a.ldq + h * DH;
    const T* K = reinterpret_cast<const T*>(a.k) + (long)krow0 * a.ldk + h * DH;
    const T* V = reinterpret_cast<const T*>(a.v) + (long)krow0 * a.ldv + h * DH;
    const T* dO = reinterpret_cast<const T*>(a.dout) + (long)qrow0 * a.ldo + h * DH;
    const T* O = reinterpret_cast<const T*>(a.oref) + (long)qrow0 * a.ldo + h * DH;
    int kend = Lk;
    if (a.mask_mode & 1) kend = min(kend, a.klen[b]);
    const int nkt = (kend + 63) / 64;
    const int ntiles = (Lq + 63) / 64;
    const int t_begin = chunk * tpw, t_end = min(t_begin + tpw, ntiles);
    const int q_live = a.qskip ? min(a.qskip[b], Lq) : Lq;
    if (t_begin * 64 < q_live) res_load_kv<T, DH>(sK, sV, K, V, a, Lk, nkt, tid);
    float gc0 = 0.f, ga_iq = 0.f, ga_ik = 0.f;
    int ga_ql = 0;
    if (a.ga_rows) {
        ga_ql = min(a.qlen[b], Lq);
        gc0 = *a.ga_scale;
        ga_iq = 1.f / (float)max(ga_ql, 1); ga_ik = 1.f / (float)max(kend, 1);
    }
    const float sl2 = a.scale * B2S_LOG2E;
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int qb0 = tile * 64, q = qb0 + wave * 16 + li, qc = min(q, Lq - 1);
        f32x4_t dq[DH / 16];
#pragma unroll
        for (int dt = 0; dt < DH / 16; ++dt) dq[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (qb0 >= q_live && a.qskip) {                            // padded query rows: d context is zero there, so is dQ
            if (q < Lq) {
                store_rows<T, DH>(reinterpret_cast<T*>(a.dq) + ((long)qrow0 + q) * a.lddq + h * DH, dq, 1.f, lg);
                if (lg == 0) a.dsum[(long)z * a.Lq + q] = 0.f;
            }
            continue;
        }
        typename AT<T>::frag qf[NKS], dof[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) { qf[ks] = frag_global<T>(Q, a.ldq, qc, ks, lg); dof[ks] = frag_global<T>(dO, a.ldo, qc, ks, lg); }
        const float lse2 = a.lse[(long)z * a.Lq + qc] * B2S_LOG2E;
        float Dq = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) Dq += frag_dot(frag_global<T>(O, a.ldo, qc, ks, lg), dof[ks]);
        Dq = group_sum(Dq);
        float gc = 0.f;
        if (a.ga_rows) {
            if (q < ga_ql) gc = gc0;
            Dq += gc * a.ga_rows[(long)z * a.Lq + qc];
        }
        if (lg == 0 && q < Lq) a.dsum[(long)z * a.Lq + q] = Dq;              // the dK/dV kernel reads it
        const uint32_t dseed = b2s_wseed(a.drop, (uint32_t)((long)z * a.Lq + qc));
    const int dts = b2s_wthresh(a.drop);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            if (kt >= nkt) continue;
            f32x4_t s[4], dp[4];
            first_product<T, DH, LD>(s, sK + kt * 64 * LD, qf, li, lg);
            first_product<T, DH, LD>(dp, sV + kt * 64 * LD, dof, li, lg);
            const bool interior = kt * 64 + 64 <= kend;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = interior || (kt * 64 + t * 16 + lg * 4 + r) < kend;
                    s[t][r] = ok ? fast_exp2(fmaf(s[t][r], sl2, -lse2)) : 0.f;
                }
            if (a.drop.thresh) drop_quads(dp, dseed, kt * 64, lg, dts, a.drop.scale);
            if (__any(gc != 0.f)) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dp[t][r] += gc * ga_w(q, kt * 64 + t * 16 + lg * 4 + r, ga_iq, ga_ik, a.ga_inv2s2);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[t][r] = s[t][r] * (dp[t][r] - Dq) * a.scale;
            SP<T, DH, LD>::run(dq, sK + kt * 64 * LD, s, li, lg);
        }
        if (q < Lq) store_rows<T, DH>(reinterpret_cast<T*>(a.dq) + ((long)qrow0 + q) * a.lddq + h * DH, dq, 1.f, lg);
    }
}

// alignment rows on demand: align[z][k][q] = softmax weight, recomputed from q, k and the saved log-sum-exp
template <typename T>
__global__ __launch_bounds__(256) void attn_align_kernel(AttnArgs a, float* align, int dh) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);           // r = z*Lq + q
    if (r >= (long)a.B * a.H * a.Lq) return;
    const int q = (int)(r % a.Lq), z = (int)(r / a.Lq), b = z / a.H, h = z - b * a.H;
    const int qrow0 = a.qoff ? a.qoff[b] : b * a.Lq, Lqb = a.qoff ? a.qoff[b + 1] - qrow0 : a.Lq;
    const int krow0 = a.koff ? a.koff[b] : b * a.Lk, Lkb = a.koff ? a.koff[b + 1] - krow0 : a.Lk;
    if (q >= Lqb) {                                    // ragged layout: the row does not exist (the caller declared padded rows unobserved)
        for (int k = lane; k < a.Lk; k += 64) align[((long)z * a.Lk + k) * a.Lq + q] = 0.f;
        return;
    }
    const T* Q = reinterpret_cast<const T*>(a.q) + ((long)qrow0 + q) * a.ldq + h * dh;
    const T* K = reinterpret_cast<const T*>(a.k) + (long)krow0 * a.ldk + h * dh;
    int kend = Lkb;
    if (a.mask_mode & 1) kend = min(kend, a.klen[b]);
    if (a.mask_mode & 2) kend = min(kend, q + 1);
    float lse = a.lse[r];
    if (a.qskip && (q & ~63) >= a.qskip[b]) {          // a row the forward kernel skipped: its log-sum-exp is computed here
        float mx = -INFINITY;
        for (int k = lane; k < kend; k += 64) {
            float s = 0.f;
            for (int d = 0; d < dh; ++d) s += TT<T>::ld(Q + d) * TT<T>::ld(K + (long)k * a.ldk + d);
            mx = fmaxf(mx, s * a.scale);
        }
        mx = wave_max(mx);
        float sum = 0.f;
        for (int k = lane; k < kend; k += 64) {
            float s = 0.f;
            for (int d = 0; d < dh; ++d) s += TT<T>::ld(Q + d) * TT<T>::ld(K + (long)k * a.ldk + d);
            sum += __expf(s * a.scale - mx);
        }
        lse = mx + __logf(wave_sum(sum));
    }
    for (int k = lane; k < a.Lk; k += 64) {
        float p = 0.f;
        if (k < kend) {
            float s = 0.f;
            for (int d = 0; d < dh; ++d) s += TT<T>::ld(Q + d) * TT<T>::ld(K + (long)k * a.ldk + d);
            p = __expf(s * a.scale - lse);
        }
        align[((long)z * a.Lk + k) * a.Lq + q] = p;
    }
}

// resident-key kernels: short key sequences without the causal mask (the encoder-decoder attention, and a short encoder's self-attention)
inline bool res_ok(const AttnArgs& a) { return !(a.mask_mode & 2) && a.Lk <= RES_KEYS && a.Lq > 64; }
// query tiles per workgroup: 2 -> B*H*ceil(tiles/2) workgroups (560 at B = 14, T = 582): one resident round at 2 workgroups per CU
#ifdef B2S_LAB
static const int g_res_tpw = getenv("B2S_LAB_ATTN_TPW") ? atoi(getenv("B2S_LAB_ATTN_TPW")) : 2;       // (0: generic kernels)
#else
constexpr int g_res_tpw = 2;
#endif
template <typename T, int DH>
int launch_res(const AttnArgs& a, int which, hipStream_t st) {
    constexpr size_t smem = (size_t)2 * RES_KEYS * (DH + AT<T>::PAD) * sizeof(T);
    static std::once_flag once;
    static hipError_t err = hipSuccess;
    std::call_once(once, [] {
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_res_kernel<T, DH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (err == hipSuccess)
            err = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_res_kernel<T, DH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    });
    B2S_HIP(err);
    dim3 grid(cdiv(cdiv(a.Lq, 64), g_res_tpw), a.B * a.H);
    if (which == 0) hipLaunchKernelGGL((attn_fwd_res_kernel<T, DH>), grid, dim3(256), smem, st, a, g_res_tpw);
    else hipLaunchKernelGGL((attn_bwd_dq_res_kernel<T, DH>), grid, dim3(256), smem, st, a, g_res_tpw);
    B2S_LAUNCH_CHECK();
    return 0;
}
template <typename T, int DH>
int launch_dh(const AttnArgs& a, int which, hipStream_t st) {
    if (which < 2 && g_res_tpw > 0 && res_ok(a)) return launch_res<T, DH>(a, which, st);
    if (which == 0) {
        dim3 grid(cdiv(a.Lq, 64), a.B * a.H);
        hipLaunchKernelGGL((attn_fwd_kernel<T, DH>), grid, dim3(256), 0, st, a);
    } else if (which == 1) {
        dim3 grid(cdiv(a.Lq, 64), a.B * a.H);
        hipLaunchKernelGGL((attn_bwd_dq_kernel<T, DH>), grid, dim3(256), 0, st, a);
    } else {
        dim3 grid(cdiv(a.Lk, 64), a.B * a.H);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, DH>), grid, dim3(256), 0, st, a);
    }
    B2S_LAUNCH_CHECK();
    return 0;
}
template <typename T>
int launch_t(const AttnArgs& a, int dh, int which, hipStream_t st) {
    switch (dh) {
        case 32: return launch_dh<T, 32>(a, which, st);
        case 64: return launch_dh<T, 64>(a, which, st);
        case 96: return launch_dh<T, 96>(a, which, st);
    }
    return b2s_fail(__FILE__, __LINE__, "fused attention supports head sizes 32/64/96 (got %d)", dh);
}
int check(const AttnArgs& a, int dtype, int dh) {
    const int ve = dtype ? 8 : 4;
    B2S_CHECK(a.q && a.k && a.v && a.B > 0 && a.H > 0 && a.Lq > 0 && a.Lk > 0, "attention: bad argument");
    B2S_CHECK(a.ldq % ve == 0 && a.ldk % ve == 0 && a.ldv % ve == 0 && (a.ldo % ve == 0), "attention: leading dimensions must be multiples of %d", ve);
    B2S_CHECK(!(a.mask_mode & 1) || a.klen, "attention: key-length mask needs klen");
    B2S_CHECK(!a.ga_rows || a.qlen, "attention: the guided-attention term needs query lengths");
    (void)dh;
    return 0;
}
}  // namespace

bool b2s_flash_supported(int dh) { return dh == 32 || dh == 64 || dh == 96; }

// bf16, causal (the decoder's self-attention: ~10 key tiles per query block): the 32x32x16 kernels of attention32.hip.  The short-key attention
// without a causal mask (encoder-decoder attention: 2 .. 4 key tiles) stays on the kernels of this file -- same box, us per launch at (14, 8, 582, 114),
// dropout on: forward 16.7 (resident keys) against 20.3, dQ 25 against 25, dK / dV 25 against 44 (one 128-key block per head leaves 112
// workgroups); at 256 keys forward 27.3 / 26.5, backward 71 / 79 (profiles/NOTES_r06.md).  Lab builds: B2S_LAB_ATTN32 is a bit mask (1: causal, 2 / 4 / 8: non-causal forward / dQ / dK dV).
#ifdef B2S_LAB
static const int g_attn32 = getenv("B2S_LAB_ATTN32") ? atoi(getenv("B2S_LAB_ATTN32")) : 1;     // bit 0: causal, bits 1 / 2 / 3: non-causal forward / dQ / dK dV
#else
constexpr int g_attn32 = 1;
#endif
static inline bool use32(int dtype, const AttnArgs& a, int dh, int which) {
    if (!dtype || !b2s_flash32_supported(dh) || a.ga_rows) return false;
    return (a.mask_mode & 2) ? (g_attn32 & 1) != 0 : ((g_attn32 >> (1 + which)) & 1) != 0;
}

int b2s_flash_fwd(int dtype, const AttnArgs& a, int dh, hipStream_t st) {
    B2S_TRY(check(a, dtype, dh));
    B2S_CHECK(a.out, "attention: null output");
    if (use32(dtype, a, dh, 0)) return b2s_flash32_launch(a, dh, 0, st);
    return dtype ? launch_t<bf16_t>(a, dh, 0, st) : launch_t<float>(a, dh, 0, st);
}
int b2s_flash_bwd(int dtype, const AttnArgs& a_in, int dh, const void* O, hipStream_t st, hipStream_t st_dkv, hipEvent_t ev_dq) {
    B2S_TRY(check(a_in, dtype, dh));
    B2S_CHECK(a_in.dout && a_in.dq && a_in.dk && a_in.dv && a_in.lse && a_in.dsum && O, "attention backward: null argument");
    B2S_CHECK(!a_in.ga_rows || a_in.ga_scale, "attention backward: the guided-attention term needs its scale");
    AttnArgs a = a_in;
    a.oref = O;
    // (both backward kernels write dsum / read it the same way, so the two families mix freely)
    if (use32(dtype, a, dh, 1)) { B2S_TRY(b2s_flash32_launch(a, dh, 1, st)); }
    else { B2S_TRY(dtype ? launch_t<bf16_t>(a, dh, 1, st) : launch_t<float>(a, dh, 1, st)); }
    if (st_dkv) {
        B2S_CHECK(ev_dq, "attention backward: the side stream needs an event");
        B2S_HIP(hipEventRecord(ev_dq, st));
        B2S_HIP(hipStreamWaitEvent(st_dkv, ev_dq, 0));
        st = st_dkv;
    }
    if (use32(dtype, a, dh, 2)) return b2s_flash32_launch(a, dh, 2, st);
    return dtype ? launch_t<bf16_t>(a, dh, 2, st) : launch_t<float>(a, dh, 2, st);
}
int b2s_flash_align(int dtype, const AttnArgs& a, int dh, float* align, hipStream_t st) {
    B2S_CHECK(a.q && a.k && a.lse && align, "attention align: null argument");
    const long rows = (long)a.B * a.H * a.Lq;
    if (dtype) hipLaunchKernelGGL((attn_align_kernel<bf16_t>), dim3(cdiv(rows, 4)), dim3(256), 0, st, a, align, dh);
    else hipLaunchKernelGGL((attn_align_kernel<float>), dim3(cdiv(rows, 4)), dim3(256), 0, st, a, align, dh);
    B2S_LAUNCH_CHECK();
    return 0;
}
