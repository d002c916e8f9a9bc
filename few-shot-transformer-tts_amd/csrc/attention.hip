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
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));      // (a first-class vector value: HIP's uint4 is a struct, and copies of it through a
                                                                    // register array that lives across loop iterations ended up in scratch memory)
template <typename T, int DH> struct TileRegs { u32x4_t v[64 * (DH / AT<T>::VE) / 256]; };
// Every load is UNCONDITIONAL: rows past the end re-read the last valid row (their logits are masked / their weights are exact zeros,
// so any finite row will do).  A load under a condition -- per lane or wave-uniform -- makes hipcc branch around it, and its wait-count
// pass then has to assume that the loads issued since an older load may not have happened: it waited vmcnt(0) for the (long landed)
// query fragments at the first MFMA of every tile, i.e. for the prefetch it had just issued -- a full memory round trip per tile.
template <typename T, int DH>
__device__ inline void tile_fetch(TileRegs<T, DH>& r, const T* src, long ld, int row0, int nrows, int tid) {
    constexpr int VE = AT<T>::VE, VPR = DH / VE, NV = 64 * VPR / 256;
    static_assert(64 * VPR % 256 == 0, "tile must split evenly over 256 threads");
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * 256, rr = v / VPR, c = (v - rr * VPR) * VE;
        r.v[i] = *reinterpret_cast<const u32x4_t*>(src + (long)min(row0 + rr, nrows - 1) * ld + c);
    }
}
// s_waitcnt vmcnt(0) as a compiler-visible instruction (hipcc's wait-count pass models it; an inline-asm wait it does not): everything
// loaded so far -- the wave's own fragments, the first tile -- has landed, so no later use of those registers needs a vmcnt wait that
// would also drain the prefetch in flight
__device__ __forceinline__ void wait_all_loads() { __builtin_amdgcn_s_waitcnt(0x0F70); }
template <typename T, int DH>
__device__ inline void tile_store(T* lds, const TileRegs<T, DH>& r, int tid) {
    constexpr int VE = AT<T>::VE, LD = DH + AT<T>::PAD, VPR = DH / VE, NV = 64 * VPR / 256;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * 256, rr = v / VPR, c = (v - rr * VPR) * VE;
        *reinterpret_cast<u32x4_t*>(lds + rr * LD + c) = r.v[i];
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

// the same for RB row blocks that share the tile: every transposed tile fragment feeds RB MFMAs
template <typename T, int DH, int LD, int RB> struct SPR;
template <int DH, int LD, int RB> struct SPR<float, DH, LD, RB> {
    __device__ static inline void run(f32x4_t (&acc)[RB][DH / 16], const float* tile, const f32x4_t (&w)[RB][4], int li, int lg) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) second_product_f32<DH, LD>(acc[rb], tile, w[rb], li, lg);
    }
    // rows [kb*32, kb*32 + 32) of the tile only: w[rb][u] belongs to tile rows (2 kb + u)*16 + lg*4 + r
    __device__ static inline void half(f32x4_t (&acc)[RB][DH / 16], const float* tile, const f32x4_t (&w)[RB][2], int kb, int li, int lg) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* row = tile + ((2 * kb + u) * 16 + lg * 4 + r) * LD + li;
#pragma unroll
                for (int dt = 0; dt < DH / 16; ++dt) {
                    const float av = row[dt * 16];
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) acc[rb][dt] = mma(av, w[rb][u][r], acc[rb][dt]);
                }
            }
    }
};
template <int DH, int LD, int RB> struct SPR<bf16_t, DH, LD, RB> {
    __device__ static inline void run(f32x4_t (&acc)[RB][DH / 16], const bf16_t* tile, const f32x4_t (&w)[RB][4], int li, int lg) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            bf16x8_t b[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) b[rb] = pack8(w[rb][2 * kb], w[rb][2 * kb + 1]);
#pragma unroll
            for (int dt = 0; dt < DH / 16; ++dt) {
                const bf16x8_t f = frag_tr<LD>(tile, kb * 32, kb * 32 + 16, dt * 16, li, lg);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[rb][dt] = mma(f, b[rb], acc[rb][dt]);
            }
        }
    }
    __device__ static inline void half(f32x4_t (&acc)[RB][DH / 16], const bf16_t* tile, const f32x4_t (&w)[RB][2], int kb, int li, int lg) {
        bf16x8_t b[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) b[rb] = pack8(w[rb][0], w[rb][1]);
#pragma unroll
        for (int dt = 0; dt < DH / 16; ++dt) {
            const bf16x8_t f = frag_tr<LD>(tile, kb * 32, kb * 32 + 16, dt * 16, li, lg);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb][dt] = mma(f, b[rb], acc[rb][dt]);
        }
    }
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
// RB: 16-row query blocks per wave.  RB = 2 (bf16): a workgroup owns 128 query rows, every K / V fragment read from LDS feeds two
// MFMAs (the kernel is bound by latency and by the LDS reads per MFMA, not by the matrix pipe: 24 MFMAs against 36 fragment reads per
// wave and key tile at RB = 1), each K / V tile is fetched and staged once per 128 rows, and the two row blocks are two independent
// dependency chains the scheduler interleaves.  K / V tiles are double-buffered in LDS: tile t + 1 is written to the other buffer
// while tile t is consumed, one barrier per tile; its global loads were issued a whole tile earlier.
// Causal: the query tiles of a head run last-tile-first (the last tile walks the most key tiles; dispatched first, the short tiles
// fill the tail of the launch).
template <int RB> __device__ inline int attn_qtile(const AttnArgs& a, int tile) { return (a.mask_mode & 2) ? (int)gridDim.x - 1 - tile : tile; }

template <typename T, int DH, int RB>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? (RB == 1 ? 3 : 2) : 1) void attn_fwd_kernel(AttnArgs a) {
    constexpr int LD = DH + AT<T>::PAD, NKS = DH / (sizeof(T) == 2 ? 32 : 4), QT = 64 * RB, NB = sizeof(T) == 2 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) T sK[NB][64 * LD];
    __shared__ __attribute__((aligned(16))) T sV[NB][64 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    int tile_, z;
    xcd_block(tile_, z);
    tile_ = attn_qtile<RB>(a, tile_);
    const int b = z / a.H, h = z - b * a.H;
    const int qb0 = tile_ * QT, qw0 = qb0 + wave * 16 * RB;          // first query row of the workgroup / of this wave
    const T* Q = reinterpret_cast<const T*>(a.q) + (long)b * a.Lq * a.ldq + h * DH;
    const T* K = reinterpret_cast<const T*>(a.k) + (long)b * a.Lk * a.ldk + h * DH;
    const T* V = reinterpret_cast<const T*>(a.v) + (long)b * a.Lk * a.ldv + h * DH;
    int q[RB], qc[RB];
    typename AT<T>::frag qf[RB][NKS];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        q[rb] = qw0 + rb * 16 + li; qc[rb] = min(q[rb], a.Lq - 1);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[rb][ks] = frag_global<T>(Q, a.ldq, qc[rb], ks, lg);
    }
    int kend = a.Lk;
    if (a.mask_mode & 1) kend = min(kend, a.klen[b]);
    int ktiles = (kend + 63) / 64;
    if (a.mask_mode & 2) ktiles = min(ktiles, (min(qb0 + QT - 1, a.Lq - 1)) / 64 + 1);
    f32x4_t o[RB][DH / 16];
    float m[RB], l[RB], g[RB];                       // m: reference exponent (log2 domain), l: this lane's part of the row sum
    uint32_t drow[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
        for (int dt = 0; dt < DH / 16; ++dt) o[rb][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        m[rb] = -INFINITY; l[rb] = 0.f; g[rb] = 0.f;
        drow[rb] = (uint32_t)(((long)z * a.Lq + qc[rb]) * a.Lk);
    }
    const bool ga = a.ga_rows != nullptr;
    float ga_iq = 0.f, ga_ik = 0.f;
    if (ga) { ga_iq = 1.f / (float)max(min(a.qlen[b], a.Lq), 1); ga_ik = 1.f / (float)max(kend, 1); }
    const float sl2 = a.scale * B2S_LOG2E;
    TileRegs<T, DH> rk, rv;
    tile_fetch<T, DH>(rk, K, a.ldk, 0, a.Lk, tid); tile_fetch<T, DH>(rv, V, a.ldv, 0, a.Lk, tid);
    wait_all_loads();
    tile_store<T, DH>(sK[0], rk, tid); tile_store<T, DH>(sV[0], rv, tid);
    tile_fetch<T, DH>(rk, K, a.ldk, 64, a.Lk, tid); tile_fetch<T, DH>(rv, V, a.ldv, 64, a.Lk, tid);
    __syncthreads();
    for (int kt = 0; kt < ktiles; ++kt) {
        const int k0 = kt * 64;
        const T* cK = sK[NB == 2 ? (kt & 1) : 0];
        const T* cV = sV[NB == 2 ? (kt & 1) : 0];
        if (NB == 2) {
            // tile kt + 1 (in registers since the previous iteration) -> the other buffer, whose last readers passed the barrier below;
            // then the loads of tile kt + 2 (unconditional, see tile_fetch: past the last tile they re-read the last rows)
            tile_store<T, DH>(sK[(kt + 1) & 1], rk, tid); tile_store<T, DH>(sV[(kt + 1) & 1], rv, tid);
            __builtin_amdgcn_sched_barrier(0);       // (the loads below re-use the registers just stored: hoisted above the stores they need 24 more)
            tile_fetch<T, DH>(rk, K, a.ldk, k0 + 128, a.Lk, tid); tile_fetch<T, DH>(rv, V, a.ldv, k0 + 128, a.Lk, tid);
            __builtin_amdgcn_sched_barrier(0);
        }
        // causal: a wave whose rows all precede this key tile has nothing to add (wave-uniform)
        const bool skip = (a.mask_mode & 2) && k0 > qw0 + 16 * RB - 1;
        if (!skip) {
        f32x4_t s[RB][4];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int t = 0; t < 4; ++t) s[rb][t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const typename AT<T>::frag kf = frag_row<LD>(cK, t * 16, ks, li, lg);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) s[rb][t] = mma(kf, qf[rb][ks], s[rb][t]);
            }
        // every key of the tile visible to every row of this wave?  (wave-uniform; the common case skips all mask math)
        const bool interior = k0 + 64 <= kend && (!(a.mask_mode & 2) || k0 + 63 <= qw0);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            float mx = -INFINITY;
            if (interior) {
#pragma unroll
                for (int t = 0; t < 4; ++t) mx = fmaxf(fmaxf(mx, fmaxf(s[rb][t][0], s[rb][t][1])), fmaxf(s[rb][t][2], s[rb][t][3]));
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = k0 + t * 16 + lg * 4 + r;
                        const bool ok = key < kend && (!(a.mask_mode & 2) || key <= q[rb]);
                        s[rb][t][r] = ok ? s[rb][t][r] : -INFINITY;
                        mx = fmaxf(mx, s[rb][t][r]);
                    }
            }
            mx = group_max(mx) * sl2;
            const bool grow = mx > m[rb] + B2S_LAZY;         // also true for the first finite maximum (m = -inf)
            if (__any(grow)) {
                const float mn = grow ? mx : m[rb];
                const float alpha = mn == m[rb] ? 1.f : fast_exp2(m[rb] - mn);      // m = -inf -> 0
                m[rb] = mn; l[rb] *= alpha; g[rb] *= alpha;
#pragma unroll
                for (int dt = 0; dt < DH / 16; ++dt) { o[rb][dt][0] *= alpha; o[rb][dt][1] *= alpha; o[rb][dt][2] *= alpha; o[rb][dt][3] *= alpha; }
            }
            const float mref = m[rb] == -INFINITY ? 0.f : m[rb];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = fast_exp2(fmaf(s[rb][t][r], sl2, -mref));   // masked: 2^-inf = 0
                    l[rb] += p;
                    s[rb][t][r] = p;
                }
            if (ga) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) g[rb] += s[rb][t][r] * ga_w(q[rb], k0 + t * 16 + lg * 4 + r, ga_iq, ga_ik, a.ga_inv2s2);
            }
            if (a.drop.thresh) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        s[rb][t][r] = b2s_keep(a.drop, drow[rb] + (uint32_t)(k0 + t * 16 + lg * 4 + r)) ? s[rb][t][r] * a.drop.scale : 0.f;
            }
        }
        SPR<T, DH, LD, RB>::run(o, cV, s, li, lg);
        }
        if (NB == 2) __syncthreads();              // tile kt + 1 is in LDS, every wave is done with tile kt
        else if (kt + 1 < ktiles) {                // fp32 (parity mode): one buffer -- re-stage in place
            __syncthreads();
            tile_fetch<T, DH>(rk, K, a.ldk, k0 + 64, a.Lk, tid); tile_fetch<T, DH>(rv, V, a.ldv, k0 + 64, a.Lk, tid);
            tile_store<T, DH>(sK[0], rk, tid); tile_store<T, DH>(sV[0], rv, tid);
            __syncthreads();
        }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const float lt = group_sum(l[rb]);
        const float gt = ga ? group_sum(g[rb]) : 0.f;
        if (q[rb] < a.Lq) {
            const float inv = 1.f / lt;
            T* out = reinterpret_cast<T*>(a.out) + ((long)b * a.Lq + q[rb]) * a.ldo + h * DH;
            store_rows<T, DH>(out, o[rb], inv, lg);
            if (lg == 0 && a.lse) a.lse[(long)z * a.Lq + q[rb]] = (m[rb] + __log2f(lt)) * B2S_LN2;
            if (lg == 0 && ga) a.ga_rows[(long)z * a.Lq + q[rb]] = q[rb] < a.qlen[b] ? gt * inv : 0.f;
        }
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

// dQ: per workgroup 64 * RB query rows; loops over key tiles (same structure as the forward kernel)
template <typename T, int DH, int RB>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 2 : 1) void attn_bwd_dq_kernel(AttnArgs a) {
    constexpr int LD = DH + AT<T>::PAD, NKS = DH / (sizeof(T) == 2 ? 32 : 4), QT = 64 * RB, NB = sizeof(T) == 2 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) T sK[NB][64 * LD];
    __shared__ __attribute__((aligned(16))) T sV[NB][64 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    int tile_, z;
    xcd_block(tile_, z);
    tile_ = attn_qtile<RB>(a, tile_);
    const int b = z / a.H, h = z - b * a.H;
    const int qb0 = tile_ * QT, qw0 = qb0 + wave * 16 * RB;
    const T* Q = reinterpret_cast<const T*>(a.q) + (long)b * a.Lq * a.ldq + h * DH;
    const T* K = reinterpret_cast<const T*>(a.k) + (long)b * a.Lk * a.ldk + h * DH;
    const T* V = reinterpret_cast<const T*>(a.v) + (long)b * a.Lk * a.ldv + h * DH;
    const T* dO = reinterpret_cast<const T*>(a.dout) + (long)b * a.Lq * a.ldo + h * DH;
    const T* O = reinterpret_cast<const T*>(a.oref) + (long)b * a.Lq * a.ldo + h * DH;
    int kend = a.Lk;
    if (a.mask_mode & 1) kend = min(kend, a.klen[b]);
    // guided attention: dP += c W, D += c * rowsum(P W) on valid query rows
    float ga_iq = 0.f, ga_ik = 0.f;
    int ql = 0;
    const bool ga_on = a.ga_rows != nullptr;             // (kernel argument: scalar branch)
    if (a.ga_rows) {
        ql = min(a.qlen[b], a.Lq);
        ga_iq = 1.f / (float)max(ql, 1); ga_ik = 1.f / (float)max(kend, 1);
    }
    int q[RB], qc[RB];
    typename AT<T>::frag qf[RB][NKS], dof[RB][NKS];
    float lse2[RB], Dq[RB], gc[RB];
    uint32_t drow[RB];
    f32x4_t dq[RB][DH / 16];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        q[rb] = qw0 + rb * 16 + li; qc[rb] = min(q[rb], a.Lq - 1);
        // D[q] = sum_d dO[q][d] * O[q][d], from the same fragments (each lane holds 1/4 of the row; two shuffles finish it)
        float d = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            qf[rb][ks] = frag_global<T>(Q, a.ldq, qc[rb], ks, lg); dof[rb][ks] = frag_global<T>(dO, a.ldo, qc[rb], ks, lg);
            d += frag_dot(frag_global<T>(O, a.ldo, qc[rb], ks, lg), dof[rb][ks]);
        }
        d = group_sum(d);
        gc[rb] = 0.f;
        if (a.ga_rows) {
            if (q[rb] < ql) gc[rb] = *a.ga_scale;
            d += gc[rb] * a.ga_rows[(long)z * a.Lq + qc[rb]];
        }
        Dq[rb] = d;
        if (lg == 0 && q[rb] < a.Lq) a.dsum[(long)z * a.Lq + q[rb]] = d;              // the dK/dV kernel reads it
        lse2[rb] = a.lse[(long)z * a.Lq + qc[rb]] * B2S_LOG2E;
        drow[rb] = (uint32_t)(((long)z * a.Lq + qc[rb]) * a.Lk);
#pragma unroll
        for (int dt = 0; dt < DH / 16; ++dt) dq[rb][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    int ktiles = (kend + 63) / 64;
    if (a.mask_mode & 2) ktiles = min(ktiles, (min(qb0 + QT - 1, a.Lq - 1)) / 64 + 1);
    const float sl2 = a.scale * B2S_LOG2E;
    TileRegs<T, DH> rk, rv;
    tile_fetch<T, DH>(rk, K, a.ldk, 0, a.Lk, tid); tile_fetch<T, DH>(rv, V, a.ldv, 0, a.Lk, tid);
    wait_all_loads();
    tile_store<T, DH>(sK[0], rk, tid); tile_store<T, DH>(sV[0], rv, tid);
    tile_fetch<T, DH>(rk, K, a.ldk, 64, a.Lk, tid); tile_fetch<T, DH>(rv, V, a.ldv, 64, a.Lk, tid);
    __syncthreads();
    for (int kt = 0; kt < ktiles; ++kt) {
        const int k0 = kt * 64;
        const T* cK = sK[NB == 2 ? (kt & 1) : 0];
        const T* cV = sV[NB == 2 ? (kt & 1) : 0];
        if (NB == 2) {
            tile_store<T, DH>(sK[(kt + 1) & 1], rk, tid); tile_store<T, DH>(sV[(kt + 1) & 1], rv, tid);
            __builtin_amdgcn_sched_barrier(0);
            tile_fetch<T, DH>(rk, K, a.ldk, k0 + 128, a.Lk, tid); tile_fetch<T, DH>(rv, V, a.ldv, k0 + 128, a.Lk, tid);
            __builtin_amdgcn_sched_barrier(0);
        }
        const bool skip = (a.mask_mode & 2) && k0 > qw0 + 16 * RB - 1;
        if (!skip) {
        // the tile's 64 keys in two halves of 32 (one bf16 MFMA k step of the second product each): S, dP of a half -> dS -> dQ += dS K,
        // so that only half of the logits / dP are live at a time (both row blocks of a 96-wide head would not fit the register file)
        const bool interior = k0 + 64 <= kend && (!(a.mask_mode & 2) || k0 + 63 <= qw0);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x4_t s[RB][2], dp[RB][2];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int u = 0; u < 2; ++u) { s[rb][u] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dp[rb][u] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const typename AT<T>::frag kf = frag_row<LD>(cK, (2 * kb + u) * 16, ks, li, lg);
                    const typename AT<T>::frag vf = frag_row<LD>(cV, (2 * kb + u) * 16, ks, li, lg);
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) { s[rb][u] = mma(kf, qf[rb][ks], s[rb][u]); dp[rb][u] = mma(vf, dof[rb][ks], dp[rb][u]); }
                }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = k0 + (2 * kb + u) * 16 + lg * 4 + r;
                        float p = fast_exp2(fmaf(s[rb][u][r], sl2, -lse2[rb]));
                        if (!interior) p = (key < kend && (!(a.mask_mode & 2) || key <= q[rb])) ? p : 0.f;
                        float d = dp[rb][u][r];
                        if (a.drop.thresh) d = b2s_keep(a.drop, drow[rb] + (uint32_t)key) ? d * a.drop.scale : 0.f;
                        if (ga_on) d += gc[rb] * ga_w(q[rb], key, ga_iq, ga_ik, a.ga_inv2s2);      // (gc = 0 on padded rows)
                        s[rb][u][r] = p * (d - Dq[rb]) * a.scale;
                    }
            }
            SPR<T, DH, LD, RB>::half(dq, cK, s, kb, li, lg);
        }
        }
        if (NB == 2) __syncthreads();
        else if (kt + 1 < ktiles) {
            __syncthreads();
            tile_fetch<T, DH>(rk, K, a.ldk, k0 + 64, a.Lk, tid); tile_fetch<T, DH>(rv, V, a.ldv, k0 + 64, a.Lk, tid);
            tile_store<T, DH>(sK[0], rk, tid); tile_store<T, DH>(sV[0], rv, tid);
            __syncthreads();
        }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
        if (q[rb] < a.Lq) store_rows<T, DH>(reinterpret_cast<T*>(a.dq) + ((long)b * a.Lq + q[rb]) * a.lddq + h * DH, dq[rb], 1.f, lg);
}

// dK, dV: per workgroup 64 * KBW keys (a wave owns KBW blocks of 16); loops over query tiles of 64, each consumed in two halves of
// 32 queries (S, dP of a half -> Pd, dS -> dV += dO^T Pd, dK += Q^T dS: only half a tile of logits is live, and the Q / dO fragments
// read from LDS feed KBW MFMAs each).  Q / dO tiles are double-buffered in LDS like the K / V tiles of the forward kernel.
template <typename T, int DH, int KBW>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 2 : 1) void attn_bwd_dkv_kernel(AttnArgs a) {
    constexpr int LD = DH + AT<T>::PAD, NKS = DH / (sizeof(T) == 2 ? 32 : 4), KT = 64 * KBW, NB = sizeof(T) == 2 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) T sQ[NB][64 * LD];
    __shared__ __attribute__((aligned(16))) T sO[NB][64 * LD];
    __shared__ __attribute__((aligned(16))) float sL[NB][64], sD[NB][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    int tile_, z;
    xcd_block(tile_, z);
    const int b = z / a.H, h = z - b * a.H;
    const int kb0 = tile_ * KT, kw0 = kb0 + wave * 16 * KBW;          // first key of the workgroup / of this wave
    const T* Q = reinterpret_cast<const T*>(a.q) + (long)b * a.Lq * a.ldq + h * DH;
    const T* K = reinterpret_cast<const T*>(a.k) + (long)b * a.Lk * a.ldk + h * DH;
    const T* V = reinterpret_cast<const T*>(a.v) + (long)b * a.Lk * a.ldv + h * DH;
    const T* dO = reinterpret_cast<const T*>(a.dout) + (long)b * a.Lq * a.ldo + h * DH;
    int kend = a.Lk;
    if (a.mask_mode & 1) kend = min(kend, a.klen[b]);
    int key[KBW], kc[KBW];
    typename AT<T>::frag kf[KBW][NKS], vf[KBW][NKS];
    f32x4_t dk[KBW][DH / 16], dv[KBW][DH / 16];
#pragma unroll
    for (int kb = 0; kb < KBW; ++kb) {
        key[kb] = kw0 + kb * 16 + li; kc[kb] = min(key[kb], a.Lk - 1);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) { kf[kb][ks] = frag_global<T>(K, a.ldk, kc[kb], ks, lg); vf[kb][ks] = frag_global<T>(V, a.ldv, kc[kb], ks, lg); }
#pragma unroll
        for (int dt = 0; dt < DH / 16; ++dt) { dk[kb][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dv[kb][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
    }
    float gc = 0.f, ga_iq = 0.f, ga_ik = 0.f;
    int ga_ql = 0;
    const bool ga_on = a.ga_rows != nullptr;
    if (a.ga_rows) {
        ga_ql = min(a.qlen[b], a.Lq);
        gc = *a.ga_scale;
        ga_iq = 1.f / (float)max(ga_ql, 1); ga_ik = 1.f / (float)max(kend, 1);
    }
    const int qtiles = (a.Lq + 63) / 64;
    const int qt0 = (a.mask_mode & 2) ? kb0 / 64 : 0;          // causal: queries before this key tile never see it
    TileRegs<T, DH> rq, ro;
    float r_l = 0.f, r_d = 0.f;
    const float sl2 = a.scale * B2S_LOG2E;
    const uint32_t zq = (uint32_t)z * (uint32_t)a.Lq;
    auto fetch = [&](int qt) {              // (unconditional loads: see tile_fetch; every wave loads the tile's 64 row statistics, wave 0 stages them)
        tile_fetch<T, DH>(rq, Q, a.ldq, qt * 64, a.Lq, tid); tile_fetch<T, DH>(ro, dO, a.ldo, qt * 64, a.Lq, tid);
        const int qq = min(qt * 64 + lane, a.Lq - 1);
        r_l = a.lse[(long)z * a.Lq + qq]; r_d = a.dsum[(long)z * a.Lq + qq];
    };
    auto stage = [&](int buf) {
        tile_store<T, DH>(sQ[buf], rq, tid); tile_store<T, DH>(sO[buf], ro, tid);
        if (tid < 64) { sL[buf][tid] = r_l * B2S_LOG2E; sD[buf][tid] = r_d; }
    };
    fetch(qt0);
    wait_all_loads();
    stage(0);
    fetch(qt0 + 1);
    __syncthreads();
    for (int qt = qt0; qt < qtiles; ++qt) {
        const int q0 = qt * 64, cur = NB == 2 ? ((qt - qt0) & 1) : 0;
        const T* cQ = sQ[cur];
        const T* cO = sO[cur];
        if (NB == 2) { stage(cur ^ 1); __builtin_amdgcn_sched_barrier(0); fetch(qt + 2); __builtin_amdgcn_sched_barrier(0); }
        // causal: a wave whose keys all come after this query tile has nothing to add (wave-uniform)
        const bool skip = (a.mask_mode & 2) && kw0 > q0 + 63;
        if (!skip) {
        // all keys of this wave valid and visible to all 64 queries of the tile?  (wave-uniform)
        const bool interior = kw0 + 16 * KBW <= kend && q0 + 64 <= a.Lq && (!(a.mask_mode & 2) || kw0 + 16 * KBW - 1 <= q0);
#pragma unroll 1
        for (int hq = 0; hq < 2; ++hq) {
            f32x4_t s[KBW][2], dp[KBW][2], pd[KBW][2];
#pragma unroll
            for (int kb = 0; kb < KBW; ++kb)
#pragma unroll
                for (int u = 0; u < 2; ++u) { s[kb][u] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dp[kb][u] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const typename AT<T>::frag fq = frag_row<LD>(cQ, (2 * hq + u) * 16, ks, li, lg);      // rows = queries
                    const typename AT<T>::frag fo = frag_row<LD>(cO, (2 * hq + u) * 16, ks, li, lg);
#pragma unroll
                    for (int kb = 0; kb < KBW; ++kb) { s[kb][u] = mma(fq, kf[kb][ks], s[kb][u]); dp[kb][u] = mma(fo, vf[kb][ks], dp[kb][u]); }
                }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f32x4_t lq = *reinterpret_cast<const f32x4_t*>(sL[cur] + (2 * hq + u) * 16 + lg * 4);
                const f32x4_t dsum4 = *reinterpret_cast<const f32x4_t*>(sD[cur] + (2 * hq + u) * 16 + lg * 4);
#pragma unroll
                for (int kb = 0; kb < KBW; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int qq = q0 + (2 * hq + u) * 16 + lg * 4 + r;
                        float p = fast_exp2(fmaf(s[kb][u][r], sl2, -lq[r]));
                        if (!interior) p = (key[kb] < kend && qq < a.Lq && (!(a.mask_mode & 2) || key[kb] <= qq)) ? p : 0.f;
                        float d = dp[kb][u][r], pdr = p;
                        if (a.drop.thresh) {
                            const bool keep = b2s_keep(a.drop, (zq + (uint32_t)qq) * (uint32_t)a.Lk + (uint32_t)kc[kb]);
                            d = keep ? d * a.drop.scale : 0.f;
                            pdr = keep ? p * a.drop.scale : 0.f;
                        }
                        if (ga_on) d += qq < ga_ql ? gc * ga_w(qq, key[kb], ga_iq, ga_ik, a.ga_inv2s2) : 0.f;
                        pd[kb][u][r] = pdr;
                        s[kb][u][r] = p * (d - dsum4[r]) * a.scale;
                    }
            }
            SPR<T, DH, LD, KBW>::half(dv, cO, pd, hq, li, lg);        // dV^T[d][key] += sum_q dO[q][d] * Pd[q][key]
            SPR<T, DH, LD, KBW>::half(dk, cQ, s, hq, li, lg);         // dK^T[d][key] += sum_q Q[q][d]  * dS[q][key]
        }
        }
        if (NB == 2) __syncthreads();
        else if (qt + 1 < qtiles) {
            __syncthreads();
            fetch(qt + 1); stage(0);
            __syncthreads();
        }
    }
#pragma unroll
    for (int kb = 0; kb < KBW; ++kb)
        if (key[kb] < a.Lk) {
            store_rows<T, DH>(reinterpret_cast<T*>(a.dk) + ((long)b * a.Lk + key[kb]) * a.lddk + h * DH, dk[kb], 1.f, lg);
            store_rows<T, DH>(reinterpret_cast<T*>(a.dv) + ((long)b * a.Lk + key[kb]) * a.lddv + h * DH, dv[kb], 1.f, lg);
        }
}

// alignment rows on demand: align[z][k][q] = softmax weight, recomputed from q, k and the saved log-sum-exp
template <typename T>
__global__ __launch_bounds__(256) void attn_align_kernel(AttnArgs a, float* align, int dh) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);           // r = z*Lq + q
    if (r >= (long)a.B * a.H * a.Lq) return;
    const int q = (int)(r % a.Lq), z = (int)(r / a.Lq), b = z / a.H, h = z - b * a.H;
    const T* Q = reinterpret_cast<const T*>(a.q) + ((long)b * a.Lq + q) * a.ldq + h * dh;
    const T* K = reinterpret_cast<const T*>(a.k) + (long)b * a.Lk * a.ldk + h * dh;
    int kend = a.Lk;
    if (a.mask_mode & 1) kend = min(kend, a.klen[b]);
    if (a.mask_mode & 2) kend = min(kend, q + 1);
    const float lse = a.lse[r];
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

// query row blocks per wave of the forward / dQ kernels: 2 in bf16 (128-row workgroups), 1 in the fp32 parity mode
// (B2S_ATTN_RB=1: A/B switch back to 64-row workgroups)
template <typename T, int DH, int RB>
int launch_rb(const AttnArgs& a, int which, hipStream_t st) {
    dim3 grid(cdiv(a.Lq, 64 * RB), a.B * a.H);
    if (which == 0) hipLaunchKernelGGL((attn_fwd_kernel<T, DH, RB>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((attn_bwd_dq_kernel<T, DH, RB>), grid, dim3(256), 0, st, a);
    B2S_LAUNCH_CHECK();
    return 0;
}
template <typename T, int DH>
int launch_dh(const AttnArgs& a, int which, hipStream_t st) {
    if (which < 2) {
        if constexpr (sizeof(T) == 2) {
            static const int rb_all = getenv("B2S_ATTN_RB") ? atoi(getenv("B2S_ATTN_RB")) : 2;
            static const int rb_f = getenv("B2S_ATTN_RB_FWD") ? atoi(getenv("B2S_ATTN_RB_FWD")) : rb_all;
            static const int rb_q = getenv("B2S_ATTN_RB_DQ") ? atoi(getenv("B2S_ATTN_RB_DQ")) : rb_all;
            if ((which == 0 ? rb_f : rb_q) == 2) return launch_rb<T, DH, 2>(a, which, st);
        }
        return launch_rb<T, DH, 1>(a, which, st);
    }
    // 128-key workgroups when they still give the chip at least one workgroup per CU (decoder self-attention: 5 x 112); the 114-key
    // memory of the encoder-decoder attention keeps 64-key workgroups (2 x 112 instead of 1 x 112)
    if constexpr (sizeof(T) == 2) {
        static const int kbw = getenv("B2S_ATTN_KBW") ? atoi(getenv("B2S_ATTN_KBW")) : 2;
        if (kbw == 2 && (long)cdiv(a.Lk, 128) * a.B * a.H >= 256) {
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, DH, 2>), dim3(cdiv(a.Lk, 128), a.B * a.H), dim3(256), 0, st, a);
            B2S_LAUNCH_CHECK();
            return 0;
        }
    }
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, DH, 1>), dim3(cdiv(a.Lk, 64), a.B * a.H), dim3(256), 0, st, a);
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

int b2s_flash_fwd(int dtype, const AttnArgs& a, int dh, hipStream_t st) {
    B2S_TRY(check(a, dtype, dh));
    B2S_CHECK(a.out, "attention: null output");
    return dtype ? launch_t<bf16_t>(a, dh, 0, st) : launch_t<float>(a, dh, 0, st);
}
int b2s_flash_bwd(int dtype, const AttnArgs& a_in, int dh, const void* O, hipStream_t st) {
    B2S_TRY(check(a_in, dtype, dh));
    B2S_CHECK(a_in.dout && a_in.dq && a_in.dk && a_in.dv && a_in.lse && a_in.dsum && O, "attention backward: null argument");
    B2S_CHECK(!a_in.ga_rows || a_in.ga_scale, "attention backward: the guided-attention term needs its scale");
    AttnArgs a = a_in;
    a.oref = O;
    B2S_TRY(dtype ? launch_t<bf16_t>(a, dh, 1, st) : launch_t<float>(a, dh, 1, st));
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
