// bf16 training attention for gfx950 on the 32x32x16 MFMA (reference: transformer/attention.py:72-92 dot_product_attention and its autograd
// backward).  Same contract as the kernels of attention.hip (AttnArgs; the fp32 parity mode and head sizes this file does not cover stay there).
//
// Structure (forward, dQ, dK/dV alike): a workgroup of NW waves owns NW x 32 rows of one (batch, head) -- query rows in the forward and dQ kernels,
// key rows in the dK/dV kernel -- and streams 64-row tiles of the other sequence through a two-deep LDS ring (register-staged: the global loads of
// tile t + 2 are in flight while tile t is consumed; ONE barrier per tile).  Every product is computed "swapped" so that a lane owns ONE row of the
// wave's 32 and 16 (of 32) columns of the streamed tile in its accumulator registers (32x32 C layout: col = lane & 31, row = (r & 3) + 8 (r >> 2)
// + 4 (lane >> 5)):
//      S^T = K Q^T, dP^T = V dO^T             A = 32 streamed rows x 16 features, one ds_read_b128 per MFMA; B = the wave's own rows, in registers
//      O^T += V^T P^T, dQ^T += K^T dS^T, ..   A = 32 features x 16 streamed rows, two ds_read_b64_tr_b16 per MFMA; B = the packed accumulator of the
//                                             first product AS IT LIES: k-slot j of lane half hi <-> streamed row 16 s + 4 hi + (j & 3) + 8 (j >> 2),
//                                             the transposed reads fetch exactly those rows, so no lane exchange is needed
// so the softmax is in registers (a row's other half of the keys lives in lane ^ 32: one v_permlane32_swap per reduction), the dropout word of a
// key PAIR is one hash in the forward / dQ kernels (b2s_common.h: b2s_wword), and 32 rows per wave halve the LDS bytes per FLOP of the 16-row
// kernels.  LDS tiles are unpadded [64][dh] images with the 16-byte chunk index XORed by row bits (conflict-free for the b128 reads of one tile row
// per lane AND for the transposed reads of 4 rows x 64 bytes per 32 lanes).
#include "b2s_common.h"
#include "attention.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((address_space(3))) bf16x4_t* lds_b64_ptr;

#ifndef A32_LAB
#define A32_LAB 0            // lab builds: bit 0 = no softmax (p = logits), bit 1 = dropout compiled out
#endif

__device__ inline f32x16_t mma32(bf16x8_t a, bf16x8_t b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
// combine a per-lane partial with the one lane ^ 32 holds (the other 16 columns of the same row); identical result in both lanes
__device__ inline float half_max(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ inline float half_sum(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ inline float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
constexpr float A32_LOG2E = 1.4426950408889634f, A32_LN2 = 0.6931471805599453f;
constexpr float A32_LAZY = 8.f;                       // running-maximum slack, as attention.hip: B2S_LAZY

// chunk swizzle of a tile row: 16 rows of one b128 lane group land on 16 distinct 16-byte slots, 4 consecutive rows x 4 chunks of a transposed read too
template <int DH> __device__ inline int swz(int row) { return DH == 64 ? ((((row >> 1) & 1) << 2) | ((row >> 2) & 3)) : ((row >> 2) & 3); }

// ---- global -> registers -> LDS staging of a [64][DH] tile by NT threads
template <int DH, int NT> struct Stage { uint4 v[(64 * (DH / 8)) / NT]; };
template <int DH, int NT>
__device__ inline void stage_fetch(Stage<DH, NT>& r, const bf16_t* src, long ld, int row0, int nrows, int tid) {
    constexpr int CPR = DH / 8, NV = 64 * CPR / NT;
    static_assert(64 * CPR % NT == 0, "tile must split evenly over the workgroup");
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * NT, rr = v / CPR, c = v - rr * CPR;
        r.v[i] = make_uint4(0, 0, 0, 0);
        if (row0 + rr < nrows) r.v[i] = *reinterpret_cast<const uint4*>(src + (long)(row0 + rr) * ld + c * 8);
    }
}
template <int DH, int NT>
__device__ inline void stage_store(bf16_t* lds, const Stage<DH, NT>& r, int tid) {
    constexpr int CPR = DH / 8, NV = 64 * CPR / NT;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * NT, rr = v / CPR, c = v - rr * CPR;
        *reinterpret_cast<uint4*>(lds + rr * DH + ((c ^ swz<DH>(rr)) * 8)) = r.v[i];
    }
}
// A operand of a "first" product: tile row `row`, features [16 ks + 8 hi, +8)
template <int DH> __device__ inline bf16x8_t frag_a(const bf16_t* tile, int row, int ks, int hi) {
    return *reinterpret_cast<const bf16x8_t*>(tile + row * DH + (((ks * 2 + hi) ^ swz<DH>(row)) * 8));
}
// A operand of a "second" product: features [d0, d0 + 32) x tile rows {rb + 4 hi + j, rb + 8 + 4 hi + j : j < 4}; lane = (g = lane >> 4, li = lane & 15)
template <int DH> __device__ inline bf16x8_t frag_t(const bf16_t* tile, int rb, int d0, int lane) {
    const int li = lane & 15, g = lane >> 4;
    const int row = rb + (g >> 1) * 4 + (li >> 2), col = d0 + (g & 1) * 16 + (li & 3) * 4;
    const bf16_t* p0 = tile + row * DH + (((col >> 3) ^ swz<DH>(row)) * 8) + (col & 7);
    const bf16_t* p1 = tile + (row + 8) * DH + (((col >> 3) ^ swz<DH>(row + 8)) * 8) + (col & 7);
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64_ptr)p0);
    const bf16x4_t hh = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64_ptr)p1);
    bf16x8_t r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hh[0]; r[5] = hh[1]; r[6] = hh[2]; r[7] = hh[3];
    return r;
}
// B operand of a "second" product: accumulator registers [8 s, 8 s + 8) of a first product, rounded to bf16
template <int S8> __device__ inline bf16x8_t pack8(const f32x16_t& v) {
    const u32x4_t u = {f2bf2(v[S8 * 8 + 0], v[S8 * 8 + 1]), f2bf2(v[S8 * 8 + 2], v[S8 * 8 + 3]), f2bf2(v[S8 * 8 + 4], v[S8 * 8 + 5]),
                       f2bf2(v[S8 * 8 + 6], v[S8 * 8 + 7])};
    return __builtin_bit_cast(bf16x8_t, u);
}
// the wave's own row as B operand, straight from global memory: features [16 ks + 8 hi, +8) of row `row`
__device__ inline bf16x8_t frag_own(const bf16_t* base, long ld, int row, int ks, int hi) {
    return *reinterpret_cast<const bf16x8_t*>(base + (long)row * ld + ks * 16 + hi * 8);
}
__device__ inline float frag_dot(bf16x8_t x, bf16x8_t y) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += bf2f((bf16_t)x[e]) * bf2f((bf16_t)y[e]);
    return s;
}
// accumulator row index of register r: (r & 3) + 8 (r >> 2) + 4 hi
__device__ inline int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// XCD-aware (block, head) assignment, as attention.hip: xcd_block
__device__ inline void a32_block(int& blk, int& z) {
    const int nx = gridDim.x, nwg = nx * gridDim.y, orig = blockIdx.y * nx + blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    z = wg / nx; blk = wg - z * nx;
}
__device__ inline float ga_w(int q, int k, float iq, float ik, float inv2s2) {
    const float d = (float)k * ik - (float)q * iq;
    return 1.f - __expf(-d * d * inv2s2);
}
// store a transposed [feature][own row] accumulator set as the own row's DH features (4 consecutive features per store)
template <int DH>
__device__ inline void store_own(bf16_t* dst, const f32x16_t (&acc)[DH / 32], float mul, int hi) {
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            uint2 u;
            u.x = f2bf2(acc[dt][g4 * 4 + 0] * mul, acc[dt][g4 * 4 + 1] * mul);
            u.y = f2bf2(acc[dt][g4 * 4 + 2] * mul, acc[dt][g4 * 4 + 3] * mul);
            *reinterpret_cast<uint2*>(dst + dt * 32 + g4 * 8 + hi * 4) = u;
        }
}
template <int DH> __device__ inline void zero_own(bf16_t* dst, int hi) {
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) *reinterpret_cast<uint2*>(dst + dt * 32 + g4 * 8 + hi * 4) = make_uint2(0, 0);
}
__device__ inline uint32_t hash_body(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
constexpr uint32_t GOLD = 0x9E3779B1u;

// ================================================================================================ forward
template <int DH, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn32_fwd_kernel(AttnArgs a) {
    constexpr int NKS = DH / 16, NDT = DH / 32, TILE = 64 * DH, NT = NW * 64, QB = NW * 32;
    __shared__ __attribute__((aligned(16))) bf16_t sK[2 * TILE];
    __shared__ __attribute__((aligned(16))) bf16_t sV[2 * TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    int blk, z;
    a32_block(blk, z);
    blk = gridDim.x - 1 - blk;                                      // causal: the blocks with the most key tiles start first
    const int b = z / a.H, h = z - b * a.H;
    const int qb0 = blk * QB, qw0 = qb0 + wave * 32, q = qw0 + l31, qc = min(q, a.Lq - 1);
    // padded query rows (AttnArgs::qskip): 64-row tiles wholly at or beyond qskip[b] are not computed, their rows are written as zeros
    const int qlive = a.qskip ? min(a.qskip[b], a.Lq) : a.Lq;
    const bool wlive = (qw0 & ~63) < qlive && qw0 < a.Lq;           // wave-uniform
    bf16_t* out = reinterpret_cast<bf16_t*>(a.out) + ((long)b * a.Lq + qc) * a.ldo + h * DH;
    const bool ga = a.ga_rows != nullptr;
    if (!wlive && q < a.Lq) {
        zero_own<DH>(out, hi);
        if (hi == 0 && a.lse) a.lse[(long)z * a.Lq + q] = 0.f;
        if (hi == 0 && ga) a.ga_rows[(long)z * a.Lq + q] = 0.f;
    }
    if (qb0 >= qlive) return;                                       // nothing live in this workgroup (uniform)
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q) + (long)b * a.Lq * a.ldq + h * DH;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(a.k) + (long)b * a.Lk * a.ldk + h * DH;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(a.v) + (long)b * a.Lk * a.ldv + h * DH;
    bf16x8_t qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = frag_own(Q, a.ldq, qc, ks, hi);

    const bool causal = a.mask_mode & 2;
    int kend = a.Lk;
    if (a.mask_mode & 1) kend = min(kend, a.klen[b]);
    int ktiles = (kend + 63) / 64;
    if (causal) {
        int qlast = min(qb0 + QB - 1, a.Lq - 1);
        if (a.qskip) qlast = min(qlast, ((qlive + 63) & ~63) - 1);
        ktiles = min(ktiles, qlast / 64 + 1);
    }
    const int wkt = !wlive ? 0 : (causal ? min(ktiles, min(qw0 + 31, a.Lq - 1) / 64 + 1) : ktiles);   // key tiles this wave computes

    f32x16_t o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m = -INFINITY, l = 0.f, g = 0.f;
    float ga_iq = 0.f, ga_ik = 0.f;
    if (ga) { ga_iq = 1.f / (float)max(min(a.qlen[b], a.Lq), 1); ga_ik = 1.f / (float)max(kend, 1); }
    const float sl2 = a.scale * A32_LOG2E;
    // dropout word of the key pair kp of this lane's row: hash_body(xrow + kp * GOLD); this lane's pairs of a tile are k0 / 2 + 16 t + 4 g4 + 2 hi + {0, 1}
    const uint32_t hk = (uint32_t)((a.Lk + 1) >> 1);
    const uint32_t xrow = ((uint32_t)((long)z * a.Lq + qc) * hk + (uint32_t)(2 * hi)) * GOLD + a.drop.key;
    const uint32_t t16 = a.drop.thresh & 0xffff0000u;

    Stage<DH, NT> rk, rv;
    if (ktiles > 0) {
        stage_fetch<DH, NT>(rk, K, a.ldk, 0, a.Lk, tid); stage_fetch<DH, NT>(rv, V, a.ldv, 0, a.Lk, tid);
        stage_store<DH, NT>(sK, rk, tid); stage_store<DH, NT>(sV, rv, tid);
        if (ktiles > 1) { stage_fetch<DH, NT>(rk, K, a.ldk, 64, a.Lk, tid); stage_fetch<DH, NT>(rv, V, a.ldv, 64, a.Lk, tid); }
    }
    __syncthreads();
    for (int kt = 0; kt < ktiles; ++kt) {
        const int k0 = kt * 64, cur = kt & 1;
        const bf16_t* tK = sK + cur * TILE;
        const bf16_t* tV = sV + cur * TILE;
        const bool active = kt < wkt;
        f32x16_t s[2];
        if (active) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) s[t] = mma32(frag_a<DH>(tK, t * 32 + l31, ks, hi), qf[ks], s[t]);
            }
        }
        // the next tile goes into the other half of the ring: its last readers finished before the barrier that ended the previous iteration
        if (kt + 1 < ktiles) {
            stage_store<DH, NT>(sK + (cur ^ 1) * TILE, rk, tid); stage_store<DH, NT>(sV + (cur ^ 1) * TILE, rv, tid);
            if (kt + 2 < ktiles) { stage_fetch<DH, NT>(rk, K, a.ldk, k0 + 128, a.Lk, tid); stage_fetch<DH, NT>(rv, V, a.ldv, k0 + 128, a.Lk, tid); }
        }
        if (active) {
            if (!(A32_LAB & 1)) {
                // every key of the tile visible to every row of this wave?  (wave-uniform; the common case skips the mask arithmetic)
                const bool interior = k0 + 64 <= kend && (!causal || k0 + 63 <= qw0);
                float mx = -INFINITY;
                if (interior) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
                } else {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = k0 + t * 32 + crow(r, hi);
                            const bool ok = key < kend && (!causal || key <= q);
                            s[t][r] = ok ? s[t][r] : -INFINITY;
                            mx = fmaxf(mx, s[t][r]);
                        }
                }
                mx = half_max(mx) * sl2;
                const bool grow = mx > m + A32_LAZY;             // also true for the first finite maximum (m = -inf)
                if (__any(grow)) {
                    const float mn = grow ? mx : m;
                    const float alpha = mn == m ? 1.f : fast_exp2(m - mn);      // m = -inf -> 0
                    m = mn; l *= alpha; g *= alpha;
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
                }
                const float mref = m == -INFINITY ? 0.f : m;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float p = fast_exp2(fmaf(s[t][r], sl2, -mref));   // masked: 2^-inf = 0
                        l += p;
                        s[t][r] = p;
                    }
            } else {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { s[t][r] *= sl2; l += s[t][r]; }
                m = 0.f;
            }
            if (ga) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) g += s[t][r] * ga_w(q, k0 + t * 32 + crow(r, hi), ga_iq, ga_ik, a.ga_inv2s2);
            }
            if (!(A32_LAB & 2) && a.drop.thresh) {
                const uint32_t xt = xrow + (uint32_t)(k0 >> 1) * GOLD;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int pr = 0; pr < 8; ++pr) {                 // pair pr: registers 2 pr, 2 pr + 1 = keys k0 + 32 t + 8 (pr >> 1) + 4 hi + 2 (pr & 1) + {0, 1}
                        const uint32_t w = hash_body(xt + (uint32_t)(t * 16 + (pr >> 1) * 4 + (pr & 1)) * GOLD);
                        s[t][2 * pr] = (w << 16) >= t16 ? s[t][2 * pr] : 0.f;
                        s[t][2 * pr + 1] = w >= t16 ? s[t][2 * pr + 1] : 0.f;
                    }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bf16x8_t p0 = pack8<0>(s[t]), p1 = pack8<1>(s[t]);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    o[dt] = mma32(frag_t<DH>(tV, t * 32, dt * 32, lane), p0, o[dt]);
                    o[dt] = mma32(frag_t<DH>(tV, t * 32 + 16, dt * 32, lane), p1, o[dt]);
                }
            }
        }
        __syncthreads();
    }
    l = half_sum(l);
    if (ga) g = half_sum(g);
    if (wlive && q < a.Lq) {
        const float inv = 1.f / l;
        store_own<DH>(out, o, inv * a.drop.scale, hi);
        if (hi == 0 && a.lse) a.lse[(long)z * a.Lq + q] = (m + __log2f(l)) * A32_LN2;
        if (hi == 0 && ga) a.ga_rows[(long)z * a.Lq + q] = q < a.qlen[b] ? g * inv : 0.f;
    }
}

// ================================================================================================ dQ
template <int DH, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn32_dq_kernel(AttnArgs a) {
    constexpr int NKS = DH / 16, NDT = DH / 32, TILE = 64 * DH, NT = NW * 64, QB = NW * 32;
    __shared__ __attribute__((aligned(16))) bf16_t sK[2 * TILE];
    __shared__ __attribute__((aligned(16))) bf16_t sV[2 * TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    int blk, z;
    a32_block(blk, z);
    blk = gridDim.x - 1 - blk;
    const int b = z / a.H, h = z - b * a.H;
    const int qb0 = blk * QB, qw0 = qb0 + wave * 32, q = qw0 + l31, qc = min(q, a.Lq - 1);
    const int qlive = a.qskip ? min(a.qskip[b], a.Lq) : a.Lq;
    const bool wlive = (qw0 & ~63) < qlive && qw0 < a.Lq;
    bf16_t* dqo = reinterpret_cast<bf16_t*>(a.dq) + ((long)b * a.Lq + qc) * a.lddq + h * DH;
    if (!wlive && q < a.Lq) {                                        // padded query rows: d context is zero there, so is dQ
        zero_own<DH>(dqo, hi);
        if (hi == 0) a.dsum[(long)z * a.Lq + q] = 0.f;
    }
    if (qb0 >= qlive) return;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q) + (long)b * a.Lq * a.ldq + h * DH;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(a.k) + (long)b * a.Lk * a.ldk + h * DH;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(a.v) + (long)b * a.Lk * a.ldv + h * DH;
    const bf16_t* dO = reinterpret_cast<const bf16_t*>(a.dout) + (long)b * a.Lq * a.ldo + h * DH;
    const bf16_t* O = reinterpret_cast<const bf16_t*>(a.oref) + (long)b * a.Lq * a.ldo + h * DH;
    bf16x8_t qf[NKS], dof[NKS];
    float Dq = 0.f;                                                  // D[q] = sum_d dO[q][d] O[q][d]: this lane holds half of the row's features
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        qf[ks] = frag_own(Q, a.ldq, qc, ks, hi); dof[ks] = frag_own(dO, a.ldo, qc, ks, hi);
        Dq += frag_dot(frag_own(O, a.ldo, qc, ks, hi), dof[ks]);
    }
    Dq = half_sum(Dq);
    const bool causal = a.mask_mode & 2;
    int kend = a.Lk;
    if (a.mask_mode & 1) kend = min(kend, a.klen[b]);
    float gc = 0.f, ga_iq = 0.f, ga_ik = 0.f;
    if (a.ga_rows) {                                                 // guided attention: dP += c W, D += c rowsum(P W) on valid query rows
        const int ql = min(a.qlen[b], a.Lq);
        if (q < ql) gc = *a.ga_scale;
        ga_iq = 1.f / (float)max(ql, 1); ga_ik = 1.f / (float)max(kend, 1);
        Dq += gc * a.ga_rows[(long)z * a.Lq + qc];
    }
    if (wlive && hi == 0 && q < a.Lq) a.dsum[(long)z * a.Lq + q] = Dq;   // the dK/dV kernel reads it
    int ktiles = (kend + 63) / 64;
    if (causal) {
        int qlast = min(qb0 + QB - 1, a.Lq - 1);
        if (a.qskip) qlast = min(qlast, ((qlive + 63) & ~63) - 1);
        ktiles = min(ktiles, qlast / 64 + 1);
    }
    const int wkt = !wlive ? 0 : (causal ? min(ktiles, min(qw0 + 31, a.Lq - 1) / 64 + 1) : ktiles);
    f32x16_t dq[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
    const float sl2 = a.scale * A32_LOG2E, lse2 = a.lse[(long)z * a.Lq + qc] * A32_LOG2E;
    const uint32_t hk = (uint32_t)((a.Lk + 1) >> 1);
    const uint32_t xrow = ((uint32_t)((long)z * a.Lq + qc) * hk + (uint32_t)(2 * hi)) * GOLD + a.drop.key;
    const uint32_t t16 = a.drop.thresh & 0xffff0000u;
    const float dscale = a.drop.scale;

    Stage<DH, NT> rk, rv;
    if (ktiles > 0) {
        stage_fetch<DH, NT>(rk, K, a.ldk, 0, a.Lk, tid); stage_fetch<DH, NT>(rv, V, a.ldv, 0, a.Lk, tid);
        stage_store<DH, NT>(sK, rk, tid); stage_store<DH, NT>(sV, rv, tid);
        if (ktiles > 1) { stage_fetch<DH, NT>(rk, K, a.ldk, 64, a.Lk, tid); stage_fetch<DH, NT>(rv, V, a.ldv, 64, a.Lk, tid); }
    }
    __syncthreads();
    for (int kt = 0; kt < ktiles; ++kt) {
        const int k0 = kt * 64, cur = kt & 1;
        const bf16_t* tK = sK + cur * TILE;
        const bf16_t* tV = sV + cur * TILE;
        const bool active = kt < wkt;
        const bool interior = k0 + 64 <= kend && (!causal || k0 + 63 <= qw0);
        const uint32_t xt = xrow + (uint32_t)(k0 >> 1) * GOLD;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t == 1 && kt + 1 < ktiles) {                         // (between the two halves: the first half's reads of this tile are issued)
                stage_store<DH, NT>(sK + (cur ^ 1) * TILE, rk, tid); stage_store<DH, NT>(sV + (cur ^ 1) * TILE, rv, tid);
                if (kt + 2 < ktiles) { stage_fetch<DH, NT>(rk, K, a.ldk, k0 + 128, a.Lk, tid); stage_fetch<DH, NT>(rv, V, a.ldv, k0 + 128, a.Lk, tid); }
            }
            if (!active) continue;
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                s = mma32(frag_a<DH>(tK, t * 32 + l31, ks, hi), qf[ks], s);
                dp = mma32(frag_a<DH>(tV, t * 32 + l31, ks, hi), dof[ks], dp);
            }
            if (interior) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = fast_exp2(fmaf(s[r], sl2, -lse2));
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + t * 32 + crow(r, hi);
                    const bool ok = key < kend && (!causal || key <= q);
                    s[r] = ok ? fast_exp2(fmaf(s[r], sl2, -lse2)) : 0.f;
                }
            }
            if (a.drop.thresh) {
#pragma unroll
                for (int pr = 0; pr < 8; ++pr) {
                    const uint32_t w = hash_body(xt + (uint32_t)(t * 16 + (pr >> 1) * 4 + (pr & 1)) * GOLD);
                    dp[2 * pr] = (w << 16) >= t16 ? dp[2 * pr] * dscale : 0.f;
                    dp[2 * pr + 1] = w >= t16 ? dp[2 * pr + 1] * dscale : 0.f;
                }
            }
            if (__any(gc != 0.f)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dp[r] += gc * ga_w(q, k0 + t * 32 + crow(r, hi), ga_iq, ga_ik, a.ga_inv2s2);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = s[r] * (dp[r] - Dq);           // (x a.scale at the end)
            const bf16x8_t p0 = pack8<0>(s), p1 = pack8<1>(s);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                dq[dt] = mma32(frag_t<DH>(tK, t * 32, dt * 32, lane), p0, dq[dt]);
                dq[dt] = mma32(frag_t<DH>(tK, t * 32 + 16, dt * 32, lane), p1, dq[dt]);
            }
        }
        __syncthreads();
    }
    if (wlive && q < a.Lq) store_own<DH>(dqo, dq, a.scale, hi);
}

// ================================================================================================ dK, dV
template <int DH, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn32_dkv_kernel(AttnArgs a) {
    constexpr int NKS = DH / 16, NDT = DH / 32, TILE = 64 * DH, NT = NW * 64, KB = NW * 32;
    __shared__ __attribute__((aligned(16))) bf16_t sQ[2 * TILE];
    __shared__ __attribute__((aligned(16))) bf16_t sO[2 * TILE];
    __shared__ __attribute__((aligned(16))) float sL[2 * 64], sD[2 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    int blk, z;
    a32_block(blk, z);
    const int b = z / a.H, h = z - b * a.H;
    const int kb0 = blk * KB, kw0 = kb0 + wave * 32, key = kw0 + l31, kc = min(key, a.Lk - 1);
    int kend = a.Lk;
    if (a.mask_mode & 1) kend = min(kend, a.klen[b]);
    bf16_t* dko = reinterpret_cast<bf16_t*>(a.dk) + ((long)b * a.Lk + kc) * a.lddk + h * DH;
    bf16_t* dvo = reinterpret_cast<bf16_t*>(a.dv) + ((long)b * a.Lk + kc) * a.lddv + h * DH;
    const bool wlive = kw0 < kend;                                   // a wave of masked keys: zero gradients
    if (!wlive && key < a.Lk) { zero_own<DH>(dko, hi); zero_own<DH>(dvo, hi); }
    if (kb0 >= kend) return;
    const bool causal = a.mask_mode & 2;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q) + (long)b * a.Lq * a.ldq + h * DH;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(a.k) + (long)b * a.Lk * a.ldk + h * DH;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(a.v) + (long)b * a.Lk * a.ldv + h * DH;
    const bf16_t* dO = reinterpret_cast<const bf16_t*>(a.dout) + (long)b * a.Lq * a.ldo + h * DH;
    bf16x8_t kf[NKS], vf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) { kf[ks] = frag_own(K, a.ldk, kc, ks, hi); vf[ks] = frag_own(V, a.ldv, kc, ks, hi); }
    const bool key_ok = key < kend;
    float gc = 0.f, ga_iq = 0.f, ga_ik = 0.f;
    int ga_ql = 0;
    if (a.ga_rows) {
        ga_ql = min(a.qlen[b], a.Lq);
        gc = *a.ga_scale;
        ga_iq = 1.f / (float)max(ga_ql, 1); ga_ik = 1.f / (float)max(kend, 1);
    }
    f32x16_t dk[NDT], dv[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
    int qtiles = (a.Lq + 63) / 64;
    if (a.qskip) qtiles = min(qtiles, (a.qskip[b] + 63) / 64);       // tiles of padded query rows contribute nothing (d context = 0)
    const int qt0 = causal ? kb0 / 64 : 0;                           // queries before this key block never see it
    const int wqt0 = !wlive ? qtiles : (causal ? kw0 / 64 : 0);      // first query tile this wave computes
    const float sl2 = a.scale * A32_LOG2E, dscale = a.drop.scale;
    // dropout: word(row = z Lq + qq, pair = key >> 1), half (key & 1); x = (row hk + (key >> 1)) GOLD + dropkey, advanced by hk GOLD per query row
    const uint32_t hk = (uint32_t)((a.Lk + 1) >> 1);
    const uint32_t hkg = hk * GOLD, hkg5 = 5u * hkg;
    const uint32_t xkey = ((uint32_t)z * (uint32_t)a.Lq * hk + (uint32_t)(kc >> 1)) * GOLD + a.drop.key + (uint32_t)(4 * hi) * hkg;
    const uint32_t hsh = (kc & 1) ? 0u : 16u;                        // shift that brings this key's half to the top
    const uint32_t t16 = a.drop.thresh & 0xffff0000u;

    Stage<DH, NT> rq, ro;
    float r_l = 0.f, r_d = 0.f;
    if (qt0 < qtiles) {
        stage_fetch<DH, NT>(rq, Q, a.ldq, qt0 * 64, a.Lq, tid); stage_fetch<DH, NT>(ro, dO, a.ldo, qt0 * 64, a.Lq, tid);
        if (tid < 64) { const int qq = min(qt0 * 64 + tid, a.Lq - 1); r_l = a.lse[(long)z * a.Lq + qq] * A32_LOG2E; r_d = a.dsum[(long)z * a.Lq + qq]; }
        stage_store<DH, NT>(sQ + (qt0 & 1) * TILE, rq, tid); stage_store<DH, NT>(sO + (qt0 & 1) * TILE, ro, tid);
        if (tid < 64) { sL[(qt0 & 1) * 64 + tid] = r_l; sD[(qt0 & 1) * 64 + tid] = r_d; }
        if (qt0 + 1 < qtiles) {
            stage_fetch<DH, NT>(rq, Q, a.ldq, qt0 * 64 + 64, a.Lq, tid); stage_fetch<DH, NT>(ro, dO, a.ldo, qt0 * 64 + 64, a.Lq, tid);
            if (tid < 64) { const int qq = min(qt0 * 64 + 64 + tid, a.Lq - 1); r_l = a.lse[(long)z * a.Lq + qq] * A32_LOG2E; r_d = a.dsum[(long)z * a.Lq + qq]; }
        }
    }
    __syncthreads();
    for (int qt = qt0; qt < qtiles; ++qt) {
        const int q0 = qt * 64, cur = qt & 1;
        const bf16_t* tQ = sQ + cur * TILE;
        const bf16_t* tO = sO + cur * TILE;
        const float* tL = sL + cur * 64;
        const float* tD = sD + cur * 64;
        const bool active = qt >= wqt0;
        // all 32 keys of this wave valid and visible to all 64 queries of the tile?  (wave-uniform)
        const bool interior = kw0 + 32 <= kend && q0 + 64 <= a.Lq && (!causal || kw0 + 31 <= q0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t == 1 && qt + 1 < qtiles) {
                stage_store<DH, NT>(sQ + (cur ^ 1) * TILE, rq, tid); stage_store<DH, NT>(sO + (cur ^ 1) * TILE, ro, tid);
                if (tid < 64) { sL[(cur ^ 1) * 64 + tid] = r_l; sD[(cur ^ 1) * 64 + tid] = r_d; }
                if (qt + 2 < qtiles) {
                    stage_fetch<DH, NT>(rq, Q, a.ldq, q0 + 128, a.Lq, tid); stage_fetch<DH, NT>(ro, dO, a.ldo, q0 + 128, a.Lq, tid);
                    if (tid < 64) { const int qq = min(q0 + 128 + tid, a.Lq - 1); r_l = a.lse[(long)z * a.Lq + qq] * A32_LOG2E; r_d = a.dsum[(long)z * a.Lq + qq]; }
                }
            }
            if (!active) continue;
            f32x16_t s, dp;                                          // s[r] = S[q = q0 + 32 t + crow(r, hi)][key = own]
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                s = mma32(frag_a<DH>(tQ, t * 32 + l31, ks, hi), kf[ks], s);
                dp = mma32(frag_a<DH>(tO, t * 32 + l31, ks, hi), vf[ks], dp);
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4_t lq = *reinterpret_cast<const f32x4_t*>(tL + t * 32 + g4 * 8 + hi * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = g4 * 4 + j;
                    if (interior) s[r] = fast_exp2(fmaf(s[r], sl2, -lq[j]));
                    else {
                        const int qq = q0 + t * 32 + crow(r, hi);
                        const bool ok = key_ok && qq < a.Lq && (!causal || key <= qq);
                        s[r] = ok ? fast_exp2(fmaf(s[r], sl2, -lq[j])) : 0.f;
                    }
                }
            }
            f32x16_t pd;
            if (a.drop.thresh) {
                uint32_t x = xkey + (uint32_t)(q0 + t * 32) * hkg;
#pragma unroll
                for (int r = 0; r < 16; ++r) {                        // query row q0 + 32 t + 8 (r >> 2) + (r & 3) (+ 4 hi, in xkey)
                    if (r) x += (r & 3) ? hkg : hkg5;
                    const uint32_t w = hash_body(x);
                    const bool keep = (w << hsh) >= t16;
                    dp[r] = keep ? dp[r] * dscale : 0.f;
                    pd[r] = keep ? s[r] : 0.f;
                }
            } else pd = s;
            if (gc != 0.f) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qq = q0 + t * 32 + crow(r, hi);
                    dp[r] += qq < ga_ql ? gc * ga_w(qq, key, ga_iq, ga_ik, a.ga_inv2s2) : 0.f;
                }
            }
            {
                const bf16x8_t p0 = pack8<0>(pd), p1 = pack8<1>(pd);    // dV^T[d][key] += sum_q dO[q][d] Pd[q][key]
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    dv[dt] = mma32(frag_t<DH>(tO, t * 32, dt * 32, lane), p0, dv[dt]);
                    dv[dt] = mma32(frag_t<DH>(tO, t * 32 + 16, dt * 32, lane), p1, dv[dt]);
                }
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(tD + t * 32 + g4 * 8 + hi * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) s[g4 * 4 + j] = s[g4 * 4 + j] * (dp[g4 * 4 + j] - d4[j]);
            }
            {
                const bf16x8_t p0 = pack8<0>(s), p1 = pack8<1>(s);      // dK^T[d][key] += sum_q Q[q][d] dS[q][key]
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    dk[dt] = mma32(frag_t<DH>(tQ, t * 32, dt * 32, lane), p0, dk[dt]);
                    dk[dt] = mma32(frag_t<DH>(tQ, t * 32 + 16, dt * 32, lane), p1, dk[dt]);
                }
            }
        }
        __syncthreads();
    }
    if (wlive && key < a.Lk) { store_own<DH>(dko, dk, a.scale, hi); store_own<DH>(dvo, dv, dscale, hi); }
}

template <int DH>
int launch32(const AttnArgs& a, int which, hipStream_t st) {
    constexpr int NW = 4;
    if (which == 0) {
        hipLaunchKernelGGL((attn32_fwd_kernel<DH, NW>), dim3(cdiv(a.Lq, NW * 32), a.B * a.H), dim3(NW * 64), 0, st, a);
    } else if (which == 1) {
        hipLaunchKernelGGL((attn32_dq_kernel<DH, NW>), dim3(cdiv(a.Lq, NW * 32), a.B * a.H), dim3(NW * 64), 0, st, a);
    } else {
        hipLaunchKernelGGL((attn32_dkv_kernel<DH, NW>), dim3(cdiv(a.Lk, NW * 32), a.B * a.H), dim3(NW * 64), 0, st, a);
    }
    B2S_LAUNCH_CHECK();
    return 0;
}
}  // namespace

bool b2s_flash32_supported(int dh) { return dh == 32 || dh == 64 || dh == 96; }
int b2s_flash32_launch(const AttnArgs& a, int dh, int which, hipStream_t st) {
    switch (dh) {
        case 32: return launch32<32>(a, which, st);
        case 64: return launch32<64>(a, which, st);
        case 96: return launch32<96>(a, which, st);
    }
    return b2s_fail(__FILE__, __LINE__, "bf16 attention supports head sizes 32/64/96 (got %d)", dh);
}
