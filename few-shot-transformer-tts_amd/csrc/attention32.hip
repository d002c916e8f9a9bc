// bf16 training attention for gfx950 on the 32x32x16 MFMA (reference: transformer/attention.py:72-92 dot_product_attention and its autograd
// backward).  Same contract as the kernels of attention.hip (AttnArgs; the fp32 parity mode and head sizes this file does not cover stay there).
//
// Structure (forward, dQ, dK/dV alike): a workgroup of NW waves owns NW x 32 rows of one (batch, head) -- query rows in the forward and dQ kernels,
// key rows in the dK/dV kernel -- and streams 64-row tiles of the other sequence through a two-deep LDS ring filled by LDS-DMA (global_load_lds_dwordx4:
// no staging registers, no ds_write pass; the DMA of the next tile is in flight while the current one is consumed; ONE barrier per tile).  Every product is computed "swapped" so that a lane owns ONE row of the
// wave's 32 and 16 (of 32) columns of the streamed tile in its accumulator registers (32x32 C layout: col = lane & 31, row = (r & 3) + 8 (r >> 2)
// + 4 (lane >> 5)):
//      S^T = K Q^T, dP^T = V dO^T             A = 32 streamed rows x 16 features, one ds_read_b128 per MFMA; B = the wave's own rows, in registers
//      O^T += V^T P^T, dQ^T += K^T dS^T, ..   A = 32 features x 16 streamed rows, two ds_read_b64_tr_b16 per MFMA; B = the packed accumulator of the
//                                             first product AS IT LIES: k-slot j of lane half hi <-> streamed row 16 s + 4 hi + (j & 3) + 8 (j >> 2),
//                                             the transposed reads fetch exactly those rows, so no lane exchange is needed
// so the softmax is in registers (a row's other half of the keys lives in lane ^ 32: one v_permlane32_swap per reduction), the dropout word of a
// four adjacent keys costs four integer operations in the forward / dQ kernels (b2s_common.h: b2s_keep_w), and 32 rows per wave halve the LDS bytes per FLOP of the 16-row
// kernels.  LDS tiles are unpadded [64][dh] images with the 16-byte chunk index XORed by row bits (conflict-free for the b128 reads of one tile row
// per lane AND for the transposed reads of 4 rows x 64 bytes per 32 lanes).
#include "b2s_common.h"
#include "attention.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((address_space(3))) bf16x4_t* lds_b64_ptr;

typedef __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
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
__device__ inline float vmax3(float a, float b, float c) { float d; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
constexpr float A32_LOG2E = 1.4426950408889634f, A32_LN2 = 0.6931471805599453f;
constexpr float A32_LAZY = 8.f;                       // running-maximum slack, as attention.hip: B2S_LAZY

// chunk swizzle of a tile row: 16 rows of one b128 lane group land on 16 distinct 16-byte slots, 4 consecutive rows x 4 chunks of a transposed read too
template <int DH> __device__ inline int swz(int row) { return DH == 64 ? ((((row >> 1) & 1) << 2) | ((row >> 2) & 3)) : ((row >> 2) & 3); }

// ---- global -> LDS staging of a [64][DH] tile by LDS-DMA.  One wave-instruction writes 64 consecutive 16-byte chunks (lane-linear), so the
// swizzle is applied on the SOURCE side: the lane that owns LDS chunk position p = row * CPR + c' fetches the global chunk c' ^ swz(row) of that row.
// Each wave issues NI = CPR / NW instructions per tile; completion is the wave's vmcnt, publication the workgroup barrier.
template <int DH, int NW> struct Dma {
    static constexpr int CPR = DH / 8, NI = CPR / NW;
    static_assert(CPR % NW == 0, "tile must split evenly over the waves");
    int row[NI], cb[NI];                                              // tile row and byte offset inside a source row of the chunk this lane fetches
    const char* base; uint32_t ldb; int last;
    __device__ inline void init(const bf16_t* b, long ld_, int nrows_, int wave, int lane) {
        base = reinterpret_cast<const char*>(b); ldb = (uint32_t)ld_ * 2u; last = nrows_ - 1;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int p = (wave * NI + i) * 64 + lane, r = p / CPR, c = (p - r * CPR) ^ swz<DH>(r);
            row[i] = r; cb[i] = c * 16;
        }
    }
    // rows beyond the end of the sequence fetch the last row (finite data; every consumer masks those rows)
    __device__ inline void issue(bf16_t* tile, int row0, int wave) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const uint32_t voff = (uint32_t)min(row0 + row[i], last) * ldb + (uint32_t)cb[i];      // uniform base + 32-bit lane offset
            __builtin_amdgcn_global_load_lds((gptr_t)(base + voff), (lptr_t)(tile + (wave * NI + i) * 512), 16, 0, 0);
        }
    }
};
// compiler-only fence: LDS fragment loads are not hoisted across it (bounds the registers the scheduler spends on loads in flight)
__device__ inline void cfence() { asm volatile("" ::: "memory"); }
__device__ inline void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
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

// XCD-aware (block rank, head) assignment.  Workgroups go to the 8 XCDs round-robin by linear id and every XCD has its own L2; with B * H a multiple
// of 8 every XCD owns B * H / 8 heads (their K / V stay in its L2) and walks them rank-major: every head's block of rank 0 first, then rank 1, ...
// -- the callers map rank 0 to the block with the most tiles (causal attention), so that the long workgroups start first on every XCD.
__device__ inline void a32_block(int& rank, int& z) {
    const int nx = gridDim.x, Z = gridDim.y, orig = blockIdx.y * nx + blockIdx.x;
    if ((Z & 7) == 0) {
        const int zc = Z >> 3, x = orig & 7, i = orig >> 3;
        rank = i / zc; z = x * zc + (i - rank * zc);
    } else {                                                            // as attention.hip: xcd_block
        const int nwg = nx * Z, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
        z = wg / nx; rank = wg - z * nx;
    }
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
constexpr uint32_t GOLD = 0x9E3779B1u;


// first products of one 64-row tile against the wave's own rows: s[t] = tile rows [32 t, 32 t + 32) x own
template <int DH>
__device__ inline void first2(f32x16_t (&s)[2], const bf16_t* tile, const bf16x8_t (&own)[DH / 16], int l31, int hi) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < DH / 16; ++ks) s[t] = mma32(frag_a<DH>(tile, t * 32 + l31, ks, hi), own[ks], s[t]);
    }
}
template <int DH>
__device__ inline void first1(f32x16_t& s, const bf16_t* tile, int t, const bf16x8_t (&own)[DH / 16], int l31, int hi) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) s = mma32(frag_a<DH>(tile, t * 32 + l31, ks, hi), own[ks], s);
}
// second product: acc[dt] += (tile rows [32 t, +32) transposed, features [32 dt, +32)) x packed w
template <int DH>
__device__ inline void second1(f32x16_t (&acc)[DH / 32], const bf16_t* tile, int t, const f32x16_t& w, int lane) {
    const bf16x8_t p0 = pack8<0>(w), p1 = pack8<1>(w);
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt) {
        acc[dt] = mma32(frag_t<DH>(tile, t * 32, dt * 32, lane), p0, acc[dt]);
        acc[dt] = mma32(frag_t<DH>(tile, t * 32 + 16, dt * 32, lane), p1, acc[dt]);
    }
}
// dropout (b2s_common.h: b2s_keep_w): the two words of each of this lane's 8 key quads of the 64-key tile at k0 -- forward / dQ, where the lane owns a
// weight row: xrow = seed(row) + hi * GOLD; quad (t, g4) = keys k0 + 32 t + 8 g4 + 4 hi + {0 .. 3} = accumulator registers 4 g4 .. 4 g4 + 3 of half t
__device__ inline void quad_words(uint32_t (&wd)[16], uint32_t xrow, int k0) {
    const uint32_t xt = xrow + (uint32_t)(k0 >> 2) * GOLD;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const uint32_t x = xt + (uint32_t)(t * 8 + g4 * 2) * GOLD, y = x ^ (x >> 16);
            wd[t * 8 + g4 * 2] = y * B2S_WC0; wd[t * 8 + g4 * 2 + 1] = y * B2S_WC1;
        }
}
// 0xffff in every 16-bit field of w that is kept: (int16) field >= ts  <=>  (ts - 1) - field < 0 (saturating, so the sign survives)
__device__ inline uint32_t keep_mask16(uint32_t w, uint32_t tsm1, uint32_t fifteen) {
    uint32_t d, m;
    asm("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(d) : "v"(tsm1), "v"(w));
    asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(m) : "v"(fifteen), "v"(d));
    return m;
}
// B operands of a second product from accumulator registers [8 S8, 8 S8 + 8), with the dropout fields of words wd[4 S8 .. 4 S8 + 3] applied to the bf16 pairs
template <int S8> __device__ inline bf16x8_t pack8_drop(const f32x16_t& v, const uint32_t* wd, uint32_t tsm1, uint32_t fifteen) {
    const u32x4_t u = {f2bf2(v[S8 * 8 + 0], v[S8 * 8 + 1]) & keep_mask16(wd[S8 * 4 + 0], tsm1, fifteen), f2bf2(v[S8 * 8 + 2], v[S8 * 8 + 3]) & keep_mask16(wd[S8 * 4 + 1], tsm1, fifteen),
                       f2bf2(v[S8 * 8 + 4], v[S8 * 8 + 5]) & keep_mask16(wd[S8 * 4 + 2], tsm1, fifteen), f2bf2(v[S8 * 8 + 6], v[S8 * 8 + 7]) & keep_mask16(wd[S8 * 4 + 3], tsm1, fifteen)};
    return __builtin_bit_cast(bf16x8_t, u);
}
template <int DH>
__device__ inline void second1p(f32x16_t (&acc)[DH / 32], const bf16_t* tile, int t, bf16x8_t p0, bf16x8_t p1, int lane) {
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt) {
        acc[dt] = mma32(frag_t<DH>(tile, t * 32, dt * 32, lane), p0, acc[dt]);
        acc[dt] = mma32(frag_t<DH>(tile, t * 32 + 16, dt * 32, lane), p1, acc[dt]);
    }
}

// ================================================================================================ forward
// Software pipeline inside a wave: the logits of tile t + 1 (MFMA) are issued before the exponentials / row sums / dropout selects / bf16 packs of
// tile t (VALU), and the dropout words of tile t + 1 are hashed beside the P V products of tile t -- the K ring therefore runs one tile ahead of the V ring.
template <int DH, int NW, bool DROP>
__global__ __launch_bounds__(NW * 64, 2) void attn32_fwd_kernel(AttnArgs a) {
    constexpr int NKS = DH / 16, NDT = DH / 32, TILE = 64 * DH, QB = NW * 32;
    __shared__ __attribute__((aligned(16))) bf16_t sK[2 * TILE];
    __shared__ __attribute__((aligned(16))) bf16_t sV[2 * TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    int blk, z;
    a32_block(blk, z);
    blk = gridDim.x - 1 - blk;                                      // rank 0 = the last query block (causal: the most key tiles)
    const int b = z / a.H, h = z - b * a.H;
    // this utterance's rows of the q-side / k-side tensors (AttnArgs::qoff / koff: ragged row offsets; default: the padded [B, L] layout)
    const int qrow0 = a.qoff ? a.qoff[b] : b * a.Lq, Lq = a.qoff ? a.qoff[b + 1] - qrow0 : a.Lq;
    const int krow0 = a.koff ? a.koff[b] : b * a.Lk, Lk = a.koff ? a.koff[b + 1] - krow0 : a.Lk;
    const int qb0 = blk * QB, qw0 = qb0 + wave * 32, q = qw0 + l31, qc = min(q, Lq - 1);
    // padded query rows (AttnArgs::qskip): 64-row tiles wholly at or beyond qskip[b] are not computed, their rows are written as zeros
    const int qlive = a.qskip ? min(a.qskip[b], Lq) : Lq;
    const bool wlive = (qw0 & ~63) < qlive && qw0 < Lq;           // wave-uniform
    bf16_t* out = reinterpret_cast<bf16_t*>(a.out) + ((long)qrow0 + qc) * a.ldo + h * DH;
    if (!wlive && q < Lq) {
        zero_own<DH>(out, hi);
        if (hi == 0 && a.lse) a.lse[(long)z * a.Lq + q] = 0.f;
    }
    if (qb0 >= qlive) return;                                       // nothing live in this workgroup (uniform)
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q) + (long)qrow0 * a.ldq + h * DH;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(a.k) + (long)krow0 * a.ldk + h * DH;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(a.v) + (long)krow0 * a.ldv + h * DH;
    const bool causal = a.mask_mode & 2;
    int kend = Lk;
    if (a.mask_mode & 1) kend = min(kend, a.klen[b]);
    int ktiles = (kend + 63) / 64;
    if (causal) {
        int qlast = min(qb0 + QB - 1, Lq - 1);
        if (a.qskip) qlast = min(qlast, ((qlive + 63) & ~63) - 1);
        ktiles = min(ktiles, qlast / 64 + 1);
    }
    const int wkt = !wlive ? 0 : (causal ? min(ktiles, min(qw0 + 31, Lq - 1) / 64 + 1) : ktiles);   // key tiles this wave computes
    Dma<DH, NW> dmk, dmv;
    dmk.init(K, a.ldk, Lk, wave, lane); dmv.init(V, a.ldv, Lk, wave, lane);
    if (ktiles > 0) { dmk.issue(sK, 0, wave); dmv.issue(sV, 0, wave); }
    if (ktiles > 1) dmk.issue(sK + TILE, 64, wave);
    bf16x8_t qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = frag_own(Q, a.ldq, qc, ks, hi);

    f32x16_t o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    const float sl2 = a.scale * A32_LOG2E;
    const uint32_t xrow = b2s_wseed(a.drop, (uint32_t)((long)z * a.Lq + qc)) + (uint32_t)hi * GOLD;
    const uint32_t tsm1 = ((uint32_t)(b2s_wthresh(a.drop) - 1) & 0xffffu) * 0x10001u, fifteen = 0x000f000fu;
    f32x16_t s0[2], s1[2];
    uint32_t wd[16];
    wait_vm0();
    __syncthreads();
    if (wkt > 0) {
        first2<DH>(s0, sK, qf, l31, hi);
        if (DROP) quad_words(wd, xrow, 0);
    }
    __syncthreads();                                                // every wave has read K(0): iteration 0 refills its slot
    auto iter = [&](const int kt, f32x16_t (&sc)[2], f32x16_t (&sn)[2]) {
        const int k0 = kt * 64, cur = kt & 1;
        if (kt + 2 < ktiles) dmk.issue(sK + cur * TILE, k0 + 128, wave);          // K(kt + 2) over K(kt), V(kt + 1) over V(kt - 1)
        if (kt + 1 < ktiles) dmv.issue(sV + (cur ^ 1) * TILE, k0 + 64, wave);
        if (kt < wkt) {
            // every key of the tile visible to every row of this wave?  (wave-uniform; the common case skips the mask arithmetic)
            const bool interior = k0 + 64 <= kend && (!causal || k0 + 63 <= qw0);
            if (!interior) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = k0 + t * 32 + crow(r, hi);
                        sc[t][r] = (key < kend && (!causal || key <= q)) ? sc[t][r] : -INFINITY;
                    }
            }
            float mx = sc[0][0];                                      // (v_max3 by hand: fmaxf on MFMA results costs a canonicalising v_max per operand)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = vmax3(mx, sc[0][r], sc[1][r]);
            mx = half_max(mx) * sl2;
            const bool grow = mx > m + A32_LAZY;                     // also true for the first finite maximum (m = -inf)
            if (__any(grow)) {
                const float mn = grow ? mx : m;
                const float alpha = mn == m ? 1.f : fast_exp2(m - mn);          // m = -inf -> 0
                m = mn; l *= alpha;
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            }
            const float mref = m == -INFINITY ? 0.f : m;
            // ---- one block from here: logits of the NEXT tile (results unused after the wave's last tile) beside the softmax arithmetic of this one
            first2<DH>(sn, sK + (cur ^ 1) * TILE, qf, l31, hi);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = fast_exp2(fmaf(sc[t][r], sl2, -mref));      // masked: 2^-inf = 0
                    l += p;
                    sc[t][r] = p;
                }
            if (DROP) {
                second1p<DH>(o, sV + cur * TILE, 0, pack8_drop<0>(sc[0], wd, tsm1, fifteen), pack8_drop<1>(sc[0], wd, tsm1, fifteen), lane);
                second1p<DH>(o, sV + cur * TILE, 1, pack8_drop<0>(sc[1], wd + 8, tsm1, fifteen), pack8_drop<1>(sc[1], wd + 8, tsm1, fifteen), lane);
            } else {
                second1<DH>(o, sV + cur * TILE, 0, sc[0], lane);
                second1<DH>(o, sV + cur * TILE, 1, sc[1], lane);
            }
            if (DROP) quad_words(wd, xrow, k0 + 64);                  // the next tile's words, beside the P V products
        }
        wait_vm0();
        __syncthreads();
    };
    for (int kt = 0; kt < ktiles; kt += 2) {
        iter(kt, s0, s1);
        if (kt + 1 < ktiles) iter(kt + 1, s1, s0);
    }
    l = half_sum(l);
    if (wlive && q < Lq) {
        const float inv = 1.f / l;
        store_own<DH>(out, o, inv * a.drop.scale, hi);
        if (hi == 0 && a.lse) a.lse[(long)z * a.Lq + q] = (m + __log2f(l)) * A32_LN2;
    }
}

// ================================================================================================ dQ
template <int DH, int NW, bool DROP>
__global__ __launch_bounds__(NW * 64, 2) void attn32_dq_kernel(AttnArgs a) {
    constexpr int NKS = DH / 16, NDT = DH / 32, TILE = 64 * DH, QB = NW * 32;
    __shared__ __attribute__((aligned(16))) bf16_t sK[2 * TILE];
    __shared__ __attribute__((aligned(16))) bf16_t sV[2 * TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    int blk, z;
    a32_block(blk, z);
    blk = gridDim.x - 1 - blk;
    const int b = z / a.H, h = z - b * a.H;
    // this utterance's rows of the q-side / k-side tensors (AttnArgs::qoff / koff: ragged row offsets; default: the padded [B, L] layout)
    const int qrow0 = a.qoff ? a.qoff[b] : b * a.Lq, Lq = a.qoff ? a.qoff[b + 1] - qrow0 : a.Lq;
    const int krow0 = a.koff ? a.koff[b] : b * a.Lk, Lk = a.koff ? a.koff[b + 1] - krow0 : a.Lk;
    const int qb0 = blk * QB, qw0 = qb0 + wave * 32, q = qw0 + l31, qc = min(q, Lq - 1);
    const int qlive = a.qskip ? min(a.qskip[b], Lq) : Lq;
    const bool wlive = (qw0 & ~63) < qlive && qw0 < Lq;
    bf16_t* dqo = reinterpret_cast<bf16_t*>(a.dq) + ((long)qrow0 + qc) * a.lddq + h * DH;
    if (!wlive && q < Lq) {                                        // padded query rows: d context is zero there, so is dQ
        zero_own<DH>(dqo, hi);
        if (hi == 0) a.dsum[(long)z * a.Lq + q] = 0.f;
    }
    if (qb0 >= qlive) return;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q) + (long)qrow0 * a.ldq + h * DH;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(a.k) + (long)krow0 * a.ldk + h * DH;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(a.v) + (long)krow0 * a.ldv + h * DH;
    const bf16_t* dO = reinterpret_cast<const bf16_t*>(a.dout) + (long)qrow0 * a.ldo + h * DH;
    const bf16_t* O = reinterpret_cast<const bf16_t*>(a.oref) + (long)qrow0 * a.ldo + h * DH;
    const bool causal = a.mask_mode & 2;
    int kend = Lk;
    if (a.mask_mode & 1) kend = min(kend, a.klen[b]);
    int ktiles = (kend + 63) / 64;
    if (causal) {
        int qlast = min(qb0 + QB - 1, Lq - 1);
        if (a.qskip) qlast = min(qlast, ((qlive + 63) & ~63) - 1);
        ktiles = min(ktiles, qlast / 64 + 1);
    }
    const int wkt = !wlive ? 0 : (causal ? min(ktiles, min(qw0 + 31, Lq - 1) / 64 + 1) : ktiles);
    Dma<DH, NW> dmk, dmv;
    dmk.init(K, a.ldk, Lk, wave, lane); dmv.init(V, a.ldv, Lk, wave, lane);
    if (ktiles > 0) { dmk.issue(sK, 0, wave); dmv.issue(sV, 0, wave); }
    bf16x8_t qf[NKS], dof[NKS];
    float Dq = 0.f;                                                  // D[q] = sum_d dO[q][d] O[q][d]: this lane holds half of the row's features
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        qf[ks] = frag_own(Q, a.ldq, qc, ks, hi); dof[ks] = frag_own(dO, a.ldo, qc, ks, hi);
        Dq += frag_dot(frag_own(O, a.ldo, qc, ks, hi), dof[ks]);
    }
    Dq = half_sum(Dq);
    if (wlive && hi == 0 && q < Lq) a.dsum[(long)z * a.Lq + q] = Dq;   // the dK/dV kernel reads it
    f32x16_t dq[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
    const float sl2 = a.scale * A32_LOG2E, lse2 = a.lse[(long)z * a.Lq + qc] * A32_LOG2E;
    const uint32_t xrow = b2s_wseed(a.drop, (uint32_t)((long)z * a.Lq + qc)) + (uint32_t)hi * GOLD;
    const int ts32 = (int)((uint32_t)b2s_wthresh(a.drop) << 16);        // a field brought to the top 16 bits, compared as a signed word
    const float dscale = a.drop.scale;
    wait_vm0();
    __syncthreads();
    for (int kt = 0; kt < ktiles; ++kt) {
        const int k0 = kt * 64, cur = kt & 1;
        const bf16_t* tK = sK + cur * TILE;
        const bf16_t* tV = sV + cur * TILE;
        if (kt + 1 < ktiles) { dmk.issue(sK + (cur ^ 1) * TILE, k0 + 64, wave); dmv.issue(sV + (cur ^ 1) * TILE, k0 + 64, wave); }
        if (kt < wkt) {
            const bool interior = k0 + 64 <= kend && (!causal || k0 + 63 <= qw0);
            f32x16_t s[2], dp[2];
            first2<DH>(s, tK, qf, l31, hi);
            first2<DH>(dp, tV, dof, l31, hi);
            if (!interior) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = k0 + t * 32 + crow(r, hi);
                        s[t][r] = (key < kend && (!causal || key <= q)) ? s[t][r] : -INFINITY;
                    }
            }
            uint32_t wd[16];
            if (DROP) quad_words(wd, xrow, k0);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[t][r] = fast_exp2(fmaf(s[t][r], sl2, -lse2));    // masked: 2^-inf = 0
                if (DROP) {
#pragma unroll
                    for (int pr = 0; pr < 8; ++pr) {
                        const uint32_t w = wd[t * 8 + pr];
                        dp[t][2 * pr] = (int)(w << 16) >= ts32 ? dp[t][2 * pr] * dscale : 0.f;
                        dp[t][2 * pr + 1] = (int)w >= ts32 ? dp[t][2 * pr + 1] * dscale : 0.f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) s[t][r] = s[t][r] * (dp[t][r] - Dq);       // (x a.scale at the end)
                second1<DH>(dq, tK, t, s[t], lane);
            }
        }
        wait_vm0();
        __syncthreads();
    }
    if (wlive && q < Lq) store_own<DH>(dqo, dq, a.scale, hi);
}

// ================================================================================================ dK, dV
template <int DH, int NW, bool DROP>
__global__ __launch_bounds__(NW * 64, 2) void attn32_dkv_kernel(AttnArgs a) {
    constexpr int NKS = DH / 16, NDT = DH / 32, TILE = 64 * DH, KB = NW * 32;
    __shared__ __attribute__((aligned(16))) bf16_t sQ[2 * TILE];
    __shared__ __attribute__((aligned(16))) bf16_t sO[2 * TILE];
    __shared__ __attribute__((aligned(16))) float sL[2 * 64], sD[2 * 64];
    __shared__ __attribute__((aligned(16))) uint32_t sS[2 * 64];     // dropout seeds of the tile's query rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    int blk, z;
    a32_block(blk, z);                                               // rank 0 = the first key block (causal: the most query tiles)
    const int b = z / a.H, h = z - b * a.H;
    // this utterance's rows of the q-side / k-side tensors (AttnArgs::qoff / koff: ragged row offsets; default: the padded [B, L] layout)
    const int qrow0 = a.qoff ? a.qoff[b] : b * a.Lq, Lq = a.qoff ? a.qoff[b + 1] - qrow0 : a.Lq;
    const int krow0 = a.koff ? a.koff[b] : b * a.Lk, Lk = a.koff ? a.koff[b + 1] - krow0 : a.Lk;
    const int kb0 = blk * KB, kw0 = kb0 + wave * 32, key = kw0 + l31, kc = min(key, Lk - 1);
    int kend = Lk;
    if (a.mask_mode & 1) kend = min(kend, a.klen[b]);
    bf16_t* dko = reinterpret_cast<bf16_t*>(a.dk) + ((long)krow0 + kc) * a.lddk + h * DH;
    bf16_t* dvo = reinterpret_cast<bf16_t*>(a.dv) + ((long)krow0 + kc) * a.lddv + h * DH;
    const bool wlive = kw0 < kend;                                   // a wave of masked keys: zero gradients
    if (!wlive && key < Lk) { zero_own<DH>(dko, hi); zero_own<DH>(dvo, hi); }
    if (kb0 >= kend) return;
    const bool causal = a.mask_mode & 2;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q) + (long)qrow0 * a.ldq + h * DH;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(a.k) + (long)krow0 * a.ldk + h * DH;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(a.v) + (long)krow0 * a.ldv + h * DH;
    const bf16_t* dO = reinterpret_cast<const bf16_t*>(a.dout) + (long)qrow0 * a.ldo + h * DH;
    int qtiles = (Lq + 63) / 64;
    if (a.qskip) qtiles = min(qtiles, (a.qskip[b] + 63) / 64);       // tiles of padded query rows contribute nothing (d context = 0)
    const int qt0 = causal ? kb0 / 64 : 0;                           // queries before this key block never see it
    const int wqt0 = !wlive ? qtiles : (causal ? kw0 / 64 : 0);      // first query tile this wave computes
    Dma<DH, NW> dmq, dmo;
    dmq.init(Q, a.ldq, Lq, wave, lane); dmo.init(dO, a.ldo, Lq, wave, lane);
    float r_l = 0.f, r_d = 0.f;
    uint32_t r_s = 0;
    if (qt0 < qtiles) {
        dmq.issue(sQ + (qt0 & 1) * TILE, qt0 * 64, wave); dmo.issue(sO + (qt0 & 1) * TILE, qt0 * 64, wave);
        if (tid < 64) {
            const int qq = min(qt0 * 64 + tid, Lq - 1);
            sL[(qt0 & 1) * 64 + tid] = a.lse[(long)z * a.Lq + qq] * A32_LOG2E; sD[(qt0 & 1) * 64 + tid] = a.dsum[(long)z * a.Lq + qq];
            if (DROP) sS[(qt0 & 1) * 64 + tid] = b2s_wseed(a.drop, (uint32_t)((long)z * a.Lq + qq));
        }
    }
    bf16x8_t kf[NKS], vf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) { kf[ks] = frag_own(K, a.ldk, kc, ks, hi); vf[ks] = frag_own(V, a.ldv, kc, ks, hi); }
    const bool key_ok = key < kend;
    f32x16_t dk[NDT], dv[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
    const float sl2 = a.scale * A32_LOG2E, dscale = a.drop.scale;
    // dropout (b2s_common.h: b2s_keep_w): y = mix(seed(query row) + (key >> 2) GOLD); this key's multiplier and the shift that brings its field to the top
    const uint32_t xk = (uint32_t)(kc >> 2) * GOLD, wck = (kc & 2) ? B2S_WC1 : B2S_WC0, hsh = (kc & 1) ? 0u : 16u;
    const int ts32 = (int)((uint32_t)b2s_wthresh(a.drop) << 16);
    wait_vm0();
    __syncthreads();
    for (int qt = qt0; qt < qtiles; ++qt) {
        const int q0 = qt * 64, cur = qt & 1;
        const bf16_t* tQ = sQ + cur * TILE;
        const bf16_t* tO = sO + cur * TILE;
        const float* tL = sL + cur * 64;
        const float* tD = sD + cur * 64;
        const uint32_t* tS = sS + cur * 64;
        if (qt + 1 < qtiles) {
            dmq.issue(sQ + (cur ^ 1) * TILE, q0 + 64, wave); dmo.issue(sO + (cur ^ 1) * TILE, q0 + 64, wave);
            if (tid < 64) {
                const int qq = min(q0 + 64 + tid, Lq - 1);
                r_l = a.lse[(long)z * a.Lq + qq] * A32_LOG2E; r_d = a.dsum[(long)z * a.Lq + qq];
                if (DROP) r_s = b2s_wseed(a.drop, (uint32_t)((long)z * a.Lq + qq));
            }
        }
        if (qt >= wqt0) {
            // all 32 keys of this wave valid and visible to all 64 queries of the tile?  (wave-uniform)
            const bool interior = kw0 + 32 <= kend && q0 + 64 <= Lq && (!causal || kw0 + 31 <= q0);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16_t s, dp;                                      // s[r] = S[q = q0 + 32 t + crow(r, hi)][key = own]
                first1<DH>(s, tQ, t, kf, l31, hi);
                cfence();
                first1<DH>(dp, tO, t, vf, l31, hi);
                cfence();
                if (!interior) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int qq = q0 + t * 32 + crow(r, hi);
                        s[r] = (key_ok && qq < Lq && (!causal || key <= qq)) ? s[r] : -INFINITY;
                    }
                }
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4_t lq = *reinterpret_cast<const f32x4_t*>(tL + t * 32 + g4 * 8 + hi * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[g4 * 4 + j] = fast_exp2(fmaf(s[g4 * 4 + j], sl2, -lq[j]));
                }
                // in place: dp <- dS = P (dropped dP - D) (x a.scale at the end), s <- dropped P (x dscale at the end)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(tD + t * 32 + g4 * 8 + hi * 4);
                    u32x4_t sd4 = {0, 0, 0, 0};
                    if (DROP) sd4 = *reinterpret_cast<const u32x4_t*>(tS + t * 32 + g4 * 8 + hi * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = g4 * 4 + j;                     // query row q0 + 32 t + 8 g4 + 4 hi + j
                        bool keep = true;
                        if (DROP) {
                            const uint32_t x = sd4[j] + xk;
                            keep = (int)(((x ^ (x >> 16)) * wck) << hsh) >= ts32;
                        }
                        const float d = keep ? dp[r] * dscale : 0.f;
                        dp[r] = s[r] * (d - d4[j]);
                        s[r] = keep ? s[r] : 0.f;
                    }
                }
                cfence();
                second1<DH>(dv, tO, t, s, lane);                     // dV^T[d][key] += sum_q dO[q][d] Pd[q][key]
                cfence();
                second1<DH>(dk, tQ, t, dp, lane);                    // dK^T[d][key] += sum_q Q[q][d] dS[q][key]
                cfence();
            }
        }
        if (qt + 1 < qtiles && tid < 64) { sL[(cur ^ 1) * 64 + tid] = r_l; sD[(cur ^ 1) * 64 + tid] = r_d; if (DROP) sS[(cur ^ 1) * 64 + tid] = r_s; }
        wait_vm0();
        __syncthreads();
    }
    if (wlive && key < Lk) { store_own<DH>(dko, dk, a.scale, hi); store_own<DH>(dvo, dv, dscale, hi); }
}

template <int DH, bool DROP>
int launch32(const AttnArgs& a, int which, hipStream_t st) {
    constexpr int NW = 4;
    if (which == 0) {
        hipLaunchKernelGGL((attn32_fwd_kernel<DH, NW, DROP>), dim3(cdiv(a.Lq, NW * 32), a.B * a.H), dim3(NW * 64), 0, st, a);
    } else if (which == 1) {
        hipLaunchKernelGGL((attn32_dq_kernel<DH, NW, DROP>), dim3(cdiv(a.Lq, NW * 32), a.B * a.H), dim3(NW * 64), 0, st, a);
    } else {
        hipLaunchKernelGGL((attn32_dkv_kernel<DH, NW, DROP>), dim3(cdiv(a.Lk, NW * 32), a.B * a.H), dim3(NW * 64), 0, st, a);
    }
    B2S_LAUNCH_CHECK();
    return 0;
}
template <int DH>
int launch32_dh(const AttnArgs& a, int which, hipStream_t st) {
    // (rates below 2^-16 drop nothing under the 16-bit field rule)
    return (a.drop.thresh >> 16) != 0 ? launch32<DH, true>(a, which, st) : launch32<DH, false>(a, which, st);
}
}  // namespace

bool b2s_flash32_supported(int dh) { return dh == 32 || dh == 64 || dh == 96; }
int b2s_flash32_launch(const AttnArgs& a, int dh, int which, hipStream_t st) {
    B2S_CHECK(!a.ga_rows, "the 32x32 attention kernels do not carry the guided-attention term (it belongs to the encoder-decoder attention: attention.hip)");
    switch (dh) {
        case 32: return launch32_dh<32>(a, which, st);
        case 64: return launch32_dh<64>(a, which, st);
        case 96: return launch32_dh<96>(a, which, st);
    }
    return b2s_fail(__FILE__, __LINE__, "bf16 attention supports head sizes 32/64/96 (got %d)", dh);
}
