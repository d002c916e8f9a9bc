// Cluster-fused encoder sublayer kernels for gfx950 (see enc_fused.h for the decomposition).
// Reference: transformer/modules.py:49-69, transformer/attention.py:53-122, transformer/modules.py:8-20.
//
// Building blocks shared by the four "fat" kernels (512 threads = 8 waves, one workgroup per CU):
//   * streamed operands arrive through an LDS-DMA ring of 8 KB granules = [64 rows][64 k] bf16, K-contiguous in memory: one
//     global_load_lds_dwordx4 per wave and granule (8 rows x 128 B), source-side chunk swizzle (chunk ^= (row >> 1) & 7) so that the
//     ds_read_b128 fragment reads are conflict-free -- the image of gemm_glds256.hip's K-contiguous operand;
//   * a "step" is a fixed number of granules; steps are issued two ahead, a wave waits for ITS OWN DMAs with a counted vmcnt and one
//     s_barrier per step publishes the stage and frees the slot consumed before it;
//   * every product is computed SWAPPED (out^T = W . X^T: A fragment = weight rows, B fragment = token rows), so a lane ends up with
//     4 consecutive output features of ONE token: 8-byte LDS writes when the result is the next product's operand, 16-byte global stores
//     for the slab;
//   * operands produced on chip (q / k / v / ctx / f / dz / dq dk dv) live in padded row-major LDS tiles [128][K + 8];
//   * all LDS traffic while DMAs are in flight is inline asm: hipcc orders every LDS access it can see behind vmcnt(0) once an
//     LDS-DMA is outstanding.
#include <algorithm>
#include <mutex>
#include "enc_fused.h"

namespace encf {

typedef __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int NTHR = 512, GRAN = 8192;

// development aid (tools/encf_lab.hip): wall-clock (100 MHz) stamps at the phase boundaries of the fat kernels, kept in SGPRs and stored once
// at the end -- a store under a condition in the pipelined part would make hipcc drain the vector-memory queue there
#ifdef ENCF_STAMPS
__device__ unsigned long long* g_encf_stamp;
#define ENCF_T0 unsigned long long encf_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}; encf_t[0] = wall_clock64()
#define ENCF_T(i) encf_t[i] = wall_clock64()
#define ENCF_FLUSH() do { if (threadIdx.x == 0) for (int i_ = 0; i_ < 8; ++i_) g_encf_stamp[(size_t)blockIdx.x * 8 + i_] = encf_t[i_]; } while (0)
#else
#define ENCF_T0
#define ENCF_T(i)
#define ENCF_FLUSH()
#endif
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

__device__ __forceinline__ int swz_n(int r) { return (r >> 1) & 7; }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void dma16(const void* src, unsigned char* dst_wave) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst_wave, 16, 0, 0);
}
// ---- LDS addressing.  Every fragment address is  (per-lane base VGPR) + (compile-time offset):  the offset goes into the DS instruction's
// 16-bit immediate (what exceeds it is added to the base by one VALU op), so a kernel keeps a handful of address registers instead of one
// per (slot, granule, block, k-half) -- with computed addresses the two-slice FFN kernel spilled 70 dwords per lane.
//   swizzled granule, row = block*16 + li, chunk c = lg + 4*kh:  block*2048 + li*128 + ((c ^ swz(li)) << 4)   [swz(block*16 + li) = swz(li)]
//   padded tile [rows][LD], row = block*16 + li, k = ks*32 + lg*8:  block*32*LD + ks*64 + 2*li*LD + lg*16
constexpr int IMM_MAX = 65535;
#define ENCF_SPLIT(off) const unsigned hi_ = (unsigned)(off) & ~(unsigned)IMM_MAX; const int lo_ = (int)((unsigned)(off) & (unsigned)IMM_MAX)
__device__ __forceinline__ bf16x8_t lds_rd128(unsigned base, const int off) {
    ENCF_SPLIT(off);
    bf16x8_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base + hi_), "i"(lo_) : "memory");
    return v;
}
__device__ __forceinline__ f32x4_t lds_rd128f(unsigned base, const int off) {
    ENCF_SPLIT(off);
    f32x4_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base + hi_), "i"(lo_) : "memory");
    return v;
}
__device__ __forceinline__ void lds_wr64(unsigned base, const int off, uint32_t lo, uint32_t hi) {
    ENCF_SPLIT(off);
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t v = {lo, hi};
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(base + hi_), "v"(v), "i"(lo_) : "memory");
}
__device__ __forceinline__ void lds_wr128(unsigned base, const int off, bf16x8_t v) {
    ENCF_SPLIT(off);
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(base + hi_), "v"(v), "i"(lo_) : "memory");
}
__device__ __forceinline__ void lds_wr32(unsigned addr, float v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
// two transposing reads: rows (r0 + lg*4 + j) and (r0 + 16 + lg*4 + j), j = 0..3, column c0 + li of a row-major [rows][LD] bf16 tile at
// `off` (= attention.hip: frag_tr; matches a B fragment packed from two MFMA C blocks of 16 rows).  trb: per-lane base (tr_base)
template <int LD> __device__ __forceinline__ unsigned tr_base(unsigned L0, int li, int lg) { return L0 + 2u * (unsigned)((lg * 4 + (li >> 2)) * LD + (li & 3) * 4); }
template <int LD>
__device__ __forceinline__ void lds_tr_issue(bf16x4_t& lo, bf16x4_t& hi, unsigned trb, const int off, const int r0, const int c0) {
    const int o0 = off + 2 * (r0 * LD + c0), o1 = o0 + 2 * 16 * LD;
    { ENCF_SPLIT(o0); asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(trb + hi_), "i"(lo_) : "memory"); }
    { ENCF_SPLIT(o1); asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(trb + hi_), "i"(lo_) : "memory"); }
}
__device__ __forceinline__ bf16x8_t join8(bf16x4_t lo, bf16x4_t hi) {
    bf16x8_t r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
// the results of inline-asm LDS reads are valid after lgkmcnt(0); the empty asm statements tie every later use to the wait
template <typename V, int N> __device__ __forceinline__ void pin(V (&f)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(f[i]));
}
template <typename V> __device__ __forceinline__ void pin1(V& f) { asm volatile("" : "+v"(f)); }

#ifdef ENCF_LAB_NOMMA       // (lab: streaming time without the matrix pipe)
__device__ __forceinline__ f32x4_t mma(bf16x8_t a, bf16x8_t b, f32x4_t c) { c[0] += (float)a[0] * (float)b[0]; return c; }
#else
__device__ __forceinline__ f32x4_t mma(bf16x8_t a, bf16x8_t b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
#endif
__device__ __forceinline__ bf16x8_t pack8(const f32x4_t& a, const f32x4_t& b) {
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
    const u32x4_t u = {f2bf2(a[0], a[1]), f2bf2(a[2], a[3]), f2bf2(b[0], b[1]), f2bf2(b[2], b[3])};
    return __builtin_bit_cast(bf16x8_t, u);
}
__device__ __forceinline__ float group_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float group_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }
__device__ __forceinline__ float frag_dot(bf16x8_t x, bf16x8_t y) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += bf2f((bf16_t)x[e]) * bf2f((bf16_t)y[e]);
    return s;
}
// slab stores: 4 consecutive features of one token
#ifdef ENCF_LAB_NOSTORE      // (lab: keep the L2s warm between launches -- one lane still stores, so nothing is optimised away)
#define ENCF_ST_OK (threadIdx.x == 1023)
#else
#define ENCF_ST_OK true
#endif
__device__ __forceinline__ void slab_st(float* p, const f32x4_t& v) { if (ENCF_ST_OK) *reinterpret_cast<f32x4_t*>(p) = v; }
__device__ __forceinline__ void slab_st(bf16_t* p, const f32x4_t& v) {
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t u = {f2bf2(v[0], v[1]), f2bf2(v[2], v[3])};
    if (ENCF_ST_OK) *reinterpret_cast<u32x2_t*>(p) = u;
}
// fragment address inside a swizzled granule: row r (0..63), 16-byte chunk c (0..7)
__device__ __forceinline__ unsigned gaddr(int r, int c) { return (unsigned)(r * 128 + ((c ^ swz_n(r)) << 4)); }
constexpr int BLK = 2048;                    // 16 rows of a granule

// ================================================================================================ FFN sublayer (forward and backward)
// workgroup (utterance b, slice pair j0 in 0..7): for the hidden slices j = j0 and j0 + 8 (128 units each)
//     T1 = E(X Wa[slice]^T) [128 x 128]  ->  acc += T1 Wb[:, slice]^T [128 x 512];       slab_j0 = acc
//   forward : X = LN(x), Wa = W1, Wb = W2, E = dropout(relu(.)), T1 -> f
//   backward: X = dY,    Wa = W2^T, Wb = W1^T, E = . * (f > 0) / (1 - p), T1 -> dz
// One DMA stream of 2 x 12 steps x 4 granules: per slice, steps 0..7 = (X k-step: 2 granules, Wa k-step: 2 granules), steps 8..11 = Wb row
// chunk of 128 output features x the slice's 128 k (4 granules: k-half major); 3-stage ring (96 KB) + the T1 tile (34 KB).  Two slices per
// workgroup halve the slab traffic (8 instead of 16 partial outputs per row: the row kernels that sum them are bound by exactly that) and
// spend the fixed cost of a workgroup -- ~3.5 us until its first stage has landed, ~3.5 us of slab stores -- once per 2 x 8.6 us of streaming.
constexpr unsigned FFN_RING = 3 * 4 * GRAN, FFN_TLD = HS + 8, FFN_SMEM = FFN_RING + 128 * FFN_TLD * 2;
constexpr int FFN_NST = 24;
template <bool BWD, typename ST>
__global__ __launch_bounds__(NTHR, 2) void k_encf_ffn(EncfFfn a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lg = lane >> 4;
    ENCF_T0;
    // slice pair j0 of every utterance on XCD j0: an XCD's L2 holds 2 weight slices (512 KB) and the rows of all utterances
    const int wg = blockIdx.x, j0 = wg & 7, b = wg >> 3;
    const int S = a.S;
    const long row0 = (long)b * S, M = (long)a.B * S;
    const unsigned L0 = (unsigned)(uintptr_t)(lptr_t)smem;
    const int drow = wave * 8 + (lane >> 3), dsw = (((lane & 7) ^ swz_n(drow)) << 3);
    // DMA sources = wave-uniform base (SGPRs, advanced per step) + a 32-bit per-lane byte offset fixed for the launch
    const unsigned ox0 = 2u * (unsigned)(min(drow, S - 1) * D + dsw), ox1 = 2u * (unsigned)(min(64 + drow, S - 1) * D + dsw);
    const unsigned owa = 2u * (unsigned)(drow * D + dsw), owb = 2u * (unsigned)(drow * FF + dsw);
    const char* xb = reinterpret_cast<const char*>(a.X + row0 * D);
    const char* wab = reinterpret_cast<const char*>(a.Wa + (long)j0 * HS * D);              // (+ 8 * HS rows for the second slice)
    const char* wbb = reinterpret_cast<const char*>(a.Wb + j0 * HS);                          // (+ 8 * HS columns)
    const int nt0 = 2 * (wave & 3), mt0 = 4 * (wave >> 2);
    // fragment bases: weight blocks nt0 + i, token blocks mt0 + t (block index * BLK spans the two granules of an operand)
    const unsigned g0 = L0 + gaddr(li, lg), g1 = L0 + gaddr(li, lg + 4);
    const unsigned an[2] = {g0 + nt0 * BLK, g1 + nt0 * BLK}, am[2] = {g0 + mt0 * BLK, g1 + mt0 * BLK};
    const unsigned at = L0 + FFN_RING + 2u * (unsigned)((mt0 * 16 + li) * FFN_TLD) + lg * 16;          // T1 rows of the token blocks, k = lg*8
    const unsigned aw = L0 + FFN_RING + 2u * (unsigned)((mt0 * 16 + li) * FFN_TLD + nt0 * 16 + lg * 4);  // T1 write: token row, features nt0*16 + lg*4
    // backward: the relu / dropout mask of this lane's 8 T1 blocks of a slice, fetched at the start of the slice (used 7 steps later)
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    u32x2_t mk[2][4];
    auto load_mask = [&](int sl) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int m = min((mt0 + t) * 16 + li, S - 1);
                mk[i][t] = *reinterpret_cast<const u32x2_t*>(a.F + (row0 + m) * FF + (j0 + 8 * sl) * HS + (nt0 + i) * 16 + lg * 4);
            }
    };
    if (BWD) load_mask(0);
    auto issue = [&](int g) {
        unsigned char* dst = smem + (g % 3) * 4 * GRAN + wave * 1024;
        const int sl = g / 12, s = g % 12;
        if (s < 8) {
            const char* x = xb + s * 128;
            const char* w = wab + (long)sl * 8 * HS * D * 2 + s * 128;
            dma16(x + ox0, dst); dma16(x + ox1, dst + GRAN); dma16(w + owa, dst + 2 * GRAN); dma16(w + 64L * D * 2 + owa, dst + 3 * GRAN);
        } else {
            const char* w = wbb + sl * 8 * HS * 2 + (long)(s - 8) * 128 * FF * 2;
            dma16(w + owb, dst); dma16(w + 64L * FF * 2 + owb, dst + GRAN); dma16(w + 128 + owb, dst + 2 * GRAN); dma16(w + 64L * FF * 2 + 128 + owb, dst + 3 * GRAN);
        }
    };
    issue(0); issue(1);
    f32x4_t acc1[2][4], acc2[4][2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc2[c][i][t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
    bf16_t* fo = BWD ? a.dz : a.F;
    // T1 tile -> HBM, 4 x 16 bytes per thread.  Rows past the utterance re-write its last row with the same bytes: unconditional stores
    // (a store under a condition in the middle of the stream would drain the vector-memory queue)
    auto store_t1 = [&](int sl) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + it * NTHR, row = min(idx >> 4, S - 1), ch = idx & 15;
            bf16x8_t v = lds_rd128(L0 + 2u * (unsigned)(row * FFN_TLD + ch * 8), FFN_RING);
            wait_lgkm0(); pin1(v);
            *reinterpret_cast<bf16x8_t*>(fo + (row0 + row) * FF + (j0 + 8 * sl) * HS + ch * 8) = v;
        }
    };
    // (one slice = 12 steps; called twice below instead of looped, so that everything indexed by the slice is a compile-time constant)
    auto run_slice = [&](const int sl) {
#pragma unroll
    for (int s = 0; s < 12; ++s) {
        const int g = sl * 12 + s;
        if (g + 1 < FFN_NST) wait_vm<4>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (g == 0) { ENCF_T(1); }
        if (g == 8) { ENCF_T(2); }
        const int so = (s % 3) * 4 * GRAN;                                  // (12 % 3 == 0: the slot of step g is s % 3)
        if (s < 8) {
            if (s == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc1[i][t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                bf16x8_t fa[2], fb[4];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = lds_rd128(an[kh], so + 2 * GRAN + i * BLK);
#pragma unroll
                for (int t = 0; t < 4; ++t) fb[t] = lds_rd128(am[kh], so + t * BLK);
                wait_lgkm0(); pin(fa); pin(fb);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc1[i][t] = mma(fa[i], fb[t], acc1[i][t]);
            }
            if (s == 7) {
                // E(.) and the T1 tile: lane holds features (nt0+i)*16 + lg*4 + r of token (mt0+t)*16 + li.  (Every wave finished reading
                // the previous slice's tile before the barrier of its last step, 8 barriers ago.)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int m = (mt0 + t) * 16 + li, n = (nt0 + i) * 16 + lg * 4;
                        float v[4] = {acc1[i][t][0], acc1[i][t][1], acc1[i][t][2], acc1[i][t][3]};
                        if (!BWD) {
                            const uint32_t idx = (uint32_t)((row0 + m) * FF + (j0 + 8 * sl) * HS + n);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                v[r] = fmaxf(v[r], 0.f);
                                if (a.dhid.thresh) v[r] = b2s_keep(a.dhid, idx + r) ? v[r] * a.dhid.scale : 0.f;
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const uint32_t bits = (mk[i][t][r >> 1] >> ((r & 1) * 16)) & 0xffffu;       // bf16 > 0 <=> non-zero, sign clear
                                v[r] = (bits != 0 && !(bits & 0x8000u)) ? v[r] * a.aux_scale : 0.f;
                            }
                        }
                        lds_wr64(aw, 2 * (t * 16 * (int)FFN_TLD + i * 16), f2bf2(v[0], v[1]), f2bf2(v[2], v[3]));
                    }
                wait_lgkm0();                       // the tile is complete before this wave reaches the next barrier
                if (BWD && sl == 0) load_mask(1);
            }
        } else {
            const int nc = s - 8;
            if (nc == 0 && sl == 0) store_t1(0);    // (the first slice's tile leaves under its own second product; the last one's at the end)
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                const int kt2 = kq >> 1, kh = kq & 1;
                bf16x8_t fa[2], fb[4];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = lds_rd128(an[kh], so + kt2 * 2 * GRAN + i * BLK);
#pragma unroll
                for (int t = 0; t < 4; ++t) fb[t] = lds_rd128(at, 2 * (t * 16 * (int)FFN_TLD) + kq * 64);
                wait_lgkm0(); pin(fa); pin(fb);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc2[nc][i][t] = mma(fa[i], fb[t], acc2[nc][i][t]);
            }
        }
        // the refill of the slot freed by this step's barrier is issued BEHIND the step's MFMAs: an LDS-DMA instruction blocks its wave
        // until the texture path accepts it (8 waves x 4 instructions ~ 500 clocks per step), which then overlaps the matrix pipe's work
        // instead of preceding it
        if (g + 2 < FFN_NST) issue(g + 2);
    }
    };
    run_slice(0);
    run_slice(1);
    ENCF_T(3);
    // ---- stores (nothing is in flight any more)
    ST* slab = reinterpret_cast<ST*>(a.slabs) + (long)j0 * M * D;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int m = (mt0 + t) * 16 + li;
        if (m < S) {
#pragma unroll
            for (int nc = 0; nc < 4; ++nc)
#pragma unroll
                for (int i = 0; i < 2; ++i) slab_st(slab + (row0 + m) * D + nc * 128 + (nt0 + i) * 16 + lg * 4, acc2[nc][i][t]);
        }
    }
    store_t1(1);
    ENCF_T(4);
    ENCF_FLUSH();
}

// ================================================================================================ attention sublayer, forward
// workgroup (utterance b, head h):  [q k v] = X Wqkv[head rows]^T (K = 512)  ->  softmax(q k^T / 8 + key mask) v on chip  ->
// slab_h = ctx_h Wo[:, head cols]^T.   LDS: phase 1 ring 3 x 5 granules (120 KB); afterwards q / k / v / ctx tiles [128][72] (72 KB)
// + the 8 granules of the head's Wo columns (64 KB).
constexpr unsigned AT_LD = DH + 8, AT_TILE = 128 * AT_LD * 2, AT_BLK = 16 * AT_LD * 2;
constexpr unsigned AF_SQ = 0, AF_SK = AT_TILE, AF_SV = 2 * AT_TILE, AF_SC = 3 * AT_TILE, AF_WO = 4 * AT_TILE, AF_SMEM = AF_WO + 8 * GRAN;
static_assert(AF_SMEM >= 3 * 5 * GRAN && AF_SMEM <= 160 * 1024, "attention forward LDS plan");
template <typename ST>
__global__ __launch_bounds__(NTHR, 2) void k_encf_attn_fwd(EncfAttnFwd a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lg = lane >> 4;
    ENCF_T0;
    const int wg = blockIdx.x, h = wg & 7, b = wg >> 3;        // head h on XCD h: its L2 holds one head's weights + all utterances' rows
    const int S = a.S;
    const long row0 = (long)b * S, M = (long)a.B * S;
    const unsigned L0 = (unsigned)(uintptr_t)(lptr_t)smem;
    const int drow = wave * 8 + (lane >> 3), dsw = (((lane & 7) ^ swz_n(drow)) << 3);
    const unsigned ox0 = 2u * (unsigned)(min(drow, S - 1) * D + dsw), ox1 = 2u * (unsigned)(min(64 + drow, S - 1) * D + dsw);
    const unsigned ow = 2u * (unsigned)(drow * D + dsw);
    const char* xb = reinterpret_cast<const char*>(a.hN + row0 * D);
    const char* wqb = reinterpret_cast<const char*>(a.Wqkv + (long)h * DH * D);
    auto issue = [&](int s) {
        unsigned char* dst = smem + (s % 3) * 5 * GRAN + wave * 1024;
        const char* x = xb + s * 128;
        const char* w = wqb + s * 128;
        dma16(x + ox0, dst); dma16(x + ox1, dst + GRAN);
        dma16(w + ow, dst + 2 * GRAN); dma16(w + 512L * D * 2 + ow, dst + 3 * GRAN); dma16(w + 1024L * D * 2 + ow, dst + 4 * GRAN);
    };
    issue(0); issue(1);
    const int ng = wave & 3, mh = wave >> 2;
    const unsigned g0 = L0 + gaddr(li, lg), g1 = L0 + gaddr(li, lg + 4);
    const unsigned an[2] = {g0 + ng * 3 * BLK, g1 + ng * 3 * BLK}, am[2] = {g0 + mh * 4 * BLK, g1 + mh * 4 * BLK};
    f32x4_t acc[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[i][t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (s + 1 < 8) wait_vm<5>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (s == 0) { ENCF_T(1); }
        const int so = (s % 3) * 5 * GRAN;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            bf16x8_t fa[3], fb[4];
#pragma unroll
            for (int i = 0; i < 3; ++i) fa[i] = lds_rd128(an[kh], so + 2 * GRAN + i * BLK);
#pragma unroll
            for (int t = 0; t < 4; ++t) fb[t] = lds_rd128(am[kh], so + t * BLK);
            wait_lgkm0(); pin(fa); pin(fb);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[i][t] = mma(fa[i], fb[t], acc[i][t]);
        }
        if (s + 2 < 8) issue(s + 2);                     // (behind the MFMAs: see k_encf_ffn)
    }
    __builtin_amdgcn_s_barrier();                        // every wave is done with the ring: its space becomes the q / k / v / ctx tiles
    ENCF_T(2);
    // per-lane bases of the padded tiles: own rows (token block = wave) and tile-wide (block in the offset)
    const unsigned pt = L0 + 2u * (unsigned)(li * AT_LD) + lg * 16, po = pt + wave * AT_BLK;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int n = ng * 48 + i * 16 + lg * 4, part = n >> 6, d = n & 63;
        const unsigned wb_ = L0 + part * AT_TILE + 2u * (unsigned)((mh * 64 + li) * AT_LD + d);
#pragma unroll
        for (int t = 0; t < 4; ++t) lds_wr64(wb_, t * (int)AT_BLK, f2bf2(acc[i][t][0], acc[i][t][1]), f2bf2(acc[i][t][2], acc[i][t][3]));
    }
    {   // the head's 64 input columns of the output projection: 8 granules of 64 output rows
        const char* wo = reinterpret_cast<const char*>(a.Wo + h * DH);
#pragma unroll
        for (int g = 0; g < 8; ++g) dma16(wo + (long)g * 64 * D * 2 + ow, smem + AF_WO + g * GRAN + wave * 1024);
    }
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();
    // ---- attention core: wave w owns queries w*16 .. w*16+15 (lane: query li, 4 keys per 16-key block)
    const int mq = wave * 16 + li;
    const int kend = min(S, a.klen[b]);
    const int z = b * NH + h;
    const unsigned trb = tr_base<AT_LD>(L0, li, lg);
    float lsum, mref;
    {
        bf16x8_t qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks] = lds_rd128(po, AF_SQ + ks * 64);
        f32x4_t s[8];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            bf16x8_t kf[8];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) kf[t * 2 + ks] = lds_rd128(pt, AF_SK + (hf * 4 + t) * (int)AT_BLK + ks * 64);
            wait_lgkm0(); pin(qf); pin(kf);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                s[hf * 4 + t] = mma(kf[t * 2], qf[0], (f32x4_t){0.f, 0.f, 0.f, 0.f});
                s[hf * 4 + t] = mma(kf[t * 2 + 1], qf[1], s[hf * 4 + t]);
            }
        }
        const float sl2 = 0.125f * LOG2E;                // 1 / sqrt(64), logits in the log2 domain
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = t * 16 + lg * 4 + r;
                s[t][r] = key < kend ? s[t][r] : -INFINITY;
                mx = fmaxf(mx, s[t][r]);
            }
        mx = group_max(mx);
        mref = mx == -INFINITY ? 0.f : mx * sl2;
        lsum = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float p = __builtin_amdgcn_exp2f(fmaf(s[t][r], sl2, -mref)); lsum += p; s[t][r] = p; }
        lsum = group_sum(lsum);
        if (a.datt.thresh) {
            const uint32_t dseed_ = b2s_wseed(a.datt, (uint32_t)((long)z * S + mq));
            const int dth_ = b2s_wthresh(a.datt);
#pragma unroll
            for (int t = 0; t < 8; ++t) {                                // keys t 16 + lg 4 + {0, 1, 2, 3}: one quad of the row (b2s_common.h: b2s_keep_w)
                const uint32_t y = b2s_wmix(dseed_, (uint32_t)(t * 4 + lg)), w0 = y * B2S_WC0, w1 = y * B2S_WC1;
                s[t][0] = (int)(int16_t)(uint16_t)w0 >= dth_ ? s[t][0] * a.datt.scale : 0.f;
                s[t][1] = (int)(int16_t)(uint16_t)(w0 >> 16) >= dth_ ? s[t][1] * a.datt.scale : 0.f;
                s[t][2] = (int)(int16_t)(uint16_t)w1 >= dth_ ? s[t][2] * a.datt.scale : 0.f;
                s[t][3] = (int)(int16_t)(uint16_t)(w1 >> 16) >= dth_ ? s[t][3] * a.datt.scale : 0.f;
            }
        }
        f32x4_t o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const bf16x8_t bp = pack8(s[2 * kb], s[2 * kb + 1]);
            bf16x4_t lo[4], hi[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) lds_tr_issue<AT_LD>(lo[dt], hi[dt], trb, AF_SV, kb * 32, dt * 16);
            wait_lgkm0(); pin(lo); pin(hi);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = mma(join8(lo[dt], hi[dt]), bp, o[dt]);
        }
        const float inv = 1.f / lsum;
        const unsigned cw = L0 + AF_SC + 2u * (unsigned)(mq * AT_LD + lg * 4);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            lds_wr64(cw, dt * 32, f2bf2(o[dt][0] * inv, o[dt][1] * inv), f2bf2(o[dt][2] * inv, o[dt][3] * inv));
    }
    if (lg == 0 && mq < S) a.lse[(long)z * S + mq] = (mref + __log2f(lsum)) * LN2;
    ENCF_T(3);
    wait_vm<0>();                                        // (the Wo granules; the lse store)
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();
    ENCF_T(4);
    // ---- output projection: wave w owns output features w*64 .. w*64+63 (granule w), all 128 tokens
    f32x4_t oacc[4][8];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) oacc[nt][mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    {
        const unsigned aw[2] = {g0 + wave * GRAN, g1 + wave * GRAN};
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            bf16x8_t fa[4], fb[8];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) fa[nt] = lds_rd128(aw[kh], AF_WO + nt * BLK);
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) fb[mt] = lds_rd128(pt, AF_SC + mt * (int)AT_BLK + kh * 64);
            wait_lgkm0(); pin(fa); pin(fb);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) oacc[nt][mt] = mma(fa[nt], fb[mt], oacc[nt][mt]);
        }
    }
    ENCF_T(5);
    ST* slab = reinterpret_cast<ST*>(a.slabs) + (long)h * M * D;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const int m = mt * 16 + li;
        if (m < S) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) slab_st(slab + (row0 + m) * D + wave * 64 + nt * 16 + lg * 4, oacc[nt][mt]);
        }
    }
    // q / k / v and ctx of this head -> HBM (the weight-gradient GEMMs and the backward read them)
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = tid + it * NTHR, part = idx >> 10, row = (idx >> 3) & 127, ch = idx & 7;       // part 0..2: q k v ; 3: ctx
        bf16x8_t v = lds_rd128(L0 + part * AT_TILE + 2u * (unsigned)(row * AT_LD + ch * 8), 0);
        wait_lgkm0(); pin1(v);
        if (row < S) {
            bf16_t* dst = part < 3 ? a.qkv + (row0 + row) * (3 * D) + part * D + h * DH + ch * 8 : a.ctx + (row0 + row) * D + h * DH + ch * 8;
            *reinterpret_cast<bf16x8_t*>(dst) = v;
        }
    }
    ENCF_T(6);
    ENCF_FLUSH();
}

// ================================================================================================ attention sublayer, backward
// workgroup (b, h):  dO = dY Wo[:, head] (K = 512, Wo^T rows are K-contiguous)  ->  attention backward on chip (dq: per query wave;
// dk, dv: per key wave; P recomputed from the saved log-sum-exp)  ->  slab_h = [dq dk dv] Wqkv[head rows] (K = 192, Wqkv^T rows).
// LDS: q / k / v / dO tiles (72 KB) + lse / D rows (1 KB) + phase-1 ring 3 x 3 granules (72 KB); after the core the [dq dk dv] tile
// [128][200] takes the tiles' place and the phase-3 ring (3 x 4 granules) follows it.
constexpr unsigned AB_SQ = 0, AB_SK = AT_TILE, AB_SV = 2 * AT_TILE, AB_SDO = 3 * AT_TILE, AB_SL = 4 * AT_TILE, AB_SD = AB_SL + 512, AB_SS = AB_SD + 512, AB_R1 = AB_SS + 512;
constexpr unsigned AB_XLD = 3 * DH + 8, AB_X = 0, AB_R3 = 128 * AB_XLD * 2, AB_XBLK = 16 * AB_XLD * 2;
constexpr unsigned AB_SMEM = (AB_R1 + 3 * 3 * GRAN) > (AB_R3 + 3 * 4 * GRAN) ? (AB_R1 + 3 * 3 * GRAN) : (AB_R3 + 3 * 4 * GRAN);
static_assert(AB_SMEM <= 160 * 1024, "attention backward LDS plan");
static_assert(AB_R3 + 4 * GRAN >= AB_R1, "phase-3 slots 1, 2 must not overlap the q / k / v / dO tiles (they are filled during the core)");
template <typename ST>
__global__ __launch_bounds__(NTHR, 2) void k_encf_attn_bwd(EncfAttnBwd a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lg = lane >> 4;
    ENCF_T0;
    const int wg = blockIdx.x, h = wg & 7, b = wg >> 3;
    const int S = a.S;
    const long row0 = (long)b * S, M = (long)a.B * S;
    const int z = b * NH + h;
    const unsigned L0 = (unsigned)(uintptr_t)(lptr_t)smem;
    const int drow = wave * 8 + (lane >> 3), dsw = (((lane & 7) ^ swz_n(drow)) << 3);
    // saved q / k / v of the head -> registers (stored to LDS behind the first DMA issues)
    bf16x8_t qkv_r[6];
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int idx = tid + it * NTHR, part = idx >> 10, row = (idx >> 3) & 127, ch = idx & 7;
        qkv_r[it] = *reinterpret_cast<const bf16x8_t*>(a.qkv + (row0 + min(row, S - 1)) * (3 * D) + part * D + h * DH + ch * 8);
    }
    const int mq = wave * 16 + li, mqc = min(mq, S - 1);
    bf16x8_t of[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) of[ks] = *reinterpret_cast<const bf16x8_t*>(a.ctx + (row0 + mqc) * D + h * DH + ks * 32 + lg * 8);
    const float lse2 = a.lse[(long)z * S + mqc] * LOG2E;
    const int kend = min(S, a.klen[b]);
    const unsigned ox0 = 2u * (unsigned)(min(drow, S - 1) * D + dsw), ox1 = 2u * (unsigned)(min(64 + drow, S - 1) * D + dsw);
    const unsigned ow = 2u * (unsigned)(drow * D + dsw), ow3 = 2u * (unsigned)(drow * 3 * D + dsw);
    const char* yb = reinterpret_cast<const char*>(a.dY + row0 * D);
    const char* wotb = reinterpret_cast<const char*>(a.WoT + (long)h * DH * D);
    auto issue1 = [&](int s) {
        unsigned char* dst = smem + AB_R1 + (s % 3) * 3 * GRAN + wave * 1024;
        dma16(yb + s * 128 + ox0, dst); dma16(yb + s * 128 + ox1, dst + GRAN); dma16(wotb + s * 128 + ow, dst + 2 * GRAN);
    };
    issue1(0); issue1(1);
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int idx = tid + it * NTHR, part = idx >> 10, row = (idx >> 3) & 127, ch = idx & 7;
        lds_wr128(L0 + part * AT_TILE + 2u * (unsigned)(row * AT_LD + ch * 8), 0, qkv_r[it]);
    }
    // ---- phase 1: dO^T [64 x 128]: wave = (feature block nt, token half mh)
    const int nt1 = wave & 3, mh = wave >> 2;
    const unsigned g0 = L0 + gaddr(li, lg), g1 = L0 + gaddr(li, lg + 4);
    f32x4_t acc1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc1[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    {
        const unsigned an[2] = {g0 + nt1 * BLK, g1 + nt1 * BLK}, am[2] = {g0 + mh * 4 * BLK, g1 + mh * 4 * BLK};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s + 1 < 8) wait_vm<3>(); else wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            if (s == 0) { ENCF_T(1); }
            const int so = AB_R1 + (s % 3) * 3 * GRAN;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                bf16x8_t fa[1], fb[4];
                fa[0] = lds_rd128(an[kh], so + 2 * GRAN);
#pragma unroll
                for (int t = 0; t < 4; ++t) fb[t] = lds_rd128(am[kh], so + t * BLK);
                wait_lgkm0(); pin(fa); pin(fb);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc1[t] = mma(fa[0], fb[t], acc1[t]);
            }
            if (s + 2 < 8) issue1(s + 2);
        }
    }
    {
        const unsigned wb_ = L0 + AB_SDO + 2u * (unsigned)((mh * 64 + li) * AT_LD + nt1 * 16 + lg * 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) lds_wr64(wb_, t * (int)AT_BLK, f2bf2(acc1[t][0], acc1[t][1]), f2bf2(acc1[t][2], acc1[t][3]));
    }
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();                        // dO (and q / k / v) visible; the phase-1 ring is dead
    ENCF_T(2);
    // phase-3 stream: sub-step u = (part p, output half nh): Wqkv^T rows nh*256 + g*64 + row, k = p*512 + h*64 .. +63.  Slot of u = (u+1) % 3:
    // slots 1, 2 lie behind the tiles and are filled while the core runs; slot 0 overlaps them and is first used after the core.
    const char* wqtb = reinterpret_cast<const char*>(a.WqkvT + h * DH);
    auto issue3 = [&](int u) {
        unsigned char* dst = smem + AB_R3 + ((u + 1) % 3) * 4 * GRAN + wave * 1024;
        const char* w = wqtb + (long)((u & 1) * 256) * (3 * D) * 2 + (u >> 1) * D * 2;
#pragma unroll
        for (int g = 0; g < 4; ++g) dma16(w + (long)g * 64 * (3 * D) * 2 + ow3, dst + g * GRAN);
    };
    issue3(0); issue3(1);
    const float sl2 = 0.125f * LOG2E, scale = 0.125f;
    const unsigned pt = L0 + 2u * (unsigned)(li * AT_LD) + lg * 16, po = pt + wave * AT_BLK;
    const unsigned trb = tr_base<AT_LD>(L0, li, lg);
    f32x4_t dq[4], dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dq[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dk[dt] = dq[dt]; dv[dt] = dq[dt]; }
    {   // ---- role A: queries mq.  s[key][q], dp[key][q]; dS -> dq
        bf16x8_t qf[2], dof[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { qf[ks] = lds_rd128(po, AB_SQ + ks * 64); dof[ks] = lds_rd128(po, AB_SDO + ks * 64); }
        wait_lgkm0(); pin(qf); pin(dof);
        float Dq = group_sum(frag_dot(of[0], dof[0]) + frag_dot(of[1], dof[1]));
        const uint32_t dseed = b2s_wseed(a.datt, (uint32_t)((long)z * S + mq));        // dropout seed of this lane's weight row (role B reads it from LDS)
        const int dth = b2s_wthresh(a.datt);
        f32x4_t ds[8];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            bf16x8_t kf[8], vf[8];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    kf[t * 2 + ks] = lds_rd128(pt, AB_SK + (hf * 4 + t) * (int)AT_BLK + ks * 64);
                    vf[t * 2 + ks] = lds_rd128(pt, AB_SV + (hf * 4 + t) * (int)AT_BLK + ks * 64);
                }
            wait_lgkm0(); pin(kf); pin(vf);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4_t s = mma(kf[t * 2], qf[0], (f32x4_t){0.f, 0.f, 0.f, 0.f}); s = mma(kf[t * 2 + 1], qf[1], s);
                f32x4_t dp = mma(vf[t * 2], dof[0], (f32x4_t){0.f, 0.f, 0.f, 0.f}); dp = mma(vf[t * 2 + 1], dof[1], dp);
                if (a.datt.thresh) {                                     // keys (hf 4 + t) 16 + lg 4 + {0 .. 3}: one quad of this lane's row
                    const uint32_t y = b2s_wmix(dseed, (uint32_t)((hf * 4 + t) * 4 + lg)), w0 = y * B2S_WC0, w1 = y * B2S_WC1;
                    dp[0] = (int)(int16_t)(uint16_t)w0 >= dth ? dp[0] * a.datt.scale : 0.f;
                    dp[1] = (int)(int16_t)(uint16_t)(w0 >> 16) >= dth ? dp[1] * a.datt.scale : 0.f;
                    dp[2] = (int)(int16_t)(uint16_t)w1 >= dth ? dp[2] * a.datt.scale : 0.f;
                    dp[3] = (int)(int16_t)(uint16_t)(w1 >> 16) >= dth ? dp[3] * a.datt.scale : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = (hf * 4 + t) * 16 + lg * 4 + r;
                    const float p = key < kend ? __builtin_amdgcn_exp2f(fmaf(s[r], sl2, -lse2)) : 0.f;
                    s[r] = p * (dp[r] - Dq) * scale;
                }
                ds[hf * 4 + t] = s;
            }
        }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const bf16x8_t bp = pack8(ds[2 * kb], ds[2 * kb + 1]);
            bf16x4_t lo[4], hi[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) lds_tr_issue<AT_LD>(lo[dt], hi[dt], trb, AB_SK, kb * 32, dt * 16);
            wait_lgkm0(); pin(lo); pin(hi);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dq[dt] = mma(join8(lo[dt], hi[dt]), bp, dq[dt]);
        }
        if (lg == 0) { lds_wr32(L0 + AB_SL + mq * 4, lse2); lds_wr32(L0 + AB_SD + mq * 4, Dq); lds_wr32(L0 + AB_SS + mq * 4, __uint_as_float(dseed)); }
    }
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();                        // lse / D rows of all queries visible
    ENCF_T(3);
    {   // ---- role B: keys kk = wave*16 + li.  s[q][key], dp[q][key]; P^T dO -> dv, dS^T q -> dk
        const int kk = wave * 16 + li;
        const bool key_ok = kk < kend;
        // dropout (b2s_common.h: b2s_keep_w) with the row seeds role A left in LDS: this key's multiplier, its field brought to the top 16 bits
        const uint32_t wcB = (kk & 2) ? B2S_WC1 : B2S_WC0, hshB = (kk & 1) ? 0u : 16u;
        const int dthB = (int)((uint32_t)b2s_wthresh(a.datt) << 16);
        const unsigned ps = L0 + lg * 16;                   // lse / D rows: 4 consecutive queries lg*4 ..
        bf16x8_t kf[2], vf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { kf[ks] = lds_rd128(po, AB_SK + ks * 64); vf[ks] = lds_rd128(po, AB_SV + ks * 64); }
        wait_lgkm0(); pin(kf); pin(vf);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            f32x4_t pd[4], dsv[4];
            bf16x8_t qf[8], dof[8];
            f32x4_t lq[4], dd[4], sd[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    qf[t * 2 + ks] = lds_rd128(pt, AB_SQ + (hf * 4 + t) * (int)AT_BLK + ks * 64);
                    dof[t * 2 + ks] = lds_rd128(pt, AB_SDO + (hf * 4 + t) * (int)AT_BLK + ks * 64);
                }
                lq[t] = lds_rd128f(ps, AB_SL + (hf * 4 + t) * 64);
                dd[t] = lds_rd128f(ps, AB_SD + (hf * 4 + t) * 64);
                sd[t] = lds_rd128f(ps, AB_SS + (hf * 4 + t) * 64);
            }
            wait_lgkm0(); pin(qf); pin(dof); pin(lq); pin(dd); pin(sd);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4_t s = mma(qf[t * 2], kf[0], (f32x4_t){0.f, 0.f, 0.f, 0.f}); s = mma(qf[t * 2 + 1], kf[1], s);
                f32x4_t dp = mma(dof[t * 2], vf[0], (f32x4_t){0.f, 0.f, 0.f, 0.f}); dp = mma(dof[t * 2 + 1], vf[1], dp);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qq = (hf * 4 + t) * 16 + lg * 4 + r;
                    const float p = (key_ok && qq < S) ? __builtin_amdgcn_exp2f(fmaf(s[r], sl2, -lq[t][r])) : 0.f;
                    float d = dp[r], pp = p;
                    if (a.datt.thresh) {
                        const bool keep = (int)((b2s_wmix(__float_as_uint(sd[t][r]), (uint32_t)(kk >> 2)) * wcB) << hshB) >= dthB;
                        d = keep ? d * a.datt.scale : 0.f; pp = keep ? p * a.datt.scale : 0.f;
                    }
                    pd[t][r] = pp;
                    dsv[t][r] = p * (d - dd[t][r]) * scale;
                }
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const bf16x8_t bpd = pack8(pd[2 * kb], pd[2 * kb + 1]), bds = pack8(dsv[2 * kb], dsv[2 * kb + 1]);
                bf16x4_t lo[8], hi[8];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    lds_tr_issue<AT_LD>(lo[dt], hi[dt], trb, AB_SDO, hf * 64 + kb * 32, dt * 16);
                    lds_tr_issue<AT_LD>(lo[4 + dt], hi[4 + dt], trb, AB_SQ, hf * 64 + kb * 32, dt * 16);
                }
                wait_lgkm0(); pin(lo); pin(hi);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    dv[dt] = mma(join8(lo[dt], hi[dt]), bpd, dv[dt]);
                    dk[dt] = mma(join8(lo[4 + dt], hi[4 + dt]), bds, dk[dt]);
                }
            }
        }
    }
    __builtin_amdgcn_s_barrier();                        // every wave is done with the tiles: [dq dk dv] takes their place
    ENCF_T(4);
    {
        const unsigned xw = L0 + AB_X + 2u * (unsigned)((wave * 16 + li) * AB_XLD + lg * 4);       // dq: query row ; dk / dv: key row -- the same token index
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            lds_wr64(xw, dt * 32, f2bf2(dq[dt][0], dq[dt][1]), f2bf2(dq[dt][2], dq[dt][3]));
            lds_wr64(xw, dt * 32 + 2 * DH, f2bf2(dk[dt][0], dk[dt][1]), f2bf2(dk[dt][2], dk[dt][3]));
            lds_wr64(xw, dt * 32 + 4 * DH, f2bf2(dv[dt][0], dv[dt][1]), f2bf2(dv[dt][2], dv[dt][3]));
        }
    }
    wait_lgkm0();
    // ---- phase 3: slab^T [512 x 128]: sub-step (p, nh); wave = 2 feature blocks of granule wave >> 1 (blocks wave*2, wave*2 + 1), all 8 token blocks
    f32x4_t acc3[2][2][8];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) acc3[nh][i][mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    {
        const unsigned an[2] = {g0 + wave * 2 * BLK, g1 + wave * 2 * BLK};
        const unsigned px = L0 + AB_X + 2u * (unsigned)(li * AB_XLD) + lg * 16;
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            if (u + 1 < 6) wait_vm<4>(); else wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            const int so = AB_R3 + ((u + 1) % 3) * 4 * GRAN;
            const int p = u >> 1, nh = u & 1;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                bf16x8_t fa[2], fb[8];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = lds_rd128(an[kh], so + i * BLK);
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) fb[mt] = lds_rd128(px, mt * (int)AB_XBLK + p * 2 * DH + kh * 64);
                wait_lgkm0(); pin(fa); pin(fb);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int mt = 0; mt < 8; ++mt) acc3[nh][i][mt] = mma(fa[i], fb[mt], acc3[nh][i][mt]);
            }
            if (u + 2 < 6) issue3(u + 2);
        }
    }
    ENCF_T(5);
    ST* slab = reinterpret_cast<ST*>(a.slabs) + (long)h * M * D;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const int m = mt * 16 + li;
        if (m < S) {
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int i = 0; i < 2; ++i) slab_st(slab + (row0 + m) * D + nh * 256 + (wave * 2 + i) * 16 + lg * 4, acc3[nh][i][mt]);
        }
    }
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int idx = tid + it * NTHR, part = idx >> 10, row = (idx >> 3) & 127, ch = idx & 7;
        bf16x8_t v = lds_rd128(L0 + AB_X + 2u * (unsigned)(row * AB_XLD + part * DH + ch * 8), 0);
        wait_lgkm0(); pin1(v);
        if (row < S) *reinterpret_cast<bf16x8_t*>(a.dqkv + (row0 + row) * (3 * D) + part * D + h * DH + ch * 8) = v;
    }
    ENCF_T(6);
    ENCF_FLUSH();
}

// ================================================================================================ row kernels
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(bf2f(u.x & 0xffff), bf2f(u.x >> 16), bf2f(u.y & 0xffff), bf2f(u.y >> 16));
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
    uint2 u; u.x = f2bf2(v.x, v.y); u.y = f2bf2(v.z, v.w);
    *reinterpret_cast<uint2*>(p) = u;
}
__device__ __forceinline__ float4 drop4(float4 v, const DropCfg& d, uint32_t idx) {
    if (!d.thresh) return v;
    v.x = b2s_keep(d, idx) ? v.x * d.scale : 0.f; v.y = b2s_keep(d, idx + 1) ? v.y * d.scale : 0.f;
    v.z = b2s_keep(d, idx + 2) ? v.z * d.scale : 0.f; v.w = b2s_keep(d, idx + 3) ? v.w * d.scale : 0.f;
    return v;
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// x_out = x_in + dropout(sum_s slab_s) ; LayerNorm of x_out.  One wave per row (512 = 2 x 64 lanes x 4), slabs summed in slab order.
template <typename ST, int NS>
__global__ __launch_bounds__(256) void k_encf_rl_fwd(const float* __restrict__ x_in, const ST* __restrict__ slabs, long slab_stride, DropCfg dres,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ x_out,
                                                     bf16_t* __restrict__ h, int ldh, float* __restrict__ h32, int ldh32, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int M) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= M) return;
    float4 v[2], g[2], be[2], sl[2][NS];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = (lane + 64 * i) * 4;
        v[i] = ld4(x_in + (long)row * D + c); g[i] = ld4(gamma + c); be[i] = ld4(beta + c);
#pragma unroll
        for (int s = 0; s < NS; ++s) sl[i][s] = ld4(slabs + s * slab_stride + (long)row * D + c);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float4 acc = sl[i][0];
#pragma unroll
        for (int s = 1; s < NS; ++s) acc = add4(acc, sl[i][s]);
        v[i] = add4(v[i], drop4(acc, dres, (uint32_t)((long)row * D + (lane + 64 * i) * 4)));
        sum += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const float mu = wave_sum(sum) * (1.f / D);
    float qs = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float a0 = v[i].x - mu, a1 = v[i].y - mu, a2 = v[i].z - mu, a3 = v[i].w - mu;
        qs += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
    }
    const float rs = 1.f / sqrtf(wave_sum(qs) * (1.f / D) + 1e-6f);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = (lane + 64 * i) * 4;
        st4(x_out + (long)row * D + c, v[i]);
        float4 o;
        o.x = (v[i].x - mu) * rs * g[i].x + be[i].x; o.y = (v[i].y - mu) * rs * g[i].y + be[i].y;
        o.z = (v[i].z - mu) * rs * g[i].z + be[i].z; o.w = (v[i].w - mu) * rs * g[i].w + be[i].w;
        if (h) st4(h + (long)row * ldh + c, o);
        if (h32) st4(h32 + (long)row * ldh32 + c, o);
    }
}

// LayerNorm backward with dy = sum_s slab_s (rowops.hip: k_ln_bwd_fast with the slab sum as its first input): dx += LN'(dy), optional
// bf16(dropout(dx)) for the next sublayer of the backward pass, partial d gamma / d beta rows into ws (one row of 2*512 per workgroup)
template <int NS> struct RlRow { float4 d[2], xv[2], pv[2]; float mu, rs; };
template <typename ST, int NS, typename TX>
__device__ __forceinline__ void rl_row_load(RlRow<NS>& r, const ST* __restrict__ slabs, long slab_stride, const float* __restrict__ x, const TX* dx,
                                            const float* __restrict__ mean, const float* __restrict__ rstd, int row, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = (lane + 64 * i) * 4;
        float4 sl[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) sl[s] = ld4(slabs + s * slab_stride + (long)row * D + c);
        r.xv[i] = ld4(x + (long)row * D + c);
        r.pv[i] = ld4(dx + (long)row * D + c);
        float4 acc = sl[0];
#pragma unroll
        for (int s = 1; s < NS; ++s) acc = add4(acc, sl[s]);
        r.d[i] = acc;
    }
    r.mu = mean[row]; r.rs = rstd[row];
}
template <typename ST, int NS, bool DY2, typename TX>
__global__ __launch_bounds__(256) void k_encf_rl_bwd(const ST* __restrict__ slabs, long slab_stride, const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd, TX* dx, int M, float* __restrict__ ws,
                                                     bf16_t* __restrict__ dy2, DropCfg drop2) {
    __shared__ float sacc[4 * 2 * D];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4 pg[2], pb[2], gm[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { pg[i] = pb[i] = make_float4(0, 0, 0, 0); gm[i] = ld4(gamma + (lane + 64 * i) * 4); }
    const int nw = gridDim.x * 4;
    int row = blockIdx.x * 4 + wave;
    RlRow<NS> cur, nxt;
    rl_row_load<ST, NS, TX>(cur, slabs, slab_stride, x, dx, mean, rstd, min(row, M - 1), lane);
    for (; row < M; row += nw) {
        rl_row_load<ST, NS, TX>(nxt, slabs, slab_stride, x, dx, mean, rstd, min(row + nw, M - 1), lane);
        const float mu = cur.mu, rs = cur.rs;
        float4 g[2], xh[2];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float4 d = cur.d[i], xv = cur.xv[i];
            xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
            pb[i].x += d.x; pb[i].y += d.y; pb[i].z += d.z; pb[i].w += d.w;
            pg[i].x += d.x * xh[i].x; pg[i].y += d.y * xh[i].y; pg[i].z += d.z * xh[i].z; pg[i].w += d.w * xh[i].w;
            g[i] = make_float4(d.x * gm[i].x, d.y * gm[i].y, d.z * gm[i].z, d.w * gm[i].w);
            s1 += g[i].x + g[i].y + g[i].z + g[i].w;
            s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
        }
        s1 = wave_sum(s1) * (1.f / D); s2 = wave_sum(s2) * (1.f / D);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = (lane + 64 * i) * 4;
            float4 o;
            o.x = rs * (g[i].x - s1 - xh[i].x * s2) + cur.pv[i].x; o.y = rs * (g[i].y - s1 - xh[i].y * s2) + cur.pv[i].y;
            o.z = rs * (g[i].z - s1 - xh[i].z * s2) + cur.pv[i].z; o.w = rs * (g[i].w - s1 - xh[i].w * s2) + cur.pv[i].w;
            st4(dx + (long)row * D + c, o);
            if (DY2) st4(dy2 + (long)row * D + c, drop4(o, drop2, (uint32_t)((long)row * D + c)));
        }
        cur = nxt;
    }
    float* mine = sacc + wave * 2 * D;
#pragma unroll
    for (int i = 0; i < 2; ++i) { st4(mine + (lane + 64 * i) * 4, pg[i]); st4(mine + D + (lane + 64 * i) * 4, pb[i]); }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * D; i += 256) ws[(long)blockIdx.x * 2 * D + i] = sacc[i] + sacc[2 * D + i] + sacc[4 * D + i] + sacc[6 * D + i];
}

// dst[c][r] = src[r][c], 64 x 64 tiles through LDS, several matrices per launch (blockIdx.y = job)
constexpr int TR_MAX = 24;
struct TrBatch { int n; EncfTransposeJob j[TR_MAX]; };
__global__ __launch_bounds__(256) void k_encf_transpose(TrBatch bt) {
    __shared__ bf16_t tile[64][66];
    const EncfTransposeJob jb = bt.j[blockIdx.y];
    const int tc = jb.C / 64, ntile = (jb.R / 64) * tc;
    for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
        const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * 32; i += 256) {
            const int r = i >> 5, c = (i & 31) * 2;
            const uint32_t u = *reinterpret_cast<const uint32_t*>(jb.src + (long)(r0 + r) * jb.C + c0 + c);
            tile[r][c] = (bf16_t)(u & 0xffff); tile[r][c + 1] = (bf16_t)(u >> 16);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * 32; i += 256) {
            const int c = i >> 5, r = (i & 31) * 2;
            const uint32_t u = (uint32_t)tile[r][c] | ((uint32_t)tile[r + 1][c] << 16);
            *reinterpret_cast<uint32_t*>(jb.dst + (long)(c0 + c) * jb.R + r0 + r) = u;
        }
    }
}

template <typename K> int set_smem(K kernel, unsigned bytes, std::once_flag& once, hipError_t& err) {
    std::call_once(once, [&] { err = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); });
    B2S_HIP(err);
    return 0;
}
int check_shape(int B, int S) {
    B2S_CHECK(B > 0 && S > 0 && S <= MAXS, "fused encoder kernels: B=%d S=%d (need 1 <= S <= %d)", B, S, MAXS);
    return 0;
}

}  // namespace encf

bool b2s_encf_supported(int D, int H, int F, int S) { return D == encf::D && H == encf::NH && F == encf::FF && S >= 1 && S <= encf::MAXS; }

int b2s_encf_ffn(const EncfFfn& a, bool bwd, int slab_bf16, hipStream_t st) {
    using namespace encf;
    B2S_TRY(check_shape(a.B, a.S));
    B2S_CHECK(a.X && a.Wa && a.Wb && a.F && a.slabs && (!bwd || a.dz), "fused FFN: null argument");
    static std::once_flag once[4];
    static hipError_t err[4];
    const dim3 grid(a.B * 8), blk(NTHR);
#define B2S_FFN(BW, T, I) do { B2S_TRY(set_smem(k_encf_ffn<BW, T>, FFN_SMEM, once[I], err[I])); \
        hipLaunchKernelGGL((k_encf_ffn<BW, T>), grid, blk, FFN_SMEM, st, a); } while (0)
    if (bwd) { if (slab_bf16) B2S_FFN(true, bf16_t, 0); else B2S_FFN(true, float, 1); }
    else     { if (slab_bf16) B2S_FFN(false, bf16_t, 2); else B2S_FFN(false, float, 3); }
#undef B2S_FFN
    B2S_LAUNCH_CHECK();
    return 0;
}
int b2s_encf_attn_fwd(const EncfAttnFwd& a, int slab_bf16, hipStream_t st) {
    using namespace encf;
    B2S_TRY(check_shape(a.B, a.S));
    B2S_CHECK(a.hN && a.Wqkv && a.Wo && a.klen && a.qkv && a.ctx && a.lse && a.slabs, "fused attention forward: null argument");
    static std::once_flag once[2];
    static hipError_t err[2];
    const dim3 grid(a.B * NH), blk(NTHR);
    if (slab_bf16) { B2S_TRY(set_smem(k_encf_attn_fwd<bf16_t>, AF_SMEM, once[0], err[0])); hipLaunchKernelGGL((k_encf_attn_fwd<bf16_t>), grid, blk, AF_SMEM, st, a); }
    else           { B2S_TRY(set_smem(k_encf_attn_fwd<float>, AF_SMEM, once[1], err[1])); hipLaunchKernelGGL((k_encf_attn_fwd<float>), grid, blk, AF_SMEM, st, a); }
    B2S_LAUNCH_CHECK();
    return 0;
}
int b2s_encf_attn_bwd(const EncfAttnBwd& a, int slab_bf16, hipStream_t st) {
    using namespace encf;
    B2S_TRY(check_shape(a.B, a.S));
    B2S_CHECK(a.dY && a.qkv && a.ctx && a.lse && a.WoT && a.WqkvT && a.klen && a.dqkv && a.slabs, "fused attention backward: null argument");
    static std::once_flag once[2];
    static hipError_t err[2];
    const dim3 grid(a.B * NH), blk(NTHR);
    if (slab_bf16) { B2S_TRY(set_smem(k_encf_attn_bwd<bf16_t>, AB_SMEM, once[0], err[0])); hipLaunchKernelGGL((k_encf_attn_bwd<bf16_t>), grid, blk, AB_SMEM, st, a); }
    else           { B2S_TRY(set_smem(k_encf_attn_bwd<float>, AB_SMEM, once[1], err[1])); hipLaunchKernelGGL((k_encf_attn_bwd<float>), grid, blk, AB_SMEM, st, a); }
    B2S_LAUNCH_CHECK();
    return 0;
}

int b2s_encf_reduce_ln_fwd(const float* x_in, const void* slabs, int ns, int slab_bf16, DropCfg dres, const float* gamma, const float* beta,
                           float* x_out, bf16_t* h, int ldh, float* h32, int ldh32, float* mean, float* rstd, int M, hipStream_t st) {
    using namespace encf;
    B2S_CHECK(x_in && slabs && gamma && beta && x_out && mean && rstd && M > 0 && (ns == NH || ns == NSF), "fused reduce + LayerNorm: bad argument (ns = %d)", ns);
    const dim3 grid(cdiv(M, 4)), blk(256);
    const long ss = (long)M * D;
#define B2S_RL(T, NS_) hipLaunchKernelGGL((k_encf_rl_fwd<T, NS_>), grid, blk, 0, st, x_in, (const T*)slabs, ss, dres, gamma, beta, x_out, h, ldh, h32, ldh32, mean, rstd, M)
    static_assert(NH == 8 && NSF == 8, "both sublayers leave 8 slabs");
    if (slab_bf16) B2S_RL(bf16_t, 8); else B2S_RL(float, 8);
#undef B2S_RL
    B2S_LAUNCH_CHECK();
    return 0;
}
int b2s_encf_reduce_ln_bwd(const void* slabs, int ns, int slab_bf16, const float* x_in, const float* gamma, const float* mean, const float* rstd,
                           float* dx, float* ws, int* nblk, bf16_t* dy2, DropCfg drop2, int M, hipStream_t st, int dx_bf16) {
    using namespace encf;
    B2S_CHECK(slabs && x_in && gamma && mean && rstd && dx && ws && nblk && M > 0 && (ns == NH || ns == NSF), "fused reduce + LayerNorm backward: bad argument");
    const int grid = std::max(1, std::min(cdiv(M, 12), 768));          // (= rowops: ~3 rows per wave, at most RO_LN_WS_ROWS partial rows)
    const long ss = (long)M * D;
#define B2S_RLB2(T, NS_, DY, TX_) hipLaunchKernelGGL((k_encf_rl_bwd<T, NS_, DY, TX_>), dim3(grid), dim3(256), 0, st, (const T*)slabs, ss, x_in, gamma, mean, rstd, (TX_*)dx, M, ws, dy2, drop2)
#define B2S_RLB(T, NS_) do { if (dy2) { if (dx_bf16) B2S_RLB2(T, NS_, true, bf16_t); else B2S_RLB2(T, NS_, true, float); } \
                             else { if (dx_bf16) B2S_RLB2(T, NS_, false, bf16_t); else B2S_RLB2(T, NS_, false, float); } } while (0)
    if (slab_bf16) B2S_RLB(bf16_t, 8); else B2S_RLB(float, 8);
#undef B2S_RLB2
#undef B2S_RLB
    B2S_LAUNCH_CHECK();
    *nblk = grid;
    return 0;
}
int b2s_encf_transpose(const EncfTransposeJob* jobs, int n, hipStream_t st) {
    using namespace encf;
    B2S_CHECK(jobs && n >= 1 && n <= TR_MAX, "transpose: %d jobs (max %d)", n, TR_MAX);
    TrBatch bt; bt.n = n;
    for (int i = 0; i < n; ++i) {
        B2S_CHECK(jobs[i].src && jobs[i].dst && jobs[i].R % 64 == 0 && jobs[i].C % 64 == 0, "transpose: dims must be multiples of 64");
        bt.j[i] = jobs[i];
    }
    hipLaunchKernelGGL(k_encf_transpose, dim3(64, n), dim3(256), 0, st, bt);
    B2S_LAUNCH_CHECK();
    return 0;
}
