// Autoregressive decode step, fused: ONE kernel per decoder sublayer and frame (20 kernel nodes per frame instead of 76).
// Reference loop: synthesize.py:35-45 (one decoder pass per frame); modules.py:123-145 (the sublayers).
//
// Why: a frame is a chain of small dependent phases (LayerNorm -> projection -> attention -> projection ...), and with one
// kernel per phase the frame cost was the number of kernel boundaries (76 x ~9 us, profiles/r01_*), not the 0.9 GB it has to
// stream.  Here the work of a sublayer is cut so that no phase inside it needs another workgroup's result:
//   * the batch is cut into groups of UB utterances and the sublayer into NS slices -- a head for the attention sublayers,
//     1/NS of the hidden units for the FFN -- one workgroup per (slice, group), slice = blockIdx % NS so that with NS = 8
//     every XCD works on one slice and that slice's weights are fetched into one L2 only (HBM reads every weight once);
//   * a workgroup recomputes what is cheap (x = X + previous partial slabs, LayerNorm of its UB rows), runs its slice of the
//     input projection, the attention of its (utterance, head) pairs over the KV cache, and its slice's SHARE of the output
//     projection (K = dh or F/NS), which it stores as partial slab `slice`;
//   * the sum over slices is taken by the next kernel's workgroups in a fixed order (no atomics: frames are bit-reproducible,
//     eager == graph replay), the residual-dropout mask depends on (utterance, column, frame) only and is applied per slab.
// With UB rows per workgroup the projections are GEMVs.  bf16 at the default sizes: on the matrix pipe (16 weight rows x 32 k per
// MFMA, the UB utterances as columns) from fragment-packed weight copies, so that every load instruction is one contiguous KB, with
// the head's weight slice pulled into the XCD's L2 while LayerNorm runs (l2_warm); otherwise VALU dot products from an LDS-staged x.
// Attention over the cache is one pass (online softmax per key slot, merged once).  Bound: the HBM stream of the KV cache, the
// hand-off of the partial slabs between kernels (other XCDs' L2s) and the per-CU delivery of the weight slices.
#include <algorithm>
#include "decode_fused.h"

namespace {

template <typename T> struct DV;
template <> struct DV<float>  { static constexpr int VE = 4; };
template <> struct DV<bf16_t> { static constexpr int VE = 8; };

constexpr int UBA = 2;                   // utterances per workgroup: attention sublayers (4: the projections take as long -- they are bound per
                                         // CU, not by the L2 -- and the attention over the cache gets slower: 0.49 -> 0.56 ms per frame)
constexpr int UBH = 2;                   //                           prenet, heads
constexpr int UBF = 8;                   //                           FFN sublayer (twice the slices: every group re-reads a slice's weights
                                         //                           from L2, and that L2 -> CU traffic is what bounds the FFN kernel)
constexpr int NT = 512;                  // threads per workgroup (8 waves: 256 VGPRs each, enough to keep a whole K walk of loads in flight)
constexpr int HT = NT / UBA;             // threads that work on one utterance's attention

typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));

// Widths.  FAST instantiation: exactly the reference's default sizes (hyperparams.py:24-35 -- every released checkpoint): the step
// counts of the GEMV are compile-time and its load / consume sequences contain no control flow at all (hipcc waits vmcnt(0) at the
// join of every conditional block, even a workgroup-uniform one, which serialises the loads).  Generic instantiation: anything up
// to the maxima below, uniform conditions per step (correct, slower); wider models run the unfused path (b2s_df_supported).
constexpr int FD_D = 768, FD_DH = 96, FD_FS = 96, FD_HP = 256, FD_NM = 80;
constexpr int PN_SL = 8;                 // prenet: workgroups per utterance group at the default sizes
constexpr int DF_MAX_D = 768, DF_MAX_DH = 128, DF_MAX_FS = 256, DF_MAX_HP = 256, DF_MAX_NM = 128;
template <typename T, bool FAST> constexpr int KD_STEPS = (FAST ? FD_D : DF_MAX_D) / (8 * DV<T>::VE);       // K = D, 8 lanes per row
template <typename T, bool FAST> constexpr int FS_STEPS = (FAST ? FD_FS : DF_MAX_FS) / (8 * DV<T>::VE);     // K = F / NS, 8 lanes per row
template <typename T, bool FAST> constexpr int HP_STEPS = (FAST ? FD_HP : DF_MAX_HP) / (4 * DV<T>::VE);     // K = prenet width, 4 lanes per row
template <typename T, bool FAST> constexpr int DH_STEPS = (FAST ? FD_DH : DF_MAX_DH) / (4 * DV<T>::VE);     // K = head width, 4 lanes per row
template <typename T> constexpr int NM_LPR = 16 / DV<T>::VE;                                                // K = num_mels: 16 elements per step
template <bool FAST> constexpr int NM_STEPS = (FAST ? FD_NM : DF_MAX_NM) / 16;

template <typename T> __device__ __forceinline__ float dotv(const uint4& w, const uint4& a, float acc);
template <> __device__ __forceinline__ float dotv<float>(const uint4& w, const uint4& a, float acc) {
    acc = fmaf(__uint_as_float(w.x), __uint_as_float(a.x), acc);
    acc = fmaf(__uint_as_float(w.y), __uint_as_float(a.y), acc);
    acc = fmaf(__uint_as_float(w.z), __uint_as_float(a.z), acc);
    acc = fmaf(__uint_as_float(w.w), __uint_as_float(a.w), acc);
    return acc;
}
template <> __device__ __forceinline__ float dotv<bf16_t>(const uint4& w, const uint4& a, float acc) {
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, w.x), __builtin_bit_cast(bf2_t, a.x), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, w.y), __builtin_bit_cast(bf2_t, a.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, w.z), __builtin_bit_cast(bf2_t, a.z), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, w.w), __builtin_bit_cast(bf2_t, a.w), acc, false);
    return acc;
}
// value as the compute dtype would store it (bf16 mode: rounded to bf16, as the unfused path's intermediate tensors are)
template <typename T> __device__ __forceinline__ float rnd(float v);
template <> __device__ __forceinline__ float rnd<float>(float v) { return v; }
template <> __device__ __forceinline__ float rnd<bf16_t>(float v) { return bf2f(f2bf(v)); }

__device__ __forceinline__ DropCfg salted(DropCfg d, int t) {      // same per-frame key as the GEMM epilogues (gemm_epi.h: drop_salt)
    if (d.thresh) d.key ^= b2s_hash32((uint32_t)t * 2246822519u + 3266489917u);
    return d;
}

// out[r][j] = sum_k a[r][k] * W[row(j)][k]  for j < ncols, r < UB.  W: global, row stride ldw elements, K contiguous (16-byte aligned
// rows), row(j) = (j / blk_rows) * blk_stride + j % blk_rows (blk_rows >= ncols: plain); a: LDS, [UB][lda] in T; out: LDS [UB][ldo] fp32.
// LPR lanes share a weight row (LPR * 16 contiguous bytes per row and load instruction: LPR = 8 fetches whole 128-byte lines),
// NT / LPR rows per pass, a lane walks its share of K in nsteps = K / (LPR * VE) <= U steps (K must be a multiple of LPR * VE: every
// lane of a step is in bounds, so the only conditions are workgroup-uniform -- a load under a per-lane condition makes hipcc branch
// around it and drain the queue).  The kernel is a chain of dependent phases on one workgroup per CU, so what it costs is memory
// round trips: the loads of JU rows (JU * nsteps 16-byte loads per lane) form one set, and the next set is issued before the current
// one is consumed -- two sets, ~200 KB per CU, are in flight for the whole walk.
template <typename T, bool FAST, int UB, int LPR, int U, int JU>
__device__ __forceinline__ void gemv_lds(const T* __restrict__ W, long ldw, int ncols, int K, const T* a, int lda, float* out, int ldo, int tid,
                                         int blk_rows = 1 << 30, long blk_stride = 0) {
    constexpr int VE = DV<T>::VE, STEP = LPR * VE, RG = NT / LPR;
    const int p = tid % LPR, rg = tid / LPR;
    const int nset = (ncols + RG * JU - 1) / (RG * JU);
    const int nsteps = FAST ? U : K / STEP;            // (uniform; FAST: compile-time)
    uint4 wA[JU][U], wB[JU][U];
    auto issue = [&](uint4 (&w)[JU][U], int set) {
#pragma unroll
        for (int q = 0; q < JU; ++q) {
            const int j = min((set * JU + q) * RG + rg, ncols - 1), blk = j / blk_rows;
            const T* wr = W + ((long)blk * blk_stride + (j - blk * blk_rows)) * ldw + p * VE;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (FAST || u < nsteps) w[q][u] = *reinterpret_cast<const uint4*>(wr + u * STEP);
        }
    };
    auto consume = [&](const uint4 (&w)[JU][U], int set) {
        float acc[JU][UB];
#pragma unroll
        for (int q = 0; q < JU; ++q)
#pragma unroll
            for (int r = 0; r < UB; ++r) acc[q][r] = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (FAST || u < nsteps) {
                const int k = u * STEP + p * VE;
                uint4 av[UB];
#pragma unroll
                for (int r = 0; r < UB; ++r) av[r] = *reinterpret_cast<const uint4*>(a + r * lda + k);
#pragma unroll
                for (int q = 0; q < JU; ++q)
#pragma unroll
                    for (int r = 0; r < UB; ++r) acc[q][r] = dotv<T>(w[q][u], av[r], acc[q][r]);
            }
        }
#pragma unroll
        for (int q = 0; q < JU; ++q) {
            const int j = (set * JU + q) * RG + rg;
#pragma unroll
            for (int r = 0; r < UB; ++r) {
                float v = acc[q][r];
#pragma unroll
                for (int o = 1; o < LPR; o <<= 1) v += __shfl_xor(v, o, 64);
                if (j < ncols && p == 0) out[r * ldo + j] = v;
            }
        }
    };
    if (sizeof(T) == 4) {           // fp32 (parity mode): one set at a time
        for (int q = 0; q < nset; ++q) { issue(wA, q); consume(wA, q); }
        return;
    }
    issue(wA, 0);
    for (int q = 0; q < nset; q += 2) {
        if (q + 1 < nset) issue(wB, q + 1);
        consume(wA, q);
        if (q + 2 < nset) issue(wA, q + 2);
        if (q + 1 < nset) consume(wB, q + 1);
    }
}

// The same product on the matrix pipe (bf16, compile-time K): out[r][j] = sum_k a[r][k] * W[row(j)][k].  With a handful of rows the
// VALU version above spends as long on v_dot2c as on the loads; one v_mfma_f32_16x16x32_bf16 takes 16 weight rows x 32 k (A fragment:
// lane l holds W[row l&15][k = (l>>4)*8 .. +8]) against the activations of up to 16 utterances (B fragment from LDS, columns >= UB
// repeat the last row and are never stored), accumulates over K in the accumulator and needs no cross-lane reduction.
// The weights come from a FRAGMENT-PACKED copy (b2s_df_pack, made once per decode job): fragment (row block RB, k step ku) is the 1 KB
// at ((RB * KU + ku) * 64 + lane) * 16 bytes, so a wave's load instruction is one contiguous KB.  Read in place from the [N][K]
// matrix, the 16 lanes of a k group touch 16 different rows -- 64 separate 16-byte requests per instruction, and the texture path
// then delivers 16 B/clk/CU (measured: 442 KB per workgroup in 11-12 us) instead of the 50+ it has for coalesced loads.
// A wave owns row blocks wave, wave + 8, ...; a set is JB row blocks x CU k-steps (K = 32 * CU * CPR; CPR = 2: the two halves of
// a row block's K walk are the two alternating sets), two sets in flight.
// Product row block r is packed row block rb0 + (r / blk_rb) * blk_rbstride + r % blk_rb, k steps [ku0, ku0 + CU * CPR) of its KU.
template <int UB, int CU, int JB, int CPR>
__device__ __forceinline__ void gemv_mfma(const bf16_t* __restrict__ Wp, int KU, int rb0, int ku0, int ncols, const bf16_t* a, int lda, float* out, int ldo,
                                          int tid, int blk_rb, int blk_rbstride) {
    static_assert(CPR == 1 || (CPR == 2 && JB == 1), "a row block's K walk is one set or two");
    constexpr int NW = NT / 64;
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int nrb = (ncols + 15) >> 4;
    const int my_nrb = nrb > wave ? (nrb - wave + NW - 1) / NW : 0;
    const bf16_t* arow = a + min(li, UB - 1) * lda + lg * 8;
    bf16x8_t wA[JB][CU], wB[JB][CU];
    auto issue = [&](bf16x8_t (&w)[JB][CU], int i, int kc) {          // row blocks i*JB .. of this wave, K part kc
#pragma unroll
        for (int q = 0; q < JB; ++q) {
            const int r = wave + min(i * JB + q, my_nrb - 1) * NW, blk = r / blk_rb;
            const long RB = rb0 + (long)blk * blk_rbstride + (r - blk * blk_rb);
            const bf16_t* wr = Wp + ((RB * KU + ku0 + kc * CU) * 64 + lane) * 8;
#pragma unroll
            for (int u = 0; u < CU; ++u) w[q][u] = *reinterpret_cast<const bf16x8_t*>(wr + u * 512);
        }
    };
    f32x4_t acc[JB];
    auto mma = [&](const bf16x8_t (&w)[JB][CU], int kc, bool first) {
#pragma unroll
        for (int q = 0; q < JB; ++q) if (first) acc[q] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            const bf16x8_t bv = *reinterpret_cast<const bf16x8_t*>(arow + (kc * CU + u) * 32);
#pragma unroll
            for (int q = 0; q < JB; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[q][u], bv, acc[q], 0, 0, 0);
        }
    };
    auto store = [&](int i) {                                          // D layout: row (= weight row) (lane>>4)*4 + r, column (= utterance) lane & 15
#pragma unroll
        for (int q = 0; q < JB; ++q) {
            if (i * JB + q >= my_nrb || li >= UB) continue;
            const int j0 = (wave + (i * JB + q) * NW) * 16 + lg * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (j0 + r < ncols) out[li * ldo + j0 + r] = acc[q][r];
        }
    };
    const int nset = (my_nrb + JB - 1) / JB;
    if (nset == 0) return;
    if (CPR == 2) {                 // A = first half of a row block's K, B = second half
        issue(wA, 0, 0);
        for (int i = 0; i < nset; ++i) {
            issue(wB, i, 1);
            mma(wA, 0, true);
            if (i + 1 < nset) issue(wA, i + 1, 0);
            mma(wB, 1, false);
            store(i);
        }
    } else {
        issue(wA, 0, 0);
        for (int i = 0; i < nset; i += 2) {
            if (i + 1 < nset) issue(wB, i + 1, 0);
            mma(wA, 0, true); store(i);
            if (i + 2 < nset) issue(wA, i + 2, 0);
            if (i + 1 < nset) { mma(wB, 0, true); store(i + 1); }
        }
    }
}

// The q / k / v projection of a head (K = D in two halves): the work is cut into (row block, K half) units, unit id = wave + 8 i, so an even
// wave always takes first halves and an odd wave second halves, of row blocks (wave >> 1) + 4 i.  18 row blocks are 36 units = 5 / 5 / 5
// / 5 / 4 / 4 / 4 / 4 per wave (whole row blocks: 3 / 3 / 2 / ... -- the kernel waited for two waves); the two K halves of a row block
// land in out and out + half_stride and the consumer adds them.  NU (units per wave, rounded up) is compile-time and every wave issues
// NU loads sets -- a wave with fewer units re-reads its last one -- so there is no control flow around the loads; NBUF sets in flight.
template <int UB, int CU, int NU, int NBUF>
__device__ __forceinline__ void gemv_mfma_halves(const bf16_t* __restrict__ Wp, int KU, int rb0, int nrb, const bf16_t* a, int lda, float* out, int ldo,
                                                 int half_stride, int tid, int blk_rb, int blk_rbstride) {
    static_assert(NBUF >= 2 && NBUF <= NU + 1, "buffers");
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int kc = wave & 1, r0 = wave >> 1;
    const int my_n = r0 < nrb ? (nrb - r0 + 3) / 4 : 0;                   // units of this wave
    const bf16_t* arow = a + min(li, UB - 1) * lda + kc * CU * 32 + lg * 8;
    bf16x8_t w[NBUF][CU];
    auto issue = [&](bf16x8_t (&wb)[CU], int i) {
        const int r = r0 + 4 * min(i, max(my_n, 1) - 1), blk = min(r, nrb - 1) / blk_rb;
        const long RB = rb0 + (long)blk * blk_rbstride + (min(r, nrb - 1) - blk * blk_rb);
        const bf16_t* wr = Wp + ((RB * KU + kc * CU) * 64 + lane) * 8;
#pragma unroll
        for (int u = 0; u < CU; ++u) wb[u] = *reinterpret_cast<const bf16x8_t*>(wr + u * 512);
    };
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i) issue(w[i], i);
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        if (i + NBUF - 1 < NU) issue(w[(i + NBUF - 1) % NBUF], i + NBUF - 1);        // (compile-time condition)
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < CU; ++u)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i % NBUF][u], *reinterpret_cast<const bf16x8_t*>(arow + u * 32), acc, 0, 0, 0);
        if (i < my_n && li < UB) {                                        // D layout: row (= weight row) (lane>>4)*4 + r, column (= utterance) lane & 15
            float* o = out + kc * half_stride + li * ldo + (r0 + 4 * i) * 16 + lg * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = acc[r];
        }
    }
}

// Dispatch: bf16 at the default widths -> matrix pipe, packed weights; anything else -> the VALU walk on the [N][K] matrix (fp32: one
// set at a time -- it is the parity mode, registers matter more than overlap there).
template <typename T, bool FAST, int UB, int LPR, int U, int JU, int M_CU, int M_JB, int M_CPR>
__device__ __forceinline__ void gemv(const T* __restrict__ W, int ldw, int row0, int col0, int ncols, int K, const T* a, int lda, float* out, int ldo, int tid,
                                     int blk_rows = 1 << 30, int blk_stride = 0) {
    // the product uses rows row0 + row(j), columns [col0, col0 + K) of the [*][ldw] matrix W (bf16 at the default widths: its packed copy)
    if constexpr (FAST && sizeof(T) == 2)
        gemv_mfma<UB, M_CU, M_JB, M_CPR>(reinterpret_cast<const bf16_t*>(W), ldw / 32, row0 / 16, col0 / 32, ncols, reinterpret_cast<const bf16_t*>(a), lda, out, ldo, tid,
                                         blk_rows / 16, blk_stride / 16);
    else gemv_lds<T, FAST, UB, LPR, U, JU>(W + (long)row0 * ldw + col0, ldw, ncols, K, a, lda, out, ldo, tid, blk_rows, blk_stride);
}

// Partial slabs hold the compute dtype (bf16 mode: half the bytes -- every slab row is read by all slices of the next kernel from
// another XCD's L2, i.e. through the fabric, and that traffic was as large as the KV stream).  A 16-byte item is CW<T> columns.
template <typename T> constexpr int CW = 16 / (int)sizeof(T);
template <typename T> __device__ __forceinline__ void slab_add(const uint4& q, float use, float* v);
template <> __device__ __forceinline__ void slab_add<float>(const uint4& q, float use, float* v) {
    v[0] += use * __uint_as_float(q.x); v[1] += use * __uint_as_float(q.y); v[2] += use * __uint_as_float(q.z); v[3] += use * __uint_as_float(q.w);
}
template <> __device__ __forceinline__ void slab_add<bf16_t>(const uint4& q, float use, float* v) {
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] += use * bf2f(w[e] & 0xffff); v[2 * e + 1] += use * bf2f(w[e] >> 16); }
}
template <typename T> __device__ __forceinline__ uint4 slab_pack(const float* v);
template <> __device__ __forceinline__ uint4 slab_pack<float>(const float* v) {
    return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
}
template <> __device__ __forceinline__ uint4 slab_pack<bf16_t>(const float* v) {
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = (uint32_t)f2bf(v[2 * e]) | ((uint32_t)f2bf(v[2 * e + 1]) << 16);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// xs[u][:] = X_in[b_u] + sum_s P_prev[s][b_u]  (fixed order), published to X_out by the slice-0 workgroups; hs[u][:] = T(LayerNorm(xs[u]))
template <typename T, int UB>
__device__ __forceinline__ void load_x_ln(const DfCommon& c, int b0, bool publish, float* xs, T* hs, float* red, int tid) {
    constexpr int W = CW<T>;
    const int D = c.D, DI = D / W;
    const T* P = reinterpret_cast<const T*>(c.P_prev);
    for (int i = tid; i < UB * DI; i += NT) {
        const int u = i / DI, c0 = (i - u * DI) * W, b = min(b0 + u, c.B - 1);
        float v[W];
#pragma unroll
        for (int h = 0; h < W / 4; ++h) {
            const float4 x = *reinterpret_cast<const float4*>(c.X_in + (long)b * D + c0 + 4 * h);
            v[4 * h] = x.x; v[4 * h + 1] = x.y; v[4 * h + 2] = x.z; v[4 * h + 3] = x.w;
        }
        for (int s = 0; s < c.np_prev; ++s)                               // (fixed summation order s = 0, 1, ...)
            slab_add<T>(*reinterpret_cast<const uint4*>(P + ((long)s * c.B + b) * D + c0), 1.f, v);
#pragma unroll
        for (int h = 0; h < W / 4; ++h) {
            const float4 o = make_float4(v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]);
            *reinterpret_cast<float4*>(xs + u * D + c0 + 4 * h) = o;
            if (publish && b0 + u < c.B) *reinterpret_cast<float4*>(c.X_out + (long)b * D + c0 + 4 * h) = o;
        }
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    if (wave < UB) {                           // one wave per row: mean, then variance around it (two passes, as k_ln_fwd)
        float s = 0.f;
        for (int k = lane; k < D; k += 64) s += xs[wave * D + k];
        const float mu = wave_sum(s) / D;
        float q = 0.f;
        for (int k = lane; k < D; k += 64) { const float d = xs[wave * D + k] - mu; q += d * d; }
        const float rs = 1.f / sqrtf(wave_sum(q) / D + c.eps);
        if (lane == 0) { red[2 * wave] = mu; red[2 * wave + 1] = rs; }
    }
    __syncthreads();
    for (int i = tid; i < UB * D; i += NT) {
        const int u = i / D, k = i - u * D;
        TT<T>::st(hs + i, (xs[i] - red[2 * u]) * red[2 * u + 1] * c.ln_g[k] + c.ln_b[k]);
    }
    __syncthreads();
}

// The same in two steps for the default-size kernels: every load of a workgroup's residual rows and partial slabs is issued at once
// ((1 + NPS) 16-byte slab / row loads per item), BEFORE the weight prefetch -- the loads of a wave complete in issue order, and
// the LayerNorm that waits for these rows heads the kernel's dependency chain, the weights are needed later.
template <typename T, int UB, int NPS>
struct XRegs {
    static constexpr int W = CW<T>, NI = UB * FD_D / W, XIT = (NI + NT - 1) / NT;
    float4 x[XIT][W / 4], g[XIT][W / 4], be[XIT][W / 4];
    uint4 p[XIT][NPS > 0 ? NPS : 1];
};
template <typename T, int UB, int NPS>
__device__ __forceinline__ void issue_x(const DfCommon& c, int b0, XRegs<T, UB, NPS>& r, int tid) {
    using XR = XRegs<T, UB, NPS>;
    constexpr int W = XR::W, DI = FD_D / W;
    const T* P = reinterpret_cast<const T*>(c.P_prev);
#pragma unroll
    for (int it = 0; it < XR::XIT; ++it) {
        const int i = min(tid + it * NT, XR::NI - 1);
        const int u = i / DI, c0 = (i - u * DI) * W, b = min(b0 + u, c.B - 1);
#pragma unroll
        for (int h = 0; h < W / 4; ++h) {
            r.x[it][h] = *reinterpret_cast<const float4*>(c.X_in + (long)b * FD_D + c0 + 4 * h);
            r.g[it][h] = *reinterpret_cast<const float4*>(c.ln_g + c0 + 4 * h);      // (LayerNorm scale / shift of the same columns: no round trip of their own later)
            r.be[it][h] = *reinterpret_cast<const float4*>(c.ln_b + c0 + 4 * h);
        }
#pragma unroll
        for (int s = 0; s < NPS; ++s) r.p[it][s] = *reinterpret_cast<const uint4*>(P + ((long)min(s, max(c.np_prev, 1) - 1) * c.B + b) * FD_D + c0);
    }
}
template <typename T, int UB, int NPS>
__device__ __forceinline__ void finish_x_ln(const DfCommon& c, int b0, bool publish, const XRegs<T, UB, NPS>& r, float* xs, T* hs, float* red, int tid) {
    using XR = XRegs<T, UB, NPS>;
    constexpr int W = XR::W, D = FD_D, DI = D / W;
    const float use = c.np_prev > 0 ? 1.f : 0.f;         // (first sublayer of a frame: no previous partial outputs; the slab loads re-read slab 0)
#pragma unroll
    for (int it = 0; it < XR::XIT; ++it) {
        const int i = tid + it * NT;
        if (i >= XR::NI) continue;
        const int u = i / DI, c0 = (i - u * DI) * W, b = b0 + u;
        float v[W];
#pragma unroll
        for (int h = 0; h < W / 4; ++h) { v[4 * h] = r.x[it][h].x; v[4 * h + 1] = r.x[it][h].y; v[4 * h + 2] = r.x[it][h].z; v[4 * h + 3] = r.x[it][h].w; }
#pragma unroll
        for (int s = 0; s < NPS; ++s) slab_add<T>(r.p[it][s], use, v);
#pragma unroll
        for (int h = 0; h < W / 4; ++h) {
            const float4 o = make_float4(v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]);
            *reinterpret_cast<float4*>(xs + u * D + c0 + 4 * h) = o;
            if (publish && b < c.B) *reinterpret_cast<float4*>(c.X_out + (long)b * D + c0 + 4 * h) = o;
        }
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    if (wave < UB) {
        float s = 0.f;
        for (int k = lane; k < D; k += 64) s += xs[wave * D + k];
        const float mu = wave_sum(s) / D;
        float q = 0.f;
        for (int k = lane; k < D; k += 64) { const float d = xs[wave * D + k] - mu; q += d * d; }
        const float rs = 1.f / sqrtf(wave_sum(q) / D + c.eps);
        if (lane == 0) { red[2 * wave] = mu; red[2 * wave + 1] = rs; }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < XR::XIT; ++it) {
        const int i = tid + it * NT;
        if (i >= XR::NI) continue;
        const int u = i / DI, c0 = (i - u * DI) * W;
        const float mu = red[2 * u], rs = red[2 * u + 1];
#pragma unroll
        for (int h = 0; h < W / 4; ++h) {
            const float4 v = *reinterpret_cast<const float4*>(xs + u * D + c0 + 4 * h);
            const float4 g = r.g[it][h], be = r.be[it][h];
            TT<T>::st(hs + u * D + c0 + 4 * h + 0, (v.x - mu) * rs * g.x + be.x);
            TT<T>::st(hs + u * D + c0 + 4 * h + 1, (v.y - mu) * rs * g.y + be.y);
            TT<T>::st(hs + u * D + c0 + 4 * h + 2, (v.z - mu) * rs * g.z + be.z);
            TT<T>::st(hs + u * D + c0 + 4 * h + 3, (v.w - mu) * rs * g.w + be.w);
        }
    }
    __syncthreads();
}

// partial slab: P_out[slice][b_u][:] = T(dropout(of[u][:]))
template <typename T, int UB>
__device__ __forceinline__ void store_partial(const DfCommon& c, int slice, int b0, const float* of, int t, int tid) {
    constexpr int W = CW<T>;
    const DropCfg d = salted(c.drop_res, t);
    const int D = c.D, DI = D / W;
    T* P = reinterpret_cast<T*>(c.P_out);
    for (int i = tid; i < UB * DI; i += NT) {
        const int u = i / DI, k = (i - u * DI) * W, b = b0 + u;
        if (b >= c.B) continue;
        float v[W];
#pragma unroll
        for (int e = 0; e < W; ++e) v[e] = of[u * D + k + e];
        if (d.thresh) {
            const uint32_t idx = (uint32_t)((long)b * D + k);
#pragma unroll
            for (int e = 0; e < W; ++e) v[e] = b2s_keep(d, idx + e) ? v[e] * d.scale : 0.f;
        }
        *reinterpret_cast<uint4*>(P + ((long)slice * c.B + b) * D + k) = slab_pack<T>(v);
    }
}

// Single-query attention of one (utterance, head) by HT threads (one half of the workgroup per utterance; both halves run the same
// phases, the barriers are workgroup-wide): keys [0, n).  sq: the query (fp32 values the compute dtype can hold, LDS); result in ctx
// (LDS, dh values).  ONE pass over the cache: 4 lanes share a key -- each holds the same 16-byte column chunks of the key row (for
// the score) and of the value row (for the weighted sum), 64 keys per pass -- and keep a running maximum / denominator / weighted
// value sum of their key slot (online softmax); the 64 slots are merged once at the end (wave butterfly, then 4 partial rows through
// LDS).  Two barriers instead of the five of the score -> softmax -> value chain (decode.hip: k_dec_attn), K and V rows of the next
// keys in flight while the current ones are consumed.  p keeps the raw scores so that the alignment row can be written afterwards,
// off the path to the output projection.
template <typename T, bool FAST>
__device__ __forceinline__ void attend(const float* sq, const T* Kb, const T* Vb, int ldkv, int n, int dh, float scale, float* p, float* part,
                                       float* red, float* prow, DropCfg dc, uint32_t drop_row, float* ctx, int tid) {
    constexpr int VE = DV<T>::VE;
    constexpr int NCH = FAST ? FD_DH / (4 * VE) : DF_MAX_DH / (4 * VE);        // 16-byte chunks per lane and row
    constexpr int SU = sizeof(T) == 2 ? 2 : 1;                                 // keys per lane and pass set
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int SL = HT / 4, NWU = HT / 64;                                  // key slots per pass, waves per utterance
    const int part4 = tid & 3, kslot = tid >> 2;
    const int nch = FAST ? NCH : dh / (4 * VE);                                // (dh is a multiple of 4 * VE: b2s_df_supported)
    const int nlast = max(n - 1, 0);
    // the query columns of this lane
    float qv[NCH * VE];
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < VE; ++e) qv[i * VE + e] = (FAST || i < nch) ? sq[(i * 4 + part4) * VE + e] : 0.f;
    uint4 kA[SU][NCH], vA[SU][NCH], kB[SU][NCH], vB[SU][NCH];
    auto load = [&](uint4 (&k)[SU][NCH], uint4 (&v)[SU][NCH], int j0) {        // (rows past the end: the last row again, never used)
#pragma unroll
        for (int uu = 0; uu < SU; ++uu) {
            const long ro = (long)min(j0 + uu * SL + kslot, nlast) * ldkv + part4 * VE;
#pragma unroll
            for (int i = 0; i < NCH; ++i)
                if (FAST || i < nch) {
                    k[uu][i] = *reinterpret_cast<const uint4*>(Kb + ro + i * 4 * VE);
                    v[uu][i] = *reinterpret_cast<const uint4*>(Vb + ro + i * 4 * VE);
                }
        }
    };
    float m = -1e30f, l = 0.f, acc[NCH * VE];
#pragma unroll
    for (int i = 0; i < NCH * VE; ++i) acc[i] = 0.f;
    auto consume = [&](const uint4 (&k)[SU][NCH], const uint4 (&v)[SU][NCH], int j0) {
        float sc[SU];
#pragma unroll
        for (int uu = 0; uu < SU; ++uu) {
            const int j = j0 + uu * SL + kslot;
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                if (!(FAST || i < nch)) continue;
                if (sizeof(T) == 2) {
                    const uint32_t w[4] = {k[uu][i].x, k[uu][i].y, k[uu][i].z, k[uu][i].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) d += qv[i * VE + 2 * e] * bf2f(w[e] & 0xffff) + qv[i * VE + 2 * e + 1] * bf2f(w[e] >> 16);
                } else {
                    const float* f = reinterpret_cast<const float*>(&k[uu][i]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) d += qv[i * VE + e] * f[e];
                }
            }
            d += __shfl_xor(d, 1, 64);
            d += __shfl_xor(d, 2, 64);
            sc[uu] = j < n ? d * scale : -1e30f;
            if (j < n && part4 == 0) p[j] = sc[uu];
        }
        float mn = m;
#pragma unroll
        for (int uu = 0; uu < SU; ++uu) mn = fmaxf(mn, sc[uu]);
        const float corr = __expf(m - mn);
        m = mn;
        l *= corr;
#pragma unroll
        for (int i = 0; i < NCH * VE; ++i) acc[i] *= corr;
#pragma unroll
        for (int uu = 0; uu < SU; ++uu) {
            const int j = j0 + uu * SL + kslot;
            const float e0 = j < n ? __expf(sc[uu] - m) : 0.f;
            l += e0;
            float w = e0;
            if (dc.thresh) w = b2s_keep(dc, drop_row + (uint32_t)j) ? e0 * dc.scale : 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                if (!(FAST || i < nch)) continue;
                if (sizeof(T) == 2) {
                    const uint32_t ww[4] = {v[uu][i].x, v[uu][i].y, v[uu][i].z, v[uu][i].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { acc[i * VE + 2 * e] += w * bf2f(ww[e] & 0xffff); acc[i * VE + 2 * e + 1] += w * bf2f(ww[e] >> 16); }
                } else {
                    const float* f = reinterpret_cast<const float*>(&v[uu][i]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i * VE + e] += w * f[e];
                }
            }
        }
    };
    constexpr int PASS = SL * SU;
    load(kA, vA, 0);
    for (int j0 = 0; j0 < n; j0 += 2 * PASS) {           // (keys past n contribute exact zeros: no conditions around loads or math)
        load(kB, vB, j0 + PASS);
        consume(kA, vA, j0);
        load(kA, vA, j0 + 2 * PASS);
        consume(kB, vB, j0 + PASS);
    }
    // merge the key slots: common maximum, then plain sums
    const float wm = wave_max(m);
    if (lane == 0) red[wave] = wm;
    __syncthreads();
    float M = red[0];
#pragma unroll
    for (int w = 1; w < NWU; ++w) M = fmaxf(M, red[w]);
    const float f = __expf(m - M);
    l *= f;
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) l += __shfl_xor(l, o, 64);
#pragma unroll
    for (int i = 0; i < NCH * VE; ++i) {
        float a = acc[i] * f;
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) a += __shfl_xor(a, o, 64);
        acc[i] = a;
    }
    if (lane < 4) {                                      // (lane = part4 here: key slot 0 of the wave holds the wave's sums)
#pragma unroll
        for (int i = 0; i < NCH; ++i)
            if (FAST || i < nch) {
#pragma unroll
                for (int e = 0; e < VE; ++e) part[wave * dh + (i * 4 + part4) * VE + e] = acc[i * VE + e];
            }
        if (lane == 0) red[4 + wave] = l;
    }
    __syncthreads();
    float lsum = red[4];
#pragma unroll
    for (int w = 1; w < NWU; ++w) lsum += red[4 + w];
    const float inv = 1.f / lsum;
    for (int d = tid; d < dh; d += HT) {
        float o = part[d];
#pragma unroll
        for (int w = 1; w < NWU; ++w) o += part[w * dh + d];
        ctx[d] = o * inv;
    }
    if (prow)
        for (int j = tid; j < n; j += HT) prow[j] = __expf(p[j] - M) * inv;
    __syncthreads();
}

inline size_t al16(size_t x) { return (x + 15) & ~(size_t)15; }

// ------------------------------------------------------------------------------------------------ attention sublayer
typedef __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
// L2 warm-up of [base, base + bytes): workgroup `my` of the `n` that run on the same XCD touches its share of the 128-byte lines once
// (4-byte LDS-DMA loads: no destination registers to keep alive; scratch: 2 KB of LDS whose next write comes after a workgroup
// barrier -- the compiler does not order a later ds_write behind the DMA, the barrier's vmcnt(0) does).  The
// weights a frame's kernels use were evicted by the KV streams since the last frame: without this every dependent projection of a
// kernel starts with a cold-miss round trip (~2 us).
__device__ __forceinline__ void l2_warm(const void* base, long bytes, int my, int n, void* scratch, int tid) {
    const int lines = (int)((bytes + 127) >> 7), share = (lines + n - 1) / max(n, 1);
    for (int i = tid; i < share; i += NT) {
        const int L = min(my * share + i, lines - 1);
        __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(base) + (long)L * 128), (lptr_t)(reinterpret_cast<char*>(scratch) + (tid >> 6) * 256), 4, 0, 0);
    }
}

template <typename T, bool SELF, bool FAST>
__global__ __launch_bounds__(NT) void k_df_attn(DfAttn a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int VE = DV<T>::VE, UB = UBA;
    const DfCommon& c = a.c;
    const int tid = threadIdx.x, D = c.D, dh = a.dh, H = a.H;
    const int h = blockIdx.x % H, b0 = (blockIdx.x / H) * UB, t = *c.t;
    // LDS carve-up (must match b2s_df_attn_lds)
    float* xs = reinterpret_cast<float*>(lds);                         // [UB][D]   residual rows; later the output projection
    T* hs = reinterpret_cast<T*>(xs + UB * D);                         // [UB][D]   LayerNorm output
    float* qf = reinterpret_cast<float*>(hs + UB * D);                 // [2][UB][3 dh] q / k / v of this head (bf16 default sizes: the two K halves)
    T* cs = reinterpret_cast<T*>(qf + 2 * 3 * UB * dh);                // [UB][dh]  attention context
    float* red0 = reinterpret_cast<float*>(cs + UB * dh);              // [8]       LayerNorm statistics
    float* scr = red0 + 8;                                             // per utterance: ctx [dh] | p [nmax] | part [4][dh] | red [8]
    const int scr_n = dh + a.nmax + 4 * dh + 8;
    const T* Wq = reinterpret_cast<const T*>(a.Wqkv);
    constexpr int NQ = SELF ? 3 : 1;
    if constexpr (FAST) {                   // default sizes: every row / slab / LayerNorm-parameter load of the workgroup in one go
        constexpr int NPS = SELF ? 32 : 8;
        XRegs<T, UB, NPS> xr;
        issue_x<T, UB, NPS>(c, b0, xr, tid);
        if constexpr (sizeof(T) == 2) {
            // L2 warm-up of this head's weight slice (q / k / v rows + output-projection columns, 590 KB packed): the previous kernels'
            // KV streams have evicted it, and the projection below would otherwise be a chain of cold-miss round trips (each of its
            // load sets waited ~2 us for the fabric).  Every workgroup of the head touches its share of the slice's 128-byte lines
            // once -- 4-byte LDS-DMA loads into a scratch area nobody reads (no destination registers to keep) -- while LayerNorm runs.
            constexpr int KU = FD_D / 32, RBH = FD_DH / 16;
            constexpr int LQ = RBH * KU * 8;                           // lines per q / k / v part: RBH row blocks x KU k steps x 1 KB
            constexpr int LO = (FD_D / 16) * (FD_DH / 32) * 8;         // output projection: FD_D / 16 row blocks x 3 k steps x 1 KB
            constexpr int TOT = NQ * LQ + LO;
            const int g = blockIdx.x / H, ng = max((int)gridDim.x / H, 1), share = (TOT + ng - 1) / ng;
            const char* wq_b = reinterpret_cast<const char*>(Wq);
            const char* wo_b = reinterpret_cast<const char*>(a.Wo);
            for (int i = tid; i < share; i += NT) {
                const int L = min(g * share + i, TOT - 1);
                const int part = L / LQ, off = L - part * LQ, Lo = L - NQ * LQ, rbk = Lo / (LO / (FD_D / 16)), o = Lo - rbk * (LO / (FD_D / 16));
                const char* pq = wq_b + ((long)(part * (FD_D / 16) + h * RBH) * KU) * 1024 + (long)off * 128;
                const char* po = wo_b + ((long)rbk * KU + h * (FD_DH / 32)) * 1024 + (long)o * 128;
                __builtin_amdgcn_global_load_lds((gptr_t)(L < NQ * LQ ? pq : po), (lptr_t)(reinterpret_cast<char*>(scr) + (tid >> 6) * 256), 4, 0, 0);
            }
        }
        finish_x_ln<T, UB, NPS>(c, b0, h == 0, xr, xs, hs, red0, tid);
    } else {
        load_x_ln<T, UB>(c, b0, h == 0, xs, hs, red0, tid);
    }
    // q (and k, v) of this head: rows [h*dh, (h+1)*dh) of each D-row block of the projection weight; the result is [UB][NQ*dh]
    constexpr bool HALVES = FAST && sizeof(T) == 2;                    // the projection arrives as two partial sums (gemv_mfma_halves)
    if constexpr (HALVES)
        gemv_mfma_halves<UB, FD_D / 64, (2 * NQ * FD_DH / 16 + 7) / 8, 3>(reinterpret_cast<const bf16_t*>(Wq), FD_D / 32, h * FD_DH / 16, NQ * FD_DH / 16,
                                                                        reinterpret_cast<const bf16_t*>(hs), FD_D, qf, NQ * FD_DH, 3 * UB * FD_DH, tid,
                                                                        FD_DH / 16, FD_D / 16);
    else
        gemv<T, FAST, UB, 8, KD_STEPS<T, FAST>, 1, FD_D / 64, 1, 2>(Wq, D, h * dh, 0, NQ * dh, D, hs, D, qf, NQ * dh, tid, dh, D);
    __syncthreads();
    T* Kc = reinterpret_cast<T*>(a.Kc);
    T* Vc = reinterpret_cast<T*>(a.Vc);
    for (int i = tid; i < UB * dh; i += NT) {
        const int u = i / dh, d = i - u * dh, b = b0 + u;
        float* row = qf + u * NQ * dh;
        const float* row2 = row + 3 * UB * dh;
        row[d] = rnd<T>(HALVES ? row[d] + row2[d] : row[d]);           // the query as the compute dtype holds it
        if (SELF && b < c.B) {                                          // this frame's key / value row goes to the caches at position t
            const long o = (long)b * a.kv_bstride + (long)h * a.kv_hstride + (long)t * a.ldkv + d;
            TT<T>::st(Kc + o, HALVES ? row[dh + d] + row2[dh + d] : row[dh + d]);
            TT<T>::st(Vc + o, HALVES ? row[2 * dh + d] + row2[2 * dh + d] : row[2 * dh + d]);
        }
    }
    __syncthreads();                                                   // (workgroup scope: the cache rows just written are visible to the loads below)
    DropCfg dc = a.drop_attn;
    dc.key ^= b2s_hash32((uint32_t)t * 2654435761u + 77u);
    {   // one half of the workgroup per utterance
        const int u = tid / HT, ht = tid - u * HT;
        const int b = min(b0 + u, c.B - 1);
        const int n = SELF ? t + 1 : a.klen[b];
        const T* Kb = Kc + (long)b * a.kv_bstride + (long)h * a.kv_hstride;
        const T* Vb = Vc + (long)b * a.kv_bstride + (long)h * a.kv_hstride;
        float* my = scr + (long)u * scr_n;
        float* prow = (a.probs && b0 + u < c.B) ? a.probs + (((long)b * H + h) * a.probs_rows + t) * a.probs_ld : nullptr;
        attend<T, FAST>(qf + u * NQ * dh, Kb, Vb, a.ldkv, n, dh, a.scale, my + dh, my + dh + a.nmax, my + dh + a.nmax + 4 * dh, prow, dc,
                  (uint32_t)((b * H + h) * 4096), my, ht);
        for (int d = ht; d < dh; d += HT) TT<T>::st(cs + u * dh + d, my[d]);
    }
    __syncthreads();
    // this head's share of the output projection: of[u][n] = sum_d ctx[u][d] * Wo[n][h*dh + d]   (K = dh: every row of a pass set in flight)
    gemv<T, FAST, UB, 4, DH_STEPS<T, FAST>, 3, FD_DH / 32, 3, 1>(reinterpret_cast<const T*>(a.Wo), D, 0, h * dh, D, dh, cs, dh, xs, D, tid);
    __syncthreads();
    store_partial<T, UB>(c, h, b0, xs, t, tid);
}

// ------------------------------------------------------------------------------------------------ FFN sublayer
template <typename T, bool FAST>
__global__ __launch_bounds__(NT) void k_df_ffn(DfFfn a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int UB = UBF;
    const DfCommon& c = a.c;
    const int tid = threadIdx.x, D = c.D, FS = a.F / a.ns;
    const int sl = blockIdx.x % a.ns, b0 = (blockIdx.x / a.ns) * UB, t = *c.t;
    float* xs = reinterpret_cast<float*>(lds);                         // [UB][D]
    T* hs = reinterpret_cast<T*>(xs + UB * D);                         // [UB][D]
    float* ff = reinterpret_cast<float*>(hs + UB * D);                 // [UB][FS]
    T* fs = reinterpret_cast<T*>(ff + UB * FS);                        // [UB][FS]
    float* red = reinterpret_cast<float*>(fs + UB * FS);               // [8]
    if constexpr (FAST && sizeof(T) == 2) {
        // Default sizes, bf16: the slice's weights are 96 x 768 (W1) + 768 x 96 (W2) = 42 16-byte loads per lane -- all of them are
        // issued before anything else, so the kernel is ONE memory round trip (weights, residual rows and partial slabs together)
        // instead of a chain of four.  MFMA fragments as in gemv_mfma.
        constexpr int KS1 = FD_D / 32, RB1 = FD_FS / 16, KS2 = FD_FS / 32, NW = NT / 64, RB2W = FD_D / 16 / NW;
        const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
        const bf16_t* W1 = reinterpret_cast<const bf16_t*>(a.W1);
        const bf16_t* W2 = reinterpret_cast<const bf16_t*>(a.W2);
        XRegs<T, UB, 8> xr;
        issue_x<T, UB, 8>(c, b0, xr, tid);
        __builtin_amdgcn_sched_barrier(0);
        bf16x8_t w1[KS1], w2[RB2W][KS2];
        {
            const bf16_t* wr = W1 + (((long)(sl * RB1 + min(wave, RB1 - 1)) * KS1) * 64 + lane) * 8;      // (packed: see gemv_mfma)
#pragma unroll
            for (int u = 0; u < KS1; ++u) w1[u] = *reinterpret_cast<const bf16x8_t*>(wr + u * 512);
        }
        __builtin_amdgcn_sched_barrier(0);
        finish_x_ln<T, UB, 8>(c, b0, sl == 0, xr, xs, hs, red, tid);
#pragma unroll
        for (int q = 0; q < RB2W; ++q) {                 // (streams in under the first projection and the ReLU)
            const bf16_t* w2r = W2 + (((long)(wave + q * NW) * (a.F / 32) + sl * KS2) * 64 + lane) * 8;
#pragma unroll
            for (int u = 0; u < KS2; ++u) w2[q][u] = *reinterpret_cast<const bf16x8_t*>(w2r + u * 512);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (wave < RB1) {
            const bf16_t* arow = reinterpret_cast<const bf16_t*>(hs) + min(li, UB - 1) * D + lg * 8;
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < KS1; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[u], *reinterpret_cast<const bf16x8_t*>(arow + u * 32), acc, 0, 0, 0);
            if (li < UB) {
#pragma unroll
                for (int r = 0; r < 4; ++r) ff[li * FS + wave * 16 + lg * 4 + r] = acc[r];
            }
        }
        __syncthreads();
        const DropCfg dh_ = salted(a.drop_hid, t);
        for (int i = tid; i < UB * FS; i += NT) {
            const int u = i / FS, k = i - u * FS, b = min(b0 + u, c.B - 1);
            float v = fmaxf(ff[i], 0.f);
            if (dh_.thresh) v = b2s_keep(dh_, (uint32_t)((long)b * a.F + sl * FS + k)) ? v * dh_.scale : 0.f;
            TT<T>::st(fs + i, v);
        }
        __syncthreads();
        {
            const bf16_t* arow = reinterpret_cast<const bf16_t*>(fs) + min(li, UB - 1) * FS + lg * 8;
            bf16x8_t bv[KS2];
#pragma unroll
            for (int u = 0; u < KS2; ++u) bv[u] = *reinterpret_cast<const bf16x8_t*>(arow + u * 32);
#pragma unroll
            for (int q = 0; q < RB2W; ++q) {
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < KS2; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[q][u], bv[u], acc, 0, 0, 0);
                if (li < UB) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) xs[li * D + (wave + q * NW) * 16 + lg * 4 + r] = acc[r];
                }
            }
        }
        __syncthreads();
        store_partial<T, UB>(c, sl, b0, xs, t, tid);
    } else {
    load_x_ln<T, UB>(c, b0, sl == 0, xs, hs, red, tid);
    gemv<T, FAST, UB, 8, KD_STEPS<T, FAST>, 1, FD_D / 64, 1, 2>(reinterpret_cast<const T*>(a.W1), D, sl * FS, 0, FS, D, hs, D, ff, FS, tid);
    __syncthreads();
    const DropCfg dh_ = salted(a.drop_hid, t);
    for (int i = tid; i < UB * FS; i += NT) {
        const int u = i / FS, k = i - u * FS, b = min(b0 + u, c.B - 1);
        float v = fmaxf(ff[i], 0.f);
        if (dh_.thresh) v = b2s_keep(dh_, (uint32_t)((long)b * a.F + sl * FS + k)) ? v * dh_.scale : 0.f;
        TT<T>::st(fs + i, v);
    }
    __syncthreads();
    gemv<T, FAST, UB, 8, FS_STEPS<T, FAST>, 3, FD_FS / 32, 3, 1>(reinterpret_cast<const T*>(a.W2), a.F, 0, sl * FS, D, FS, fs, FS, xs, D, tid);
    __syncthreads();
    store_partial<T, UB>(c, sl, b0, xs, t, tid);
    }
}

// ------------------------------------------------------------------------------------------------ prenet + decoder input
// tacotron.py:55-65 at one position + modules.py:113-120: x = (t > 0 and t-1 < len ? prenet(mel[t-1]) : 0) + PE[t] * pe_scale, dropout
template <typename T, bool FAST>
__global__ __launch_bounds__(NT) void k_df_prenet(DfPrenet a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int UB = UBH;
    // default sizes: PN_SL workgroups per utterance pair, each with 1 / PN_SL of dense_final's rows (48 row blocks on 8 waves were six
    // dependent round trips on 32 CUs); dense0 / dense1 are recomputed by each of them (172 KB of weights from L2)
    constexpr int NSL = FAST ? PN_SL : 1;
    const int tid = threadIdx.x, b0 = (blockIdx.x / NSL) * UB, sl = blockIdx.x % NSL, t = *a.t;
    const int NMp = (a.NM + 31) & ~31, HP = a.HP, D = a.D, DS = D / NSL;
    T* tg = reinterpret_cast<T*>(lds);                                 // [UB][NMp]
    T* a1 = tg + UB * NMp;                                             // [UB][HP]
    T* a2 = a1 + UB * HP;                                              // [UB][HP]
    float* o = reinterpret_cast<float*>(a2 + UB * HP);                 // [UB][max(HP, D)]
    const int ldo = max(HP, D);
    if constexpr (FAST && sizeof(T) == 2) {            // (slice = blockIdx % 8 = XCD: the workgroups of a slice share one L2)
        const int my = blockIdx.x / NSL, n = gridDim.x / NSL;
        l2_warm(a.W0, (long)FD_HP * FD_NM * 2, my, n, o, tid);
        l2_warm(a.W1, (long)FD_HP * FD_HP * 2, my, n, o, tid);
        l2_warm(reinterpret_cast<const char*>(a.Wf) + (long)sl * (FD_D / NSL) * FD_HP * 2, (long)(FD_D / NSL) * FD_HP * 2, my, n, o, tid);
    }
    for (int i = tid; i < UB * NMp; i += NT) {
        const int u = i / NMp, k = i - u * NMp, b = min(b0 + u, a.B - 1);
        TT<T>::st(tg + i, (t > 0 && k < a.NM) ? a.mels[((long)b * a.maxT + (t - 1)) * a.NM + k] : 0.f);
    }
    __syncthreads();
    gemv_lds<T, FAST, UB, NM_LPR<T>, NM_STEPS<FAST>, 1>(reinterpret_cast<const T*>(a.W0), a.NM, HP, a.NM, tg, NMp, o, ldo, tid);
    __syncthreads();
    const DropCfg d0 = salted(a.drop0, t), d1 = salted(a.drop1, t);
    for (int i = tid; i < UB * HP; i += NT) {
        const int u = i / HP, k = i - u * HP, b = min(b0 + u, a.B - 1);
        float v = fmaxf(o[u * ldo + k] + a.b0[k], 0.f);
        if (d0.thresh) v = b2s_keep(d0, (uint32_t)(b * HP + k)) ? v * d0.scale : 0.f;
        TT<T>::st(a1 + i, v);
    }
    __syncthreads();
    gemv<T, FAST, UB, 4, HP_STEPS<T, FAST>, 1, FD_HP / 32, 1, 1>(reinterpret_cast<const T*>(a.W1), HP, 0, 0, HP, HP, a1, HP, o, ldo, tid);
    __syncthreads();
    for (int i = tid; i < UB * HP; i += NT) {
        const int u = i / HP, k = i - u * HP, b = min(b0 + u, a.B - 1);
        float v = fmaxf(o[u * ldo + k] + a.b1[k], 0.f);
        if (d1.thresh) v = b2s_keep(d1, (uint32_t)(b * HP + k)) ? v * d1.scale : 0.f;
        TT<T>::st(a2 + i, v);
    }
    __syncthreads();
    gemv<T, FAST, UB, 4, HP_STEPS<T, FAST>, 1, FD_HP / 32, 1, 1>(reinterpret_cast<const T*>(a.Wf), HP, sl * DS, 0, DS, HP, a2, HP, o + sl * DS, ldo, tid);
    __syncthreads();
    DropCfg dx = a.drop_x;
    dx.key ^= b2s_hash32((uint32_t)t + 0x9e3779b9u);
    const float sc = *a.pe_scale;
    for (int i = tid; i < UB * DS; i += NT) {
        const int u = i / DS, k = sl * DS + (i - u * DS), b = b0 + u;
        if (b >= a.B) continue;
        const bool have = t > 0 && (t - 1) < a.lengths[b];
        float v = (have ? o[u * ldo + k] : 0.f) + a.pe[(long)t * D + k] * sc;
        if (dx.thresh) v = b2s_keep(dx, (uint32_t)(b * D + k)) ? v * dx.scale : 0.f;
        a.X[(long)b * D + k] = v;
    }
}

// ------------------------------------------------------------------------------------------------ heads + stop logic
// output LayerNorm, mel_net, stop_net (tacotron.py:112-115 at one position), then synthesize.py:42-45; the last workgroup to
// finish advances the frame counter and publishes {frames, all finished}.
template <typename T, bool FAST>
__global__ __launch_bounds__(NT) void k_df_final(DfFinal a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ int last;
    constexpr int UB = UBH;
    const int tid = threadIdx.x, D = a.D, b0 = blockIdx.x * UB, t = *a.t;
    float* xs = reinterpret_cast<float*>(lds);                         // [UB][D]
    T* hs = reinterpret_cast<T*>(xs + UB * D);                         // [UB][D]
    float* mo = reinterpret_cast<float*>(hs + UB * D);                 // [UB][NM]
    float* red = mo + UB * a.NM;                                       // [8 + UB]
    DfCommon c;
    c.X_in = a.X_in; c.X_out = nullptr; c.P_prev = a.P_prev; c.np_prev = a.np_prev; c.P_out = nullptr; c.B = a.B; c.D = D;
    c.ln_g = a.ln_g; c.ln_b = a.ln_b; c.eps = a.eps; c.t = a.t; c.drop_res = DropCfg{0, 0, 1.f};
    if constexpr (FAST) {
        XRegs<T, UB, 32> xr;
        issue_x<T, UB, 32>(c, b0, xr, tid);
        finish_x_ln<T, UB, 32>(c, b0, false, xr, xs, hs, red, tid);
    } else {
        load_x_ln<T, UB>(c, b0, false, xs, hs, red, tid);
    }
    gemv<T, FAST, UB, 8, KD_STEPS<T, FAST>, 1, FD_D / 64, 1, 2>(reinterpret_cast<const T*>(a.Wmel), D, 0, 0, a.NM, D, hs, D, mo, a.NM, tid);
    const int wave = tid >> 6, lane = tid & 63;
    if (wave < UB) {                           // stop logit: fp32 weights against the compute-dtype activations (ro_rowdot_fwd)
        float s = 0.f;
        for (int k = lane; k < D; k += 64) s += TT<T>::ld(hs + wave * D + k) * a.wstop[k];
        s = wave_sum(s);
        if (lane == 0) red[8 + wave] = s + a.bstop[0];
    }
    __syncthreads();
    for (int i = tid; i < UB * a.NM; i += NT) {
        const int u = i / a.NM, k = i - u * a.NM, b = b0 + u;
        if (b >= a.B) continue;
        const bool active = t < a.lengths[b];
        a.mels[((long)b * a.maxT + t) * a.NM + k] = active ? mo[i] : 0.f;
    }
    __syncthreads();
    if (tid < UB && b0 + tid < a.B) {
        const int b = b0 + tid;
        const bool active = t < a.lengths[b];
        const bool stop = active && red[8 + tid] > 0.f;
        const int fin = a.finished[b] | (stop ? 1 : 0);
        a.finished[b] = fin;
        if (!fin) a.lengths[b] += 1;
    }
    __syncthreads();
    if (tid == 0) {
        __threadfence();                                               // publish finished[] / lengths[] before the ticket
        last = atomicAdd(a.done_cnt, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (last) {                                                        // (L2-served loads: every workgroup's finished[] is visible)
        int mine = 1;
        for (int b = tid; b < a.B; b += NT) mine &= __hip_atomic_load(a.finished + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int all = __syncthreads_and(mine);
        if (tid == 0) {
            *a.done_cnt = 0;
            *a.t = t + 1;
            a.status[0] = t + 1; a.status[1] = all;
        }
    }
}

// W [N][K] (row-major bf16) -> MFMA A-fragment order (gemv_mfma): out[((rb * K/32 + ku) * 64 + l) * 8 + j] = W[rb*16 + (l&15)][ku*32 + (l>>4)*8 + j]
__global__ __launch_bounds__(256) void k_df_pack(const bf16_t* __restrict__ W, int N, int K, bf16_t* __restrict__ out) {
    const long nfrag = (long)(N / 16) * (K / 32) * 64;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nfrag; i += (long)gridDim.x * 256) {
        const int l = (int)(i & 63);
        const long f = i >> 6;
        const int ku = (int)(f % (K / 32));
        const long rb = f / (K / 32);
        *reinterpret_cast<uint4*>(out + i * 8) = *reinterpret_cast<const uint4*>(W + (rb * 16 + (l & 15)) * K + ku * 32 + (l >> 4) * 8);
    }
}

template <typename K, typename A>
int launch(K kern, int grid, size_t lds, const A& a, hipStream_t st) {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, a);
    B2S_LAUNCH_CHECK();
    return 0;
}

}  // namespace

size_t b2s_df_attn_lds(int dtype, int D, int dh, int nmax) {
    const size_t e = dtype ? 2 : 4;
    constexpr size_t UB = UBA;
    return al16((size_t)UB * D * 4 + UB * D * e + 2 * 3 * UB * dh * 4 + UB * dh * e + 8 * 4 + (size_t)UB * (dh + nmax + 4 * dh + 8) * 4 + 64);
}
size_t b2s_df_ffn_lds(int dtype, int D, int F, int ns) {
    const size_t e = dtype ? 2 : 4, FS = F / ns;
    constexpr size_t UB = UBF;
    return al16((size_t)UB * D * 4 + UB * D * e + UB * FS * 4 + UB * FS * e + 2 * UB * 4 + 64);
}
int b2s_df_ffn_slices(int dtype, int D, int H, int F) {
    if (b2s_df_fast_model(D, H, F)) return 32;         // the default sizes: 32 slices of FD_FS hidden units (FAST instantiation only: 96 is
                                                       // not a whole number of the generic walk's 64-element bf16 steps)
    const int step = 8 * (dtype ? 8 : 4);              // generic: the slice is a K walk of 8 lanes per row
    for (int ns = 16; ns >= 1; --ns)
        if (F % ns == 0 && (F / ns) % step == 0 && F / ns <= DF_MAX_FS) return ns;
    return 0;
}
bool b2s_df_supported(int dtype, int D, int H, int F, int NM, int HP, int nmax) {
    const int ve = dtype ? 8 : 4;
    if (H <= 0 || D % H) return false;
    const int dh = D / H;
    if (D > DF_MAX_D || dh > DF_MAX_DH || HP > DF_MAX_HP || NM > DF_MAX_NM || b2s_df_ffn_slices(dtype, D, H, F) == 0) return false;
    // every GEMV walks K in whole steps of (lanes per row) x (16 bytes)
    return D % (8 * ve) == 0 && dh % (4 * ve) == 0 && HP % (4 * ve) == 0 && NM % 16 == 0 &&
           b2s_df_attn_lds(dtype, D, dh, nmax) <= 64 * 1024 && b2s_df_ffn_lds(dtype, D, F, b2s_df_ffn_slices(dtype, D, H, F)) <= 64 * 1024;
}
// the FAST instantiations serve exactly the default sizes (see the top of the file); in bf16 they read fragment-packed weights
bool b2s_df_fast_model(int D, int H, int F) { return D == FD_D && H == FD_D / FD_DH && F == 32 * FD_FS; }
bool b2s_df_attn_packed(int dtype, int D, int H, int F) { return dtype == 1 && b2s_df_fast_model(D, H, F); }
bool b2s_df_ffn_packed(int dtype, int D, int H, int F) { return dtype == 1 && b2s_df_fast_model(D, H, F); }
bool b2s_df_prenet_packed(int dtype, int HP, int NM, int D) { return dtype == 1 && HP == FD_HP && NM == FD_NM && D == FD_D; }
bool b2s_df_final_packed(int dtype, int D, int H, int F) { return dtype == 1 && b2s_df_fast_model(D, H, F); }
int b2s_df_pack(const void* W, int N, int K, void* out, hipStream_t st) {
    B2S_CHECK(W && out && N > 0 && K > 0 && N % 16 == 0 && K % 32 == 0, "pack: %d x %d (needs multiples of 16 x 32)", N, K);
    const long nfrag = (long)(N / 16) * (K / 32) * 64;
    hipLaunchKernelGGL(k_df_pack, dim3((unsigned)std::min<long>((nfrag + 255) / 256, 2048)), dim3(256), 0, st, (const bf16_t*)W, N, K, (bf16_t*)out);
    B2S_LAUNCH_CHECK();
    return 0;
}
int b2s_df_prenet(int dtype, const DfPrenet& a, hipStream_t st) {
    const size_t e = dtype ? 2 : 4;
    constexpr int UB = UBH;
    const size_t lds = al16((size_t)UB * ((a.NM + 31) & ~31) * e + 2 * UB * a.HP * e + (size_t)UB * std::max(a.HP, a.D) * 4 + 64);
    const bool fast = a.HP == FD_HP && a.NM == FD_NM && a.D == FD_D;
    const int grid = (a.B + UB - 1) / UB * (fast ? PN_SL : 1);
    if (dtype) return fast ? launch(k_df_prenet<bf16_t, true>, grid, lds, a, st) : launch(k_df_prenet<bf16_t, false>, grid, lds, a, st);
    return fast ? launch(k_df_prenet<float, true>, grid, lds, a, st) : launch(k_df_prenet<float, false>, grid, lds, a, st);
}
int b2s_df_attn(int dtype, bool self, const DfAttn& a, hipStream_t st) {
    const size_t lds = b2s_df_attn_lds(dtype, a.c.D, a.dh, a.nmax);
    const int grid = a.H * ((a.c.B + UBA - 1) / UBA);
    // (the default-size instantiation sums a compile-time number of partial slabs: every sublayer of the stack must have the default widths)
    const bool fast = a.c.fast && a.c.D == FD_D && a.dh == FD_DH && a.H == FD_D / FD_DH && (a.c.np_prev == 0 || a.c.np_prev == (self ? 32 : 8));
    if (dtype) {
        if (fast) return self ? launch(k_df_attn<bf16_t, true, true>, grid, lds, a, st) : launch(k_df_attn<bf16_t, false, true>, grid, lds, a, st);
        return self ? launch(k_df_attn<bf16_t, true, false>, grid, lds, a, st) : launch(k_df_attn<bf16_t, false, false>, grid, lds, a, st);
    }
    if (fast) return self ? launch(k_df_attn<float, true, true>, grid, lds, a, st) : launch(k_df_attn<float, false, true>, grid, lds, a, st);
    return self ? launch(k_df_attn<float, true, false>, grid, lds, a, st) : launch(k_df_attn<float, false, false>, grid, lds, a, st);
}
int b2s_df_ffn(int dtype, const DfFfn& a, hipStream_t st) {
    const size_t lds = b2s_df_ffn_lds(dtype, a.c.D, a.F, a.ns);
    const int grid = a.ns * ((a.c.B + UBF - 1) / UBF);
    const bool fast = a.c.fast && a.c.D == FD_D && a.ns == 32 && a.F / a.ns == FD_FS && a.c.np_prev == 8;
    if (dtype) return fast ? launch(k_df_ffn<bf16_t, true>, grid, lds, a, st) : launch(k_df_ffn<bf16_t, false>, grid, lds, a, st);
    return fast ? launch(k_df_ffn<float, true>, grid, lds, a, st) : launch(k_df_ffn<float, false>, grid, lds, a, st);
}
int b2s_df_final(int dtype, const DfFinal& a, hipStream_t st) {
    const size_t e = dtype ? 2 : 4;
    constexpr int UB = UBH;
    const size_t lds = al16((size_t)UB * a.D * 4 + UB * a.D * e + (size_t)UB * a.NM * 4 + (8 + UB) * 4 + 64);
    const int grid = (a.B + UB - 1) / UB;
    const bool fast = a.fast && a.D == FD_D && a.np_prev == 32;
    if (dtype) return fast ? launch(k_df_final<bf16_t, true>, grid, lds, a, st) : launch(k_df_final<bf16_t, false>, grid, lds, a, st);
    return fast ? launch(k_df_final<float, true>, grid, lds, a, st) : launch(k_df_final<float, false>, grid, lds, a, st);
}
