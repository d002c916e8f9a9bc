// bf16 MFMA GEMM, 256x128 tiles, for gfx950: the large-M forms of the training step (forward NT, dX NN, dW TN, conv).
//
// Why a second tile shape (measured, tools/gemm_lab.hip + tools/dma_lab.hip): the 128x128 kernel (gemm_glds.hip) is bound
// by the L2 -> LDS operand stream, not by MFMA, LDS or latency (no-DMA variant: 1.75 PFLOP/s at 8192^3 against 0.87 with
// it).  Two things make that stream cheaper here:
//   * 256x128 tiles move 25 % fewer operand bytes per FLOP, and
//   * K-contiguous operands are fetched in whole 128-byte lines: BK = 64, one LDS-DMA instruction = 8 rows x 128 B
//     (16 rows x 64 B half lines stream at ~28 B/clk/CU, whole lines at ~47).
// One workgroup = 8 MFMA waves (4 along M x 2 along N, 64x64 accumulators each, the same 4x4 MFMA block as the 128x128
// kernel) + 4 producer waves (one per SIMD) that do nothing but issue the LDS-DMA loads; one workgroup per CU, 3-stage LDS
// ring of 48 KB stages (two stages in flight), one s_barrier per 64-deep K step shared by both roles.
// Why producer waves (s_memtime attribution, profiles/README.md): a wave that issues an LDS-DMA instruction blocks until
// the texture path accepts it, ~400 clocks per K step for 6 instructions, and all 8 waves hit that phase together right
// after the barrier, so the matrix pipe idled.  With the issue moved to waves that have nothing else to do the MFMA waves
// never touch the vector-memory queue: 592 -> 660 TFLOP/s on the step's shape mix, 768 -> 965 on 8148x768x3072.
// (12 waves = 3 per SIMD cap the kernel at 168 VGPRs: 148-166 used.)
//
// LDS images (the LDS-DMA writes lane-linearly, so every swizzle is applied on the SOURCE address):
//   K-contiguous operand ("N layout"):  [rows][64 k], 128 B rows, physical 16-B chunk = chunk ^ ((row >> 1) & 7)
//        -> ds_read_b128 of 16 consecutive rows x one chunk touches 16 distinct 16-B slots of the 256-B bank row.
//   reduction-major operand ("T layout"): units of [32 k][128 cols] exactly as in gemm_glds.hip (ds_read_b64_tr_b16),
//        A: 4 units (k half, column half), B: 2 units (k half).
#include <algorithm>
#include <vector>
#include <cstdlib>
#include <mutex>
#include <atomic>
#include "gemm.h"
#include "gemm_epi.h"

#ifndef B2S_DMA_AUX
#define B2S_DMA_AUX 0       // cache-policy bits of the LDS-DMA loads (sc0 = 1, nt = 2, sc1 = 16); measured: no policy beats the default
#endif

#ifndef B2S_GROUP_M
#define B2S_GROUP_M 4
#endif
#ifndef B2S_NPROD
#define B2S_NPROD 4         // >0: that many extra waves per workgroup do nothing but issue the LDS-DMA loads (producer / consumer split)
#endif

namespace t256 {

constexpr int NPROD = B2S_NPROD, GROUP_M = B2S_GROUP_M;
// GATHER: 0 = plain operands, 1 = conv gather with general addressing (issued by the MFMA waves), 2 = conv gather on a
// K-contiguous A whose channel count is a multiple of BK (every K step lies inside one tap: wave-uniform tap / channel base,
// a compare + select per lane; issued by the producer waves)
// MW: rows of MFMA waves.  4 -> 256-row tiles (8 MFMA + 4 producer waves), 2 -> 128-row tiles (4 + 2 waves, 96 KB ring) for
// problems whose 256-row tiling leaves most CUs without a workgroup (the 1596-row encoder GEMMs: 42 tiles on 256 CUs).
constexpr int nprod_of(int gather, int mw = 4) { return gather == 1 ? 0 : NPROD * mw / 4; }
constexpr int nthreads_of(int gather, int mw = 4) { return 64 * (2 * mw + nprod_of(gather, mw)); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int BM = 256, BK = 64, NSTAGE = 3;            // BN = 128 or 96 (template parameter NB = BN / 32)
constexpr int A_BYTES = BM * BK * 2, B_BYTES = 128 * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;     // 32 KB + 16 KB (12 KB used at BN = 96)
constexpr int UNIT = 8192;                                                                        // [32 k][128 cols] bf16

typedef __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ inline int swz_t(int k) { return ((k & 3) | ((k >> 1) & 4)) << 1; }
__device__ inline int swz_n(int r) { return (r >> 1) & 7; }

// floor(x / d) for 0 <= x < 2^24, 1 <= d: one reciprocal multiply + a correction step (the gather indices below would
// otherwise cost two ~30-instruction integer divisions per DMA instruction and K step)
__device__ inline int fast_div(int x, int d, float inv_d) {
    int q = (int)(((float)x + 0.5f) * inv_d);
    const int r = x - q * d;
    q += (r >= d) - (r < 0);
    return q;
}
__device__ inline const bf16_t* chunk_src(const GemmOperand& o, const bf16_t* base, int r, int c, const bf16_t* zero) {
    if (r >= o.R || c >= o.C) return zero;
    if (o.g_cin > 0) {
        const int j = fast_div(c, o.g_cin, __builtin_amdgcn_rcpf((float)o.g_cin)), ci = c - j * o.g_cin;
        const int b = fast_div(r, o.g_T, __builtin_amdgcn_rcpf((float)o.g_T)), t = r - b * o.g_T;
        const int ts = t + j - 2;
        const int lim = o.g_len ? min(o.g_len[b], o.g_T) : o.g_T;
        if (ts < 0 || ts >= lim) return zero;
        return base + (long)(b * o.g_T + ts) * o.ld + ci;
    }
    return base + (long)r * o.ld + c;
}

// one LDS-DMA instruction of an operand tile: which stored (row offset, col offset) does this lane fetch?
//   N layout, instruction q: rows q*8 .. q*8+7, whole 128-B lines.     returns (row in tile, k element offset)
//   T layout, instruction idx = unit*8 + qi: unit = k half [* 2 + column half]; k row = khalf*32 + qi*4 + lane/16
struct LaneSrc { int r, c; };     // r: offset along the operand's stored rows, c: offset along its stored columns (elements)
template <bool T, bool IS_A>
__device__ inline LaneSrc lane_src(int idx, int lane) {
    LaneSrc s;
    if (!T) {
        const int row = idx * 8 + (lane >> 3), pc = lane & 7;
        s.r = row; s.c = (pc ^ swz_n(row)) << 3;
    } else {
        const int unit = idx >> 3, qi = idx & 7;
        const int khalf = IS_A ? (unit >> 1) : unit, chalf = IS_A ? (unit & 1) : 0;
        const int krl = qi * 4 + (lane >> 4), pc = lane & 15;
        s.r = khalf * 32 + krl; s.c = chalf * 128 + ((pc ^ swz_t(krl)) << 3);
    }
    return s;
}

// linear workgroup id -> XCD-aware tile id.  Workgroups are dealt to the 8 XCDs round-robin (linear id % 8) and every XCD
// has its own L2: remap so that XCD x walks one contiguous range of the row-major tile list, i.e. the 32 workgroups
// resident on an XCD share a handful of A row panels and all of its B column panels instead of ~32 different ones.
__device__ inline int xcd_tile_id(int orig, int nwg) {
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

template <bool TA, bool TB, int GATHER, int NB, int MW = 4>
__device__ __forceinline__ void gemm256_body(GemmArgs g, const bf16_t* zero, float* splitk_ws, int bx, int by, int bz) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // conv-gather operands keep the DMA issue on the MFMA waves: their per-chunk address arithmetic (two divisions, a length
    // lookup) serialised on 4 producer waves costs more than the blocking issue does (60 -> 84 us on the postnet GEMMs)
    static_assert(MW == 4 || (MW == 2 && !TA && GATHER == 0), "128-row tiles: plain K-contiguous A only");
    constexpr int NP = nprod_of(GATHER, MW), NC = 2 * MW;                       // producer / MFMA waves
    constexpr int BMt = MW * 64, A_B = BMt * BK * 2, STG = A_B + B_BYTES;       // tile rows, A image and stage bytes of this variant

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform -> SGPR: DMA destinations / branches on it are scalar
    const int li = lane & 15, lg = lane >> 4;
    constexpr int BN = NB * 32;                      // two waves along N, NB 16-column MFMA blocks each
    const int wrow = (wave >> 1) * 64, wcol = (wave & 1) * (NB * 16);
    const int m0 = by * BMt, n0 = bx * BN;
    const int z = bz / g.splitk, ksplit = bz - z * g.splitk;
    const int zo = z / g.batch_inner, zi = z - zo * g.batch_inner;
    const bf16_t* Ab = reinterpret_cast<const bf16_t*>(g.A.p) + zo * g.A.bs_o + zi * g.A.bs_i;
    const bf16_t* Bb = reinterpret_cast<const bf16_t*>(g.B.p) + zo * g.B.bs_o + zi * g.B.bs_i;

    // DMA issue: wave w owns A instructions w*4 .. w*4+3 and B instructions w*2, w*2+1 of every stage (6 per wave).
    // Plain operands: the source address of a lane is affine in the stage index -> pointer at stage 0 + per-stage step.
    // With producer waves (NP > 0) the 32 A + 16 (12 at BN = 96, K-contiguous B) instructions are dealt to waves 8.. instead.
    constexpr int NBI = (NB == 4 || TB) ? 16 : NB * 4;      // (8 rows of the B image per instruction: 64 / 96 / 128 columns)
    static_assert(NB >= 3 || (GATHER == 2 && !TB), "64-column tiles: the aligned conv gather only");
    constexpr int NPD = NP ? NP : 1, NIA = NP ? (BMt / 8) / NPD : (BMt / 8) / NC, NIB = NP ? NBI / NPD : 2, IPW = NIA + NIB;
    const int iw = NP ? max(wave - NC, 0) : wave;
    const int ia0 = iw * NIA, ib0 = iw * NIB;
    const bf16_t* pa[NIA]; const bf16_t* pb[NIB];
    int ka[NIA], kb_[NIB];        // reduction offset this lane's chunk covers inside a stage; -1 = never valid
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
        const LaneSrc s = lane_src<TA, true>(ia0 + i, lane);
        if (TA) { ka[i] = (m0 + s.c) < g.A.C ? s.r : -1; pa[i] = Ab + (long)s.r * g.A.ld + m0 + s.c; }
        else    { ka[i] = (m0 + s.r) < g.A.R ? s.c : -1; pa[i] = Ab + (long)(m0 + s.r) * g.A.ld + s.c; }
    }
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
        const LaneSrc s = lane_src<TB, false>(ib0 + i, lane);
        // (at BN = 96 the last quarter of the 128-wide B image belongs to no wave: it is filled from the zero page)
        if (TB) { kb_[i] = ((n0 + s.c) < g.B.C && s.c < BN) ? s.r : -1; pb[i] = Bb + (long)s.r * g.B.ld + n0 + s.c; }
        else    { kb_[i] = ((n0 + s.r) < g.B.R && s.r < BN) ? s.c : -1; pb[i] = Bb + (long)(n0 + s.r) * g.B.ld + s.c; }
    }
    const long stepA = TA ? (long)BK * g.A.ld : BK, stepB = TB ? (long)BK * g.B.ld : BK;
    const int limA = TA ? g.A.R : g.A.C, limB = TB ? g.B.R : g.B.C;      // bound of the reduction index
    // Fast DMA issue for K steps that lie completely inside the matrices (measured with s_memtime: the address arithmetic
    // of the generic issue below -- 64-bit pointer math, bound checks, zero-page selects, ~10 VALU per instruction -- cost
    // as much as a whole 32-deep half step of MFMAs on every wave).  A lane's source address is
    //     (operand base + K step * stride)  [wave-uniform, SGPRs]  +  a 32-bit per-lane byte offset fixed for the launch,
    // so a step costs no VALU work at all.  Rows / columns past the M / N edge are clamped to the last valid one: they only
    // feed output rows / columns that are never stored.  Only a partial last K step needs zero fill -> generic path.
    unsigned goffA[NIA], goffB[NIB];
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
        const LaneSrc s = lane_src<TA, true>(ia0 + i, lane);
        if (TA) goffA[i] = 2u * (unsigned)((long)s.r * g.A.ld + min(m0 + s.c, max(g.A.C - 8, 0)));
        else    goffA[i] = 2u * (unsigned)((long)min(m0 + s.r, g.A.R - 1) * g.A.ld + s.c);
    }
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
        const LaneSrc s = lane_src<TB, false>(ib0 + i, lane);
        if (TB) goffB[i] = 2u * (unsigned)((long)s.r * g.B.ld + min(n0 + s.c, max(g.B.C - 8, 0)));
        else    goffB[i] = 2u * (unsigned)((long)min(n0 + s.r, g.B.R - 1) * g.B.ld + s.c);
    }
    const int nfull = g.K / BK;                                          // K steps completely in bounds
    // aligned conv gather (GATHER == 2, conditions checked by the launcher): token position / valid length of every row
    // this lane fetches; a K step then needs one unsigned compare per DMA instruction
    constexpr bool GF = GATHER == 2;
    int gt[GF ? NIA : 1], glim[GF ? NIA : 1];
    if (GF) {
        const float inv_T = __builtin_amdgcn_rcpf((float)g.A.g_T);
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
            const LaneSrc s = lane_src<false, true>(ia0 + i, lane);
            const int r = min(m0 + s.r, g.A.R - 1), b = fast_div(r, g.A.g_T, inv_T);
            gt[i] = r - b * g.A.g_T;
            glim[i] = g.A.g_len ? min(g.A.g_len[b], g.A.g_T) : g.A.g_T;
        }
    }
    const int g_spt = GF ? g.A.g_cin / BK : 1;                           // K steps per filter tap
    const bool fast_ok = GATHER != 1 && nfull > 0 && (TA ? (g.A.C >= 8 && (g.A.C & 7) == 0) : true) && (TB ? (g.B.C >= 8 && (g.B.C & 7) == 0) : true) &&
                         (TA ? (long)g.A.R * g.A.ld : (long)g.A.R * g.A.ld) < (1L << 30) && (long)g.B.R * g.B.ld < (1L << 30);
    // B instructions 12..15 of a 96-column tile would fill image rows no wave reads: waves 6, 7 skip them (4 DMAs per step)
    // (K-contiguous B only: a reduction-major B instruction covers 4 k rows x all 128 columns, every one is needed)
    const bool b_active0 = NP || NB == 4 || TB || wave * 2 + 0 < 12, b_active1 = NP || NB == 4 || TB || wave * 2 + 1 < 12;
    const bool six = b_active0 && b_active1;                             // this wave issues 6 (else 4) DMA instructions per step

    const int nk_all = (g.K + BK - 1) / BK;
    const int per = (nk_all + g.splitk - 1) / g.splitk;
    const int kt0 = ksplit * per;
    const int kt_end = min(nk_all, kt0 + per);
    const int nk = kt_end - kt0;
    if (nk <= 0) return;

    // Always exactly 6 DMA instructions per wave and stage (stages past the end fetch the zero page into a slot nobody
    // reads again): the in-flight count is a compile-time constant and the counted waits below never drain the queue.
    auto issue = [&](int kt, int slot) {
        unsigned char* sbase = smem_raw + slot * STG;
        if (fast_ok && (kt < nfull || kt >= kt_end)) {                   // (steps past the end re-fetch the last full one: never consumed)
            const long ks = min(kt, nfull - 1);
            const char* sa = reinterpret_cast<const char*>(Ab) + ks * stepA * 2;
            const char* sb = reinterpret_cast<const char*>(Bb) + ks * stepB * 2;
            if (GF) {
                // virtual column ks*64 + c = tap j, channel ci0 + c  ->  x[row + j - 2][ci0 + c], zero outside the utterance
                const int j = fast_div((int)ks, g_spt, __builtin_amdgcn_rcpf((float)g_spt)), ci0 = ((int)ks - j * g_spt) * BK;
                sa = reinterpret_cast<const char*>(Ab) + ((long)(j - 2) * g.A.ld + ci0) * 2;
#pragma unroll
                for (int i = 0; i < NIA; ++i) {
                    const bool ok = (unsigned)(gt[i] + j - 2) < (unsigned)glim[i];
                    const char* src = ok ? sa + goffA[i] : reinterpret_cast<const char*>(zero);
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sbase + (ia0 + i) * 1024), 16, 0, B2S_DMA_AUX);
                }
            } else {
#pragma unroll
            for (int i = 0; i < NIA; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(sa + goffA[i]), (lptr_t)(sbase + (ia0 + i) * 1024), 16, 0, B2S_DMA_AUX);
            }
#pragma unroll
            for (int i = 0; i < NIB; ++i)
                if (NP || (i == 0 ? b_active0 : b_active1))
                    __builtin_amdgcn_global_load_lds((gptr_t)(sb + goffB[i]), (lptr_t)(sbase + A_B + (ib0 + i) * 1024), 16, 0, B2S_DMA_AUX);
            return;
        }
        if (GF) return;                      // (the aligned gather has no partial K steps: nothing below is reachable, keep it out of the register budget)
        const int kb = kt < kt_end ? kt * BK : (1 << 28);
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
            const int idx = ia0 + i;
            const bf16_t* sa;
            if (GATHER == 1) {
                const LaneSrc s = lane_src<TA, true>(idx, lane);
                sa = TA ? chunk_src(g.A, Ab, kb + s.r, m0 + s.c, zero) : chunk_src(g.A, Ab, m0 + s.r, kb + s.c, zero);
            } else {
                sa = (ka[i] >= 0 && kb + ka[i] < limA) ? pa[i] + kt * stepA : zero;
            }
            __builtin_amdgcn_global_load_lds((gptr_t)sa, (lptr_t)(sbase + idx * 1024), 16, 0, B2S_DMA_AUX);
        }
#pragma unroll
        for (int i = 0; i < NIB; ++i) {
            const int idx = ib0 + i;
            if (!NP && !(i == 0 ? b_active0 : b_active1)) continue;
            const bf16_t* sb;
            if (GATHER == 1) {
                const LaneSrc s = lane_src<TB, false>(idx, lane);
                sb = TB ? (s.c < BN ? chunk_src(g.B, Bb, kb + s.r, n0 + s.c, zero) : zero)
                        : (s.r < BN ? chunk_src(g.B, Bb, n0 + s.r, kb + s.c, zero) : zero);
            } else {
                sb = (kb_[i] >= 0 && kb + kb_[i] < limB) ? pb[i] + kt * stepB : zero;
            }
            __builtin_amdgcn_global_load_lds((gptr_t)sb, (lptr_t)(sbase + A_B + idx * 1024), 16, 0, B2S_DMA_AUX);
        }
    };

    // per-lane byte offsets of the fragment reads inside a stage, for the two 32-deep halves of the K step
    unsigned offA[2][4], offB[2][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {      // (offB[.][t] for t >= NB is never used)
        if (TA) {
            const int k = lg * 8 + (li >> 2), col = (wrow & 127) + t * 16 + (li & 3) * 4, sw = ((li >> 2) | ((lg & 1) << 2)) << 1;
            const unsigned o = 2u * (k * 128 + (((col >> 3) ^ sw) << 3) + (col & 7)) + (unsigned)((wrow >> 7) * UNIT);
            offA[0][t] = o; offA[1][t] = o + 2 * UNIT;
        } else {
            const int r = wrow + t * 16 + li;
            offA[0][t] = (unsigned)(r * 128 + ((lg ^ swz_n(r)) << 4)); offA[1][t] = offA[0][t] ^ 64u;
        }
        if (TB) {
            const int k = lg * 8 + (li >> 2), col = wcol + t * 16 + (li & 3) * 4, sw = ((li >> 2) | ((lg & 1) << 2)) << 1;
            const unsigned o = (unsigned)A_B + 2u * (k * 128 + (((col >> 3) ^ sw) << 3) + (col & 7));
            offB[0][t] = o; offB[1][t] = o + UNIT;
        } else {
            const int r = wcol + t * 16 + li;
            offB[0][t] = (unsigned)(A_B + r * 128 + ((lg ^ swz_n(r)) << 4)); offB[1][t] = offB[0][t] ^ 64u;
        }
    }
    const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)smem_raw;
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

    f32x4_t acc[4][NB];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

#define B2S_MMA16(CA, CB)                                                                                         \
    _Pragma("unroll") for (int a = 0; a < 4; ++a) _Pragma("unroll") for (int b = 0; b < NB; ++b)                  \
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(CA[a], CB[b], acc[a][b], 0, 0, 0);
#define B2S_READ8(FA, FB, SLOT, H)                                                                                \
    {                                                                                                              \
        const unsigned sb_ = lds_base + (unsigned)((SLOT) * STG);                                          \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                            \
            frag_issue(FA[t], TA, sb_ + offA[H][t]);                                                               \
            if (t < NB) frag_issue(FB[t], TB, sb_ + offB[H][t]);                                                   \
        }                                                                                                          \
    }

    if (NP && wave >= NC) {
        // Producer wave: keeps two stages in flight and never touches the matrix pipe.  Barrier k of the workgroup (k = 0 in
        // the prologue) says "stage k has landed and every consumer is done with stage k - 1"; the freed slot is refilled
        // right behind it.  The consumers run the same barrier sequence and never wait on a DMA issue.
#pragma unroll
        for (int p = 0; p < NSTAGE; ++p)
            if (p < nk) issue(kt0 + p, p);
        if (nk >= 3) wait_vm<2 * IPW>(); else if (nk == 2) wait_vm<IPW>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        int slot = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 2 < nk) wait_vm<IPW>(); else wait_vm<0>();          // stage kt + 1 landed (stage kt + 2 may still fly)
            __builtin_amdgcn_s_barrier();
            if (kt + NSTAGE < nk) issue(kt0 + kt + NSTAGE, slot);
            slot = slot + 1 == NSTAGE ? 0 : slot + 1;
        }
        return;
    }

    // prologue: three stages in flight, stage 0 landed and published, its first-half fragments in registers
    if (!NP) {
#pragma unroll
        for (int p = 0; p < NSTAGE; ++p) issue(kt0 + p, p);
    }
    bf16x8_t fa0[4], fb0[4], fa1[4], fb1[4];
    if (!NP) { if (six) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    __builtin_amdgcn_s_barrier();
    B2S_READ8(fa0, fb0, 0, 0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int nslot = slot + 1 == NSTAGE ? 0 : slot + 1;
        // first half: second-half fragments of this stage fly under the first-half MFMAs
        B2S_READ8(fa1, fb1, slot, 1)
        __builtin_amdgcn_sched_barrier(0);
        B2S_MMA16(fa0, fb0)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave no longer reads stage kt
        // stage kt+1 landed (own DMAs; the 6 of stage kt+2 stay in flight) ... for every wave, and slot `slot` is free
        if (!NP) { if (six) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
        // second half: first-half fragments of the next stage fly under the second-half MFMAs; the DMA issue of stage kt+3
        // (6 instructions + their address arithmetic) comes AFTER the MFMAs have been issued, so its VALU work runs while
        // the matrix pipe drains instead of delaying this wave's MFMAs right after the barrier
        B2S_READ8(fa0, fb0, nslot, 0)
        __builtin_amdgcn_sched_barrier(0);
        B2S_MMA16(fa1, fb1)
        if (!NP) issue(kt0 + kt + NSTAGE, slot);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        slot = nslot;
    }
    if (!NP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the trailing zero-page DMAs before the LDS is reused
#undef B2S_READ8
#undef B2S_MMA16

    // ---------------- epilogue (gemm_epi.h): accumulators -> per-wave LDS staging -> vectorised, fused stores
    gemm_wave_epilogue<NB>(g, acc, reinterpret_cast<float*>(smem_raw) + wave * (64 * 64), m0 + wrow, n0 + wcol, lane, z, zo, zi, ksplit, splitk_ws);
}

template <bool TA, bool TB, int GATHER, int NB, int MW = 4>
__global__ __launch_bounds__(nthreads_of(GATHER, MW), 1) void gemm_glds256_kernel(GemmArgs g, const bf16_t* zero, float* splitk_ws, int tiles_m, int tiles_n) {
    const int wg = xcd_tile_id(blockIdx.x, gridDim.x);
    const int per_z = tiles_n * tiles_m;
    const int bz = wg / per_z, rem = wg - bz * per_z;
    // Tile order inside the (XCD-contiguous) id range.  Wide outputs (>= 16 column panels): groups of GROUP_M row panels
    // walked column by column, so the 32 workgroups an XCD runs at a time cover 4 row panels x 8 column panels -- equal A and
    // B bytes per round, and the working set (3 MB at K = 768) fits the XCD's 4 MB L2.  Row-major order made every XCD
    // re-fetch the whole weight matrix once per round (PMC: 126 MB fetched per forward ffn-in launch for 50 MB of
    // per-XCD operands); measured 53.8 -> 50.2 us (N = 2304), 65.5 -> 59.3 (dX, N = 3072), 977 -> 927 (8192^3).  Narrow
    // outputs keep the row-major order: an XCD's range is the same set of tiles either way, and the column-major walk
    // measured 8-17 % slower on the K >= 2304 shapes.
    int by, bx;
    if (GROUP_M > 0 && tiles_n >= 16) {
        const int gsz = GROUP_M * tiles_n, grp = rem / gsz, in = rem - grp * gsz;
        const int rows = min(GROUP_M, tiles_m - grp * GROUP_M);
        bx = in / rows; by = grp * GROUP_M + (in - bx * rows);
    } else { by = rem / tiles_n; bx = rem - by * tiles_n; }
    gemm256_body<TA, TB, GATHER, NB, MW>(g, zero, splitk_ws, bx, by, bz);
}

// Grouped weight-gradient launch: up to B2S_MAX_GROUP independent dW = dY^T X problems (TN form, fp32 accumulate) in one
// grid.  A training layer's weight gradients are 18 .. 72 tiles each -- far too few to fill 256 CUs one at a time, which
// is what split-K + a slab-reduce kernel used to paper over; together they are ~290 tiles with the full 8148-deep K.
template <int NB>
__global__ __launch_bounds__(nthreads_of(false), 1) void gemm_glds256_grouped_kernel(b2s_gemm_group grp, const bf16_t* zero) {
    // Workgroup i runs on XCD i % 8.  The list is walked in launch order (problem ranges tile0[]), so every XCD gets an equal
    // share of every problem -- a globally XCD-contiguous order handed all 36 short tiles of a decoder layer to one XCD and
    // 36 long tiles to each of the other seven (32 CUs each): a second round of 8148-deep tiles, 213 us instead of ~120.
    // Inside a problem the ids that land on one XCD are mapped to a contiguous piece of its row-major tile list (shared
    // A / B panels stay in that XCD's L2).
    const int i = blockIdx.x, xcd = i & 7;
    int p = 0, local;
    if (grp.order == 1) {
        // One launch holds problems of ONE depth class (engine: flush_dw), so a globally XCD-contiguous order is balanced: XCD x takes
        // the x-th eighth of the whole tile list -- ~32 neighbouring tiles, mostly of one problem, i.e. ~5 A panels + 6 B panels per XCD
        // instead of an eighth of EVERY problem's tiles (1-2 A panels + up to 10 B panels of each of the 6-7 problems)
        const int T = grp.tile0[grp.n];
        int start = 0;
        for (int x = 0; x < xcd; ++x) start += (T - x + 7) >> 3;        // ids x, x + 8, ... < T
        const int g = start + (i >> 3);
        while (p + 1 < grp.n && g >= grp.tile0[p + 1]) ++p;
        local = g - grp.tile0[p];
    } else {
        while (p + 1 < grp.n && i >= grp.tile0[p + 1]) ++p;
        const int s0 = grp.tile0[p], cnt = grp.tile0[p + 1] - s0;
        int before = 0;                                     // tiles of this problem on lower-numbered XCDs
        for (int x = 0; x < xcd; ++x) {
            const int first = s0 + ((x - s0) & 7);          // smallest id >= s0 on XCD x
            before += first < s0 + cnt ? (s0 + cnt - 1 - first) / 8 + 1 : 0;
        }
        local = before + (i - (s0 + ((xcd - s0) & 7))) / 8;
    }
    const int tiles_n = (grp.p[p].N + NB * 32 - 1) / (NB * 32), tiles_m = (grp.p[p].M + BM - 1) / BM;
    // K split over `splitk` workgroups per output tile (fp32 atomic accumulate; with two halves added to a zeroed gradient the result does
    // not depend on their order): the K parts of a tile are adjacent in the XCD's list, so they share its A / B panels' neighbours in L2
    const int sk = grp.p[p].splitk, t = local / sk, bz = local - t * sk;
    int by, bx;
    // wide outputs (tiles_n > 8: W2's 3 x 24 tiles) are walked column by column: a run of 32 tiles is then 3 A panels x 11 B panels
    // instead of 2 x 24
    if (grp.order == 1 && tiles_n > 8) { bx = t / tiles_m; by = t - bx * tiles_m; }
    else { by = t / tiles_n; bx = t - by * tiles_n; }
    gemm256_body<true, true, 0, NB>(grp.p[p], zero, nullptr, bx, by, bz);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Persistent variant for launches of MORE than one round of tiles (the step's N >= 2304 forms at M = 8148: 768 tiles of 256 x 96 / 128):
// 256 workgroups walk the (XCD-aware) tile list.  The operand ring keeps rolling across the tile boundary -- the producer waves number
// the K steps of all of the workgroup's tiles consecutively, so the next tile's first three stages are in flight while the MFMA waves
// store the finished one -- and the epilogue is OFF the ring: a wave stages 8 rows x (NB * 16) columns of compute-dtype results at a
// time through a private 1 KB slice behind the ring (the plain kernel's epilogue takes 128 of the ring's 144 KB, which is why it cannot
// overlap anything).  What a tile boundary costs is then the epilogue itself, not workgroup launch + address set-up + a three-stage
// prologue (8148 x 2304 x 768: 43.6 -> 32.6 us).
// Which tiles a workgroup walks: its first one is static (workgroup p -> list position p, as in the plain launch), every further one is
// a TICKET drawn from a per-XCD counter (XCD x hands out positions x + 8 j of the list, j = 32, 33, ...), so workgroups that could not
// be placed right away -- a CU held by a weight-gradient tile of the second stream or by a collective's channel -- leave their share to
// the ones that are running instead of starting a three-tile walk late.  Producer wave 0 draws the tickets (one atomic, one tile ahead,
// its result consumed two K steps later, behind the counted waits) and hands them to the other waves through a 4-entry LDS queue.
// Restricted to what the multi-round launches of the step are: plain K-contiguous A, K a multiple of 64 and >= 384, N a multiple of the
// tile width, compute-dtype output, epilogue MODE 0 / 1 / 2 of gemm_epi.h (same arithmetic, same dropout indices: bit-identical results).
constexpr int PST_STG = NSTAGE * STAGE_BYTES, PST_TQ = PST_STG + 8 * 1024, PST_LDS = PST_TQ + 64;
constexpr int PST_END = 0x7fffffff;

// The persistent kernel feeds the MFMA with its operands swapped (B fragment first): the products and their summation order are the same, but a
// lane then holds FOUR CONSECUTIVE COLUMNS of one row -- acc[a][b][j] = C[a*16 + li][b*16 + lg*4 + j] -- so the staging writes are 8 bytes wide
// (32 per tile and wave instead of 64 two-byte ones) and the dropout index of an element is one add away from its neighbour's.
template <int NB, int MODE>   // MODE: 0 plain, 1 ReLU, 2 ReLU mask of relu_aux, 3 ReLU + dropout
__device__ __forceinline__ void epi_small(const EpiFast& e, const DropCfg dcfg, f32x4_t (&acc)[4][NB], unsigned stg, int mb, int nb, int lane) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    constexpr int CH = NB * 2, TASKS = 8 * CH;                        // one pass = 8 rows: 64 (48) tasks of one 16-byte chunk, <= 1 per lane
    const int li = lane & 15, lg = lane >> 4;
    const int trow = lane / CH, tch = lane - trow * CH;
    const bool tvalid = lane < TASKS;
    u32x4 qa[MODE == 2 ? 8 : 1];
    if (MODE == 2) {
        const bf16_t* aux = reinterpret_cast<const bf16_t*>(e.relu_aux);
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int m = min(mb + ps * 8 + min(trow, 7), e.M - 1);              // (rows past M: clamped load, never stored)
            qa[ps] = *reinterpret_cast<const u32x4*>(aux + (long)m * e.ld_aux + nb + min(tch, CH - 1) * 8);
        }
    }
    bf16_t* Ct = reinterpret_cast<bf16_t*>(e.C);
    // staging slice: 8 rows x 128 bytes, 16-byte chunk c of row r at chunk c ^ r (32 active lanes x 8 bytes = every bank once).  This lane's
    // four columns b*16 + lg*4 .. + 3 of pass row li & 7 are the (lg & 1) half of chunk 2b + lg/2: one address register, b*32 is XORed in -- the
    // XOR of a multiple of 32 commutes with the row swizzle, so it is (address ^ b*32), a compile-time constant per b
    const unsigned wa0 = stg + (unsigned)((li & 7) * 128 + ((((lg >> 1) ^ (li & 7)) << 4) | ((lg & 1) << 3)));
    const unsigned ra = stg + (unsigned)(trow * 128 + ((tch ^ (trow & 7)) << 4));
    // dropout: hash input of element (m, n) = (m N + n) C + key (b2s_keep) = x00 + a*16 (N C) [wave-uniform] + (b*16 + j) C [literal]
    constexpr uint32_t HC = 0x9E3779B1u;
    const uint32_t x00 = ((uint32_t)(mb + li) * (uint32_t)e.N + (uint32_t)(nb + lg * 4)) * HC + dcfg.key;
    const uint32_t nc = (uint32_t)e.N * HC;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        u32x2 pk[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = acc[a][b][j];
                if (MODE == 1 || MODE == 3) v[j] = fmaxf(v[j], 0.f);
                if (MODE == 3) v[j] = b2s_hash32(x00 + (uint32_t)(a * 16) * nc + (uint32_t)(b * 16 + j) * HC) >= dcfg.thresh ? v[j] * dcfg.scale : 0.f;
                if (MODE == 2) v[j] *= e.aux_scale;
            }
            pk[b][0] = f2bf2(v[0], v[1]); pk[b][1] = f2bf2(v[2], v[3]);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if ((li >> 3) == h) {                                     // rows a*16 + h*8 .. + 7 belong to the lanes with li / 8 == h
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const unsigned wa = wa0 ^ (unsigned)(b * 32);
                    asm volatile("ds_write_b64 %0, %1" ::"v"(wa), "v"(pk[b]) : "memory");
                }
            }
            // (same-wave LDS hand-off: DS operations of one wave complete in order)
            if (tvalid) {
                u32x4 o;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(o) : "v"(ra) : "memory");
                if (MODE == 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t w = qa[a * 2 + h][j];
                        const uint32_t lo = (w & 0xffffu) != 0 && !(w & 0x8000u) ? 0xffffu : 0u;
                        const uint32_t hi = (w >> 16) != 0 && !(w & 0x80000000u) ? 0xffff0000u : 0u;
                        o[j] &= (lo | hi);
                    }
                }
                const int m = mb + a * 16 + h * 8 + trow;
                if (m < e.M) *reinterpret_cast<u32x4*>(Ct + (long)m * e.ldc + nb + tch * 8) = o;
            }
        }
    }
}

template <bool TB, int NB>
__global__ __launch_bounds__(nthreads_of(0, 4), 1) void gemm_glds256_persist_kernel(GemmArgs g, int tiles_m, int tiles_n, int mode, int* tickets) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NP = nprod_of(0, 4), NC = 8, STG = STAGE_BYTES, A_B = A_BYTES, BN = NB * 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int wrow = (wave >> 1) * 64, wcol = (wave & 1) * (NB * 16);
    const char* Ab = reinterpret_cast<const char*>(g.A.p);
    const char* Bb = reinterpret_cast<const char*>(g.B.p);
    const int total = tiles_m * tiles_n, G = gridDim.x, xcd = (int)blockIdx.x & 7;
    const int nk = g.K / BK;
    // position v of the XCD-aware list -> tile origin
    auto tile_at = [&](int v, int& m0, int& n0) {
        const int id = xcd_tile_id(v, total);
        int by, bx;
        if (GROUP_M > 0 && tiles_n >= 16) {
            const int gsz = GROUP_M * tiles_n, grp = id / gsz, in = id - grp * gsz;
            const int rows = min(GROUP_M, tiles_m - grp * GROUP_M);
            bx = in / rows; by = grp * GROUP_M + (in - bx * rows);
        } else { by = id / tiles_n; bx = id - by * tiles_n; }
        m0 = by * BM; n0 = bx * BN;
    };
    const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)smem_raw;
    // T[t]: list position of this workgroup's t-th tile (PST_END: none).  T[0] = blockIdx.x; without a ticket buffer T[t] = blockIdx.x + t G.
    // The queue entry of T[t], t >= 1, is written by producer wave 0 at least two workgroup barriers before anybody reads it.
    // (explicit DS instructions: through a generic pointer the compiler emits FLAT accesses, which count on vmcnt and drain the producers' DMA queue)
    const unsigned tq = lds_base + (unsigned)PST_TQ;
    const bool dyn = tickets != nullptr && total > G;          // (a one-round launch draws nothing: T[1] is the end for everybody)
    auto tq_put = [&](int t, int val) {       // one lane
        asm volatile("ds_write_b32 %0, %1" ::"v"(tq + (unsigned)((t & 3) * 4)), "v"(val) : "memory");
    };
    auto queued = [&](int t) -> int {
        if (!dyn || t < 1) { const int v = (int)blockIdx.x + t * G; return v < total ? v : PST_END; }
        int val;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(val) : "v"(tq + (unsigned)((t & 3) * 4)) : "memory");
        return __builtin_amdgcn_readfirstlane(val);
    };

    if (wave >= NC) {
        // ---- producer waves: one flat sequence of K steps over all tiles
        constexpr int NBI = (NB == 4 || TB) ? 16 : 12, NIA = (BM / 8) / NP, NIB = NBI / NP, IPW = NIA + NIB;
        const int iw = wave - NC, ia0 = iw * NIA, ib0 = iw * NIB;
        const bool drawer = dyn && iw == 0;
        const long stepA = 2L * BK, stepB = TB ? 2L * BK * g.B.ld : 2L * BK;        // bytes per K step
        unsigned goffA[NIA], goffB[NIB];
        auto set_tile = [&](int v) {
            int m0, n0; tile_at(v, m0, n0);
#pragma unroll
            for (int i = 0; i < NIA; ++i) {
                const LaneSrc s = lane_src<false, true>(ia0 + i, lane);
                goffA[i] = 2u * (unsigned)((long)min(m0 + s.r, g.A.R - 1) * g.A.ld + s.c);
            }
#pragma unroll
            for (int i = 0; i < NIB; ++i) {
                const LaneSrc s = lane_src<TB, false>(ib0 + i, lane);
                if (TB) goffB[i] = 2u * (unsigned)((long)s.r * g.B.ld + min(n0 + s.c, max(g.B.C - 8, 0)));
                else    goffB[i] = 2u * (unsigned)((long)min(n0 + s.r, g.B.R - 1) * g.B.ld + s.c);
            }
        };
        // ticket draw (drawer wave, lane 0): an atomic whose result register is read two K steps later -- the counted waits of the steps in
        // between (loads return in order, the atomic is older than the loads they leave in flight) have retired it by then
        int tk_raw = 0;                   // j of the draw in flight
        int tk_for = -1;                  // ... and the tile index t it is for (-1: none)
        int tk_age = 0;
        bool drew_end = false;
        auto draw = [&](int t) {
            if (!drawer || drew_end) return;
            if (lane == 0) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(tk_raw) : "v"(tickets + xcd), "v"(1) : "memory");
            tk_for = t; tk_age = 0;
        };
        auto land = [&]() {               // called once per K step
            if (!drawer || tk_for < 0) return;
            if (++tk_age < 2) return;
            const int j = __builtin_amdgcn_readfirstlane(tk_raw);
            const int v = xcd + 8 * (G / 8 + j);
            const int val = v < total ? v : PST_END;
            if (val == PST_END) drew_end = true;
            if (lane == 0) tq_put(tk_for, val);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            tk_for = -1;
        };
        int pt = 0, pk = 0;
        bool more = true;                 // a stage is left to issue
        set_tile((int)blockIdx.x);
        draw(1);                                                  // (T[0] is static)
        int issued = 0;
        auto issue_next = [&](int slot) {
            unsigned char* sbase = smem_raw + slot * STG;
            const char* sa = Ab + pk * stepA;
            const char* sb = Bb + pk * stepB;
#pragma unroll
            for (int i = 0; i < NIA; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(sa + goffA[i]), (lptr_t)(sbase + (ia0 + i) * 1024), 16, 0, B2S_DMA_AUX);
#pragma unroll
            for (int i = 0; i < NIB; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(sb + goffB[i]), (lptr_t)(sbase + A_B + (ib0 + i) * 1024), 16, 0, B2S_DMA_AUX);
            ++issued;
            if (++pk == nk) {
                pk = 0; ++pt;
                const int v = queued(pt);
                if (v == PST_END) more = false;
                else { set_tile(v); draw(pt + 1); }
            }
        };
        // (nk >= 6: the three prologue stages are all of tile 0, and T[1] is in the queue long before the first wrap)
#pragma unroll
        for (int p = 0; p < NSTAGE; ++p) issue_next(p);
        wait_vm<2 * IPW>();
        __builtin_amdgcn_s_barrier();
        int slot = 0;
        for (int s = 0; s < issued; ++s) {
            if (issued - s >= 3) wait_vm<IPW>(); else wait_vm<0>();          // stage s + 1 landed (stage s + 2 may still fly)
            land();
            __builtin_amdgcn_s_barrier();
            if (more) issue_next(slot);
            slot = slot + 1 == NSTAGE ? 0 : slot + 1;
        }
        if (drawer && lane == 0) {
            // the last workgroup to get here (every drawer has seen the end of its XCD's list by now) re-arms the counters for the launch
            // that uses this set next (csrc: a pool of sets handed out round-robin)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (atomicAdd(tickets + 8, 1) == G - 1) {
#pragma unroll
                for (int x = 0; x < 9; ++x) tickets[x] = 0;
            }
        }
        return;
    }

    // ---- MFMA waves
    unsigned offA[2][4], offB[2][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int r = wrow + t * 16 + li;
        offA[0][t] = (unsigned)(r * 128 + ((lg ^ swz_n(r)) << 4)); offA[1][t] = offA[0][t] ^ 64u;
        if (TB) {
            const int k = lg * 8 + (li >> 2), col = wcol + t * 16 + (li & 3) * 4, sw = ((li >> 2) | ((lg & 1) << 2)) << 1;
            const unsigned o = (unsigned)A_B + 2u * (k * 128 + (((col >> 3) ^ sw) << 3) + (col & 7));
            offB[0][t] = o; offB[1][t] = o + UNIT;
        } else {
            const int rb = wcol + t * 16 + li;
            offB[0][t] = (unsigned)(A_B + rb * 128 + ((lg ^ swz_n(rb)) << 4)); offB[1][t] = offB[0][t] ^ 64u;
        }
    }
    auto frag_issue = [&](bf16x8_t& dst, bool trans, unsigned addr) {
        if (!trans) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
        } else {
            bf16x4_t lo, hi;
            asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:1024"
                         : "=&v"(lo), "=&v"(hi) : "v"(addr) : "memory");
            dst[0] = lo[0]; dst[1] = lo[1]; dst[2] = lo[2]; dst[3] = lo[3];
            dst[4] = hi[0]; dst[5] = hi[1]; dst[6] = hi[2]; dst[7] = hi[3];
        }
    };
#define B2S_MMA16(CA, CB)                                                                                         \
    _Pragma("unroll") for (int a = 0; a < 4; ++a) _Pragma("unroll") for (int b = 0; b < NB; ++b)                  \
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(CB[b], CA[a], acc[a][b], 0, 0, 0);     /* operands swapped: epi_small */
#define B2S_READ8(FA, FB, SLOT, H)                                                                                \
    {                                                                                                              \
        const unsigned sb_ = lds_base + (unsigned)((SLOT) * STG);                                                  \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                            \
            frag_issue(FA[t], false, sb_ + offA[H][t]);                                                            \
            if (t < NB) frag_issue(FB[t], TB, sb_ + offB[H][t]);                                                   \
        }                                                                                                          \
    }
    const EpiFast ef = {g.C, g.ldc, g.M, g.N, nullptr, 0, g.epi.relu_aux, g.epi.ld_aux, g.epi.aux_scale, g.epi.drop, g.epi.drop_salt};
    DropCfg dcfg = g.epi.drop;
    if (dcfg.thresh && g.epi.drop_salt) dcfg.key ^= b2s_hash32((uint32_t)(*g.epi.drop_salt) * 2246822519u + 3266489917u);
    const unsigned stg = lds_base + (unsigned)(PST_STG + wave * 1024);
    f32x4_t acc[4][NB];
    bf16x8_t fa0[4], fb0[4], fa1[4], fb1[4];
    __builtin_amdgcn_s_barrier();                                      // stage 0 of the first tile has landed
    int slot = 0;
    int v = (int)blockIdx.x;
    for (int t = 0; v != PST_END; ++t) {
        int m0, n0; tile_at(v, m0, n0);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        B2S_READ8(fa0, fb0, slot, 0)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        for (int kt = 0; kt < nk; ++kt) {
            const int nslot = slot + 1 == NSTAGE ? 0 : slot + 1;
            B2S_READ8(fa1, fb1, slot, 1)
            __builtin_amdgcn_sched_barrier(0);
            B2S_MMA16(fa0, fb0)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // (the last step of a tile reads the first half of the NEXT tile's stage 0 here and drops it: the loop body stays the tuned one,
            // the epilogue below gets the registers)
            B2S_READ8(fa0, fb0, nslot, 0)
            __builtin_amdgcn_sched_barrier(0);
            B2S_MMA16(fa1, fb1)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            slot = nslot;
        }
        __builtin_amdgcn_sched_barrier(0);
        v = queued(t + 1);                                             // (written two or more barriers ago)
        if (mode == 0) epi_small<NB, 0>(ef, dcfg, acc, stg, m0 + wrow, n0 + wcol, lane);
        else if (mode == 1 && !dcfg.thresh) epi_small<NB, 1>(ef, dcfg, acc, stg, m0 + wrow, n0 + wcol, lane);
        else if (mode == 1) epi_small<NB, 3>(ef, dcfg, acc, stg, m0 + wrow, n0 + wcol, lane);
        else epi_small<NB, 2>(ef, dcfg, acc, stg, m0 + wrow, n0 + wcol, lane);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef B2S_READ8
#undef B2S_MMA16
}

// which epilogue MODE of gemm_epi.h a launch takes when every wave is on the fast path (-1: none of 0 / 1 / 2)
inline int persist_mode(const GemmArgs& g) {
    const GemmEpilogue& e = g.epi;
    if (g.c_fp32 || e.residual || e.bias || e.row_len || e.alpha != 1.f || e.kv_k || e.accumulate || e.conv_dw_cin || e.colstat) return -1;
    if ((g.ldc & 7) || (reinterpret_cast<uintptr_t>(g.C) & 15) || g.cs_o || g.cs_i) return -1;
    if (!e.relu && !e.relu_aux && !e.drop.thresh) return 0;
    if (e.relu && !e.relu_aux) return 1;
    if (e.relu_aux && !e.relu && !e.drop.thresh && (e.ld_aux & 7) == 0 && (reinterpret_cast<uintptr_t>(e.relu_aux) & 15) == 0) return 2;
    return -1;
}
#ifdef B2S_LAB
static const int g_persist = getenv("B2S_LAB_GEMM_PERSIST") ? atoi(getenv("B2S_LAB_GEMM_PERSIST")) : 1;
static const int g_persist_min = getenv("B2S_LAB_GEMM_PERSIST_MIN") ? atoi(getenv("B2S_LAB_GEMM_PERSIST_MIN")) : 1;      // fewest tiles of a persistent launch
#else
constexpr int g_persist = 1;
// one-round launches take the kernel too (one tile per workgroup: its epilogue has no workgroup barrier and half the staging instructions;
// step 7.34 -> 7.29 ms, three interleaved rounds)
constexpr int g_persist_min = 1;
#endif
// Ticket counters of the persistent launches: a pool of 1024 sets (8 per-XCD counters + 1 arrival counter each, 64 bytes apart), zeroed once and
// re-armed by the last workgroup of every launch that used a set; sets go round-robin, so a set is next used 1024 persistent launches (~12 steps) later --
// far more than can be in flight on the streams of a process.  Device memory of the process, like the zero page.
int* persist_ticket_set() {
    // one pool per DEVICE (device memory is not addressable from another device's kernels unless peer access happens to be on): the launch takes the
    // pool of the device that is current on the launching thread -- the device the stream belongs to in every caller of this library
    constexpr int MAXDEV = 16;
    static int* pool[MAXDEV] = {};
    static std::once_flag once[MAXDEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) { (void)hipGetLastError(); return nullptr; }     // (no pool: static tile assignment)
    std::call_once(once[dev], [dev] {
        void* p = nullptr;
        if (hipMalloc(&p, 1024 * 64) == hipSuccess && hipMemset(p, 0, 1024 * 64) == hipSuccess) pool[dev] = (int*)p;
        else (void)hipGetLastError();
    });
    static std::atomic<unsigned> next{0};
    return pool[dev] ? pool[dev] + (next.fetch_add(1, std::memory_order_relaxed) & 1023u) * 16 : nullptr;
}
template <bool TB, int NB>
int launch256_persist(const GemmArgs& g, int mode, hipStream_t stream) {
    constexpr size_t smem = PST_LDS;               // 144 KB ring + 8 KB of per-wave staging + the ticket queue
    static std::once_flag attr_once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(attr_once, [] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_glds256_persist_kernel<TB, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    });
    B2S_HIP(attr_err);
    const int tiles_m = cdiv(g.M, BM), tiles_n = g.N / (NB * 32);
    int* tickets = (g_persist == 2 || tiles_m * tiles_n <= 256) ? nullptr : persist_ticket_set();      // (a one-round launch draws nothing)
    const int grid = std::min(256, tiles_m * tiles_n);             // (a multiple of 8: try_persist)
    hipLaunchKernelGGL((gemm_glds256_persist_kernel<TB, NB>), dim3(grid), dim3(nthreads_of(0, 4)), smem, stream, g, tiles_m, tiles_n, mode, tickets);
    B2S_LAUNCH_CHECK();
    return 0;
}
// >= 0: launched (or failed) through the persistent kernel; -1: not eligible
template <bool TA, bool TB, int GATHER, int NB>
int try_persist(const GemmArgs& g, hipStream_t stream) {
    if (TA || GATHER != 0 || !g_persist) return -1;
    const long tiles = (long)cdiv(g.M, BM) * cdiv(g.N, NB * 32);
    if (tiles < g_persist_min || (tiles < 256 && (tiles & 7)) || tiles >= (1 << 24) || g.batch != 1 || g.splitk != 1 || g.K < 6 * BK || g.K % BK != 0 || g.N % (NB * 32) != 0) return -1;
    if ((long)g.A.R * g.A.ld >= (1L << 30) || (long)g.B.R * g.B.ld >= (1L << 30) || g.A.g_cin || g.B.g_cin) return -1;
    if (TB && !(g.B.C >= 8 && (g.B.C & 7) == 0)) return -1;
    const int mode = persist_mode(g);
    if (mode < 0) return -1;
    return launch256_persist<TB, NB>(g, mode, stream);
}

template <bool TA, bool TB, int GATHER, int NB, int MW = 4>
int launch256_nb(const GemmArgs& g_in, const bf16_t* zero, hipStream_t stream) {
    GemmArgs g = g_in;
    if (g.splitk > 1) {                     // every split must own at least one K step (empty splits would leave slabs unwritten)
        const int nk_all = cdiv(g.K, BK), per = cdiv(nk_all, g.splitk);
        g.splitk = cdiv(nk_all, per);
    }
    constexpr size_t smem = (size_t)NSTAGE * (MW * 64 * BK * 2 + B_BYTES);           // 144 KB (96 KB with 128-row tiles)
    static std::once_flag attr_once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(attr_once, [] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_glds256_kernel<TA, TB, GATHER, NB, MW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    });
    B2S_HIP(attr_err);
    const int tiles_m = cdiv(g.M, MW * 64), tiles_n = cdiv(g.N, NB * 32);
    dim3 grid(tiles_m * tiles_n * g.batch * g.splitk);               // 1-D: the kernel maps linear ids to tiles (XCD-aware)
    float* ws = nullptr;
    if (g.splitk > 1 && g.batch == 1 && g.c_fp32 && g.epi.accumulate && (g.N & 3) == 0 && (g.ldc & 3) == 0 &&
        g.ws && (size_t)g.splitk * g.M * g.N <= g.ws_floats)
        ws = g.ws;
    hipLaunchKernelGGL((gemm_glds256_kernel<TA, TB, GATHER, NB, MW>), grid, dim3(nthreads_of(GATHER, MW)), smem, stream, g, zero, ws, tiles_m, tiles_n);
    B2S_LAUNCH_CHECK();
    if (ws) B2S_TRY(b2s_splitk_reduce_launch(ws, (float*)g.C, g.M, g.N, g.ldc, g.splitk, g.epi.conv_dw_cin, stream));
    return 0;
}

// BN = 96 when that needs fewer (or cheaper) rounds of one workgroup per CU: N = 768 / 2304 with M = 8148 give exactly
// 256 / 768 tiles of 256x96, against 192 / 576 tiles of 256x128 (a quarter of the CUs idle)
std::atomic<int> g_tile_policy{0};          // b2s_gemm_set_tile_policy
inline int pick_nb(const GemmArgs& g, bool allow64 = false) {
    static const int env_force = getenv("B2S_GEMM256_NB") ? atoi(getenv("B2S_GEMM256_NB")) : 0;
    const int force = env_force ? env_force : g_tile_policy.load(std::memory_order_relaxed);
    if (force == 3 || force == 4) return force;
    const long per = (long)cdiv(g.M, BM) * g.batch * std::max(1, g.splitk);
    const long r128 = (per * cdiv(g.N, 128) + 255) / 256 * 128, r96 = (per * cdiv(g.N, 96) + 255) / 256 * 101;   // 96 * 1.05
    // 64-column tiles (aligned conv gather only): the postnet's 512-channel layers are 32 x 8 = exactly 256 tiles of 256 x 64, against 192 tiles of 256 x 96
    // whose last column panel is one third full
    if (allow64) {
        const long r64 = (per * cdiv(g.N, 64) + 255) / 256 * 72;
        if (r64 < std::min(r96, r128)) return 2;
    }
    return r96 < r128 ? 3 : 4;
}
template <bool TA, bool TB, int GATHER>
int launch256_t(const GemmArgs& g, const bf16_t* zero, hipStream_t stream) {
    if (!TA && GATHER == 0) {
        // 128-row tiles when 256-row tiles would occupy at most half of the CUs
#ifdef B2S_LAB
        static const int small_tiles = getenv("B2S_LAB_SMALL_TILES") ? atoi(getenv("B2S_LAB_SMALL_TILES")) : 128;
#else
        constexpr int small_tiles = 128;
#endif
        const int nb = pick_nb(g);
        const long t256n = (long)cdiv(g.M, BM) * cdiv(g.N, nb * 32) * g.batch * std::max(1, g.splitk);
        if (t256n <= small_tiles && g.M > 128) {
            constexpr bool ta = false;          // (the 128-row variant is instantiated for K-contiguous A only)
            return nb == 3 ? launch256_nb<ta, TB, 0, 3, 2>(g, zero, stream)
                           : launch256_nb<ta, TB, 0, 4, 2>(g, zero, stream);
        }
    }
#ifdef B2S_LAB
    static const bool lab_no64 = getenv("B2S_LAB_NO_NB2") != nullptr;
#else
    constexpr bool lab_no64 = false;
#endif
    const int nb = pick_nb(g, GATHER == 2 && !TB && !lab_no64);
    if constexpr (GATHER == 2 && !TB) {
        if (nb == 2) return launch256_nb<TA, TB, GATHER, 2>(g, zero, stream);
    }
    const int rc = nb == 3 ? try_persist<TA, TB, GATHER, 3>(g, stream) : try_persist<TA, TB, GATHER, 4>(g, stream);
    if (rc >= 0) return rc;
    return nb == 3 ? launch256_nb<TA, TB, GATHER, 3>(g, zero, stream)
                   : launch256_nb<TA, TB, GATHER, 4>(g, zero, stream);
}

}  // namespace t256

int b2s_gemm_glds256_grouped_launch(const GemmArgs* probs, int n, const bf16_t* zero, hipStream_t stream) {
    B2S_CHECK(n >= 1 && n <= B2S_MAX_GROUP, "grouped GEMM: %d problems (max %d)", n, B2S_MAX_GROUP);
    b2s_gemm_group grp;
    grp.n = n;
    constexpr int xcd_order = 1;
    grp.order = xcd_order;
    // Launch order = dispatch order.  Tiles cost ~K; with one workgroup per CU the makespan of "deepest first" against
    // "shallowest first" is decided by list scheduling on the CU count (288 tiles of a decoder layer: 252 deep + 36 shallow --
    // shallow first lets the 36 early finishers take the last 32 deep tiles, deep first leaves 32 shallow tiles to 4 CUs)
    int order[B2S_MAX_GROUP];
    auto makespan = [&](bool deep_first) {
        for (int i = 0; i < n; ++i) order[i] = i;
        std::stable_sort(order, order + n, [&](int a, int b) { return deep_first ? probs[a].K > probs[b].K : probs[a].K < probs[b].K; });
        std::vector<long> cu(256, 0);                 // min-heap of CU finish times
        auto cmp = [](long a, long b) { return a > b; };
        long end = 0;
        for (int k = 0; k < n; ++k) {
            const GemmArgs& g = probs[order[k]];
            const int t = cdiv(g.M, t256::BM) * cdiv(g.N, 128);
            for (int j = 0; j < t; ++j) {
                std::pop_heap(cu.begin(), cu.end(), cmp);
                cu.back() += g.K + 1024;               // K steps + fixed per-tile cost, in units of one K element
                end = std::max(end, cu.back());
                std::push_heap(cu.begin(), cu.end(), cmp);
            }
        }
        return end;
    };
    const long deep = makespan(true), shallow = makespan(false);
    if (deep <= shallow) makespan(true);               // (leaves `order` as chosen)
    int tiles = 0;
    constexpr int split = 1;
    for (int k = 0; k < n; ++k) {
        const GemmArgs& g = probs[order[k]];
        B2S_CHECK(g.batch == 1 && g.c_fp32 && g.A.g_cin == 0 && g.B.g_cin == 0, "grouped GEMM: problem %d is not a plain fp32 dW", order[k]);
        B2S_CHECK(g.epi.accumulate || split == 1, "grouped GEMM: split K needs an accumulating output (problem %d overwrites)", order[k]);
        // B2S_DW_SPLIT = 2: every tile's K walk in two halves.  The weight-gradient groups share the chip with the backward's main-stream
        // kernels; ~290 tiles of 127 K steps each on fewer than 256 free CUs need a second full-length round, half-length units pack better.
        grp.p[k] = g; grp.p[k].splitk = (split > 1 && g.K >= 4 * split * t256::BK) ? split : 1;
        grp.tile0[k] = tiles;
        tiles += cdiv(g.M, t256::BM) * cdiv(g.N, 128) * grp.p[k].splitk;
    }
    grp.tile0[n] = tiles;
    constexpr size_t smem = (size_t)t256::NSTAGE * t256::STAGE_BYTES;
    static std::once_flag attr_once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(attr_once, [] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(t256::gemm_glds256_grouped_kernel<4>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    });
    B2S_HIP(attr_err);
    hipLaunchKernelGGL((t256::gemm_glds256_grouped_kernel<4>), dim3(tiles), dim3(t256::nthreads_of(false)), smem, stream, grp, zero);
    B2S_LAUNCH_CHECK();
    return 0;
}

// 0: per-shape choice between 256x96 and 256x128 tiles (fewest rounds of one workgroup per CU on 256 CUs); 4: 256x128 everywhere.
// The per-shape choice makes N = 768 / 2304 GEMMs exactly 256 / 768 tiles -- one or three FULL rounds, which become two / four as soon as
// another kernel holds a few CUs (a communication library's channels): measured +11 % on the step with 8 CUs held, +4 % with 256x128 tiles
// (192 / 576 tiles), at no cost with all CUs free (profiles/NOTES_r03.md).  The data-parallel trainer selects 4.
extern "C" int b2s_gemm_set_tile_policy(int policy) {
    if (policy != 0 && policy != 3 && policy != 4) return b2s_fail(__FILE__, __LINE__, "tile policy must be 0 (auto), 3 (256x96) or 4 (256x128), got %d", policy);
    t256::g_tile_policy.store(policy, std::memory_order_relaxed);
    return 0;
}

// number of 256x128 tiles a problem decomposes into (the dispatcher in gemm_glds.hip uses it to pick the tile shape)
long b2s_gemm_glds256_tiles(const GemmArgs& g) { return (long)cdiv(g.M, t256::BM) * cdiv(g.N, 128) * g.batch * std::max(1, g.splitk); }

int b2s_gemm_glds256_launch(const GemmArgs& g, bool ta, bool tb, const bf16_t* zero, hipStream_t stream) {
    const bool gather = g.A.g_cin > 0 || g.B.g_cin > 0;
    if (gather) {
        if (!ta && !tb) {
            // forward / backward-data convolution over >= 64-aligned channel counts: aligned gather on the producer waves
            const bool aligned = g.A.g_cin > 0 && g.B.g_cin == 0 && g.A.g_cin % t256::BK == 0 && g.K % t256::BK == 0 && g.splitk == 1 &&
                                 (long)g.A.R * g.A.ld < (1L << 29) && (long)g.B.R * g.B.ld < (1L << 30) && g.A.g_T > 0;
            constexpr bool no_fast = false;      // A/B switch
            if (aligned && !no_fast) return t256::launch256_t<false, false, 2>(g, zero, stream);
            return t256::launch256_t<false, false, 1>(g, zero, stream);
        }
        if (ta && tb) return t256::launch256_t<true, true, 1>(g, zero, stream);
        return b2s_fail(__FILE__, __LINE__, "conv gather is supported for the NT and TN forms only");
    }
    if (!ta && !tb) return t256::launch256_t<false, false, 0>(g, zero, stream);
    if (!ta && tb) return t256::launch256_t<false, true, 0>(g, zero, stream);
    if (ta && !tb) return t256::launch256_t<true, false, 0>(g, zero, stream);
    return t256::launch256_t<true, true, 0>(g, zero, stream);
}
