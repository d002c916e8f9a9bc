// Epilogue shared by the bf16 LDS-DMA GEMM kernels (gemm_glds.hip: 128x128 tiles, gemm_glds256.hip: 256x128 tiles): one
// wave owns a 64 x (NB*16) accumulator block at (mb, nb) in the MFMA layout and a private 16 KB LDS staging area `stg`.
#pragma once
#include "gemm.h"

// 16-byte global store.  (Measured: write-through `sc1` stores, meant to spare the end-of-kernel L2 write-back, gain 1 us on
// the bf16-output GEMMs and lose 5 us on the fp32 residual ones -- plain stores stay.)
__device__ __forceinline__ void epi_store16(void* p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }
// Column sums of a wave's accumulator block (MFMA C layout: column nb + b*16 + (lane & 15), rows mb + a*16 + (lane >> 4)*4 + r) added to
// colstat[n] / colstat[N + n]: two shuffles fold the four 16-lane groups, one atomic pair per column and wave.
template <int NBv>
__device__ __forceinline__ void epi_colstat(const f32x4_t (&acc)[4][NBv], float* colstat, float alpha, int M, int N, int mb, int nb, int lane) {
    const int li = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int b = 0; b < NBv; ++b) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = (mb + a * 16 + lg * 4 + r) < M ? acc[a][b][r] * alpha : 0.f;
                s1 += v; s2 += v * v;
            }
        s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
        const int n = nb + b * 16 + li;
        if (lg == 0 && n < N) { atomicAdd(colstat + n, s1); atomicAdd(colstat + N + n, s2); }
    }
}

// Fast paths for the epilogues the training step is made of (a wave's 64 x NB*16 block completely inside the matrix, 16-byte aligned
// rows, alpha = 1, no bias / row masking / accumulation):
//   MODE 0  compute-dtype output, plain                          (qkv / q projections, every dX GEMM)
//   MODE 1  compute-dtype output, ReLU [+ dropout]               (FFN input layer)
//   MODE 2  compute-dtype output, ReLU-mask of `relu_aux`        (FFN dX: d hidden)
//   MODE 3  fp32 output = dropout(acc) + fp32 residual           (attention output / FFN output projections: the residual stream)
//   MODE 4  fp32 output, plain                                   (postnet convolutions: the BatchNorm input)
// No control flow between the first load and the last store: the generic epilogue below branches around every optional operand, and
// hipcc waits vmcnt(0) at the join of every block that contains a load or a store -- its passes ran as a chain of memory round trips.
// Here the residual / mask rows of ALL of a lane's tasks are requested before the accumulators are staged through LDS, every lane task
// is one row x 8 consecutive columns (NB*2 tasks per lane, consecutive lanes = consecutive 16-byte chunks of a row), stores fire
// back to back.
#ifndef B2S_EPI_FAST
#define B2S_EPI_FAST 1
#endif
struct EpiFast {                 // the few fields of GemmArgs the fast paths read, by value (a reference to the kernel's GemmArgs copy kept
    void* C;                     // the whole structure addressable: 500+ bytes of scratch per lane)
    int ldc, M, N;
    const float* residual; int ldr;
    const void* relu_aux; int ld_aux; float aux_scale;
    DropCfg drop; const int* drop_salt;
};
template <int NB, int MODE>
__device__ __forceinline__ void gemm_wave_epilogue_fast(const EpiFast g, f32x4_t (&acc)[4][NB], float* stg, int mb, int nb, int lane, int z, long cbase) {
    constexpr int CH = NB * 2;                           // 8-column chunks per block row = tasks per lane
    const int li = lane & 15, lg = lane >> 4;
    const EpiFast& e = g;
    int trow[CH], tcol[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) { const int id = k * 64 + lane; trow[k] = id / CH; tcol[k] = (id - trow[k] * CH) * 8; }
    typedef uint32_t epi_u32x4 __attribute__((ext_vector_type(4)));      // (first-class vector values: arrays of HIP's float4 / uint4 structs ended up in scratch)
    f32x4_t r0[MODE == 3 ? CH : 1], r1[MODE == 3 ? CH : 1];
    epi_u32x4 qa[MODE == 2 ? CH : 1];
    // (128-column tiles: 16 residual vectors + 16 accumulator blocks do not fit the 168-register budget of the 12-wave workgroup --
    // there the residual rows are requested right after the accumulators have been staged)
    constexpr bool EARLY = NB <= 3;
    auto load_res = [&]() {
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const float* rp = e.residual + (long)(mb + trow[k]) * e.ldr + nb + tcol[k];
            r0[k] = *reinterpret_cast<const f32x4_t*>(rp); r1[k] = *reinterpret_cast<const f32x4_t*>(rp + 4);
        }
    };
    if (MODE == 3 && EARLY) load_res();
    if (MODE == 2) {
        const bf16_t* aux = reinterpret_cast<const bf16_t*>(e.relu_aux);
#pragma unroll
        for (int k = 0; k < CH; ++k) qa[k] = *reinterpret_cast<const epi_u32x4*>(aux + (long)(mb + trow[k]) * e.ld_aux + nb + tcol[k]);
    }
    __builtin_amdgcn_s_barrier();                          // every wave is done reading the operand ring
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = a * 16 + lg * 4 + r;
#pragma unroll
            for (int b = 0; b < NB; ++b) stg[row * 64 + ((b * 16 + li) ^ (((row >> 2) & 1) << 4))] = acc[a][b][r];
        }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 3 && !EARLY) load_res();
    DropCfg dcfg = e.drop;
    if (e.drop.thresh && e.drop_salt) dcfg.key ^= b2s_hash32((uint32_t)(*e.drop_salt) * 2246822519u + 3266489917u);
    float* Cf = reinterpret_cast<float*>(g.C);
    bf16_t* Ct = reinterpret_cast<bf16_t*>(g.C);
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const int row = trow[k], m = mb + row, n = nb + tcol[k];
        const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(stg + row * 64 + (tcol[k] ^ (((row >> 2) & 1) << 4)));
        const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(stg + row * 64 + (tcol[k] ^ (((row >> 2) & 1) << 4)) + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        if (MODE == 2) {
            const uint32_t w[4] = {qa[k][0], qa[k][1], qa[k][2], qa[k][3]};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t bits = (w[j >> 1] >> ((j & 1) * 16)) & 0xffffu;      // bf16 > 0  <=>  sign clear and non-zero
                v[j] = (bits != 0 && !(bits & 0x8000u)) ? v[j] * e.aux_scale : 0.f;
            }
        }
        if ((MODE == 1 || MODE == 3) && dcfg.thresh) {     // (kernel-argument condition: scalar branch, no memory operation inside)
            const uint32_t idx = (uint32_t)(((long)z * g.M + m) * g.N + n);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = b2s_keep(dcfg, idx + j) ? v[j] * dcfg.scale : 0.f;
        }
        const long off = cbase + (long)m * g.ldc + n;
        if (MODE == 4) {
            const f32x4_t o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
            *reinterpret_cast<f32x4_t*>(Cf + off) = o0; *reinterpret_cast<f32x4_t*>(Cf + off + 4) = o1;
        } else if (MODE == 3) {
            const f32x4_t o0 = {v[0] + r0[k][0], v[1] + r0[k][1], v[2] + r0[k][2], v[3] + r0[k][3]};
            const f32x4_t o1 = {v[4] + r1[k][0], v[5] + r1[k][1], v[6] + r1[k][2], v[7] + r1[k][3]};
            *reinterpret_cast<f32x4_t*>(Cf + off) = o0; *reinterpret_cast<f32x4_t*>(Cf + off + 4) = o1;
        } else {
            const epi_u32x4 o = {f2bf2(v[0], v[1]), f2bf2(v[2], v[3]), f2bf2(v[4], v[5]), f2bf2(v[6], v[7])};
            *reinterpret_cast<epi_u32x4*>(Ct + off) = o;
        }
    }
}

template <int NB>     // the wave's block is 64 rows x NB*16 columns
__device__ __forceinline__ void gemm_wave_epilogue(GemmArgs g, f32x4_t (&acc)[4][NB], float* stg, int mb, int nb, int lane, int z, int zo, int zi,
                                          int ksplit, float* splitk_ws) {
    const int li = lane & 15, lg = lane >> 4;
    if (g.epi.colstat && g.splitk == 1) epi_colstat<NB>(acc, g.epi.colstat, g.epi.alpha, g.M, g.N, mb, nb, lane);
#if B2S_EPI_FAST
    {
        const GemmEpilogue& e = g.epi;
        const long cb = zo * g.cs_o + zi * g.cs_i;
        // (all wave-uniform: kernel arguments and the wave's block origin)
        const bool common = g.splitk == 1 && e.conv_dw_cin == 0 && !e.accumulate && !e.bias && !e.row_len && e.alpha == 1.f && !e.kv_k &&
                            mb + 64 <= g.M && nb + NB * 16 <= g.N && (g.ldc & 7) == 0 && (cb & 7) == 0 &&
                            (reinterpret_cast<uintptr_t>(g.C) & 15) == 0;
        if (common) {
            const EpiFast f = {g.C, g.ldc, g.M, g.N, e.residual, e.ldr, e.relu_aux, e.ld_aux, e.aux_scale, e.drop, e.drop_salt};
            if (!g.c_fp32 && !e.residual) {
                if (!e.relu && !e.relu_aux && !e.drop.thresh) { gemm_wave_epilogue_fast<NB, 0>(f, acc, stg, mb, nb, lane, z, cb); return; }
                if (e.relu && !e.relu_aux) { gemm_wave_epilogue_fast<NB, 1>(f, acc, stg, mb, nb, lane, z, cb); return; }
                if (e.relu_aux && !e.relu && !e.drop.thresh && (e.ld_aux & 7) == 0 && (reinterpret_cast<uintptr_t>(e.relu_aux) & 15) == 0) {
                    gemm_wave_epilogue_fast<NB, 2>(f, acc, stg, mb, nb, lane, z, cb); return;
                }
            } else if (g.c_fp32 && e.residual && !e.relu && !e.relu_aux && (e.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(e.residual) & 15) == 0) {
                gemm_wave_epilogue_fast<NB, 3>(f, acc, stg, mb, nb, lane, z, cb); return;
            } else if (g.c_fp32 && !e.residual && !e.relu && !e.relu_aux && !e.drop.thresh) {
                gemm_wave_epilogue_fast<NB, 4>(f, acc, stg, mb, nb, lane, z, cb); return;
            }
        }
    }
#endif
    // ---------------- epilogue.  The accumulators (MFMA layout: col = lane & 15, row = (lane >> 4)*4 + r) are staged
    // through LDS (the operand ring is dead now) so that every lane owns 8 consecutive output columns of one row:
    // bf16 results leave as 16-byte stores, residual / ReLU-mask / bias operands arrive as 16-byte loads.
    GemmEpilogue e = g.epi;
    if (g.splitk > 1 && splitk_ws) {
        // split-K partial tile: plain vector stores into workspace slab `ksplit` ([splitk][M][N] fp32); a reduce kernel
        // adds the slabs into the gradient (fp32 atomics cost ~8 ns per 64-lane instruction and were the bottleneck)
        g.C = splitk_ws + (long)ksplit * g.M * g.N; g.c_fp32 = 1; g.ldc = g.N; g.cs_o = 0; g.cs_i = 0; g.splitk = 1;
        e.accumulate = 0; e.conv_dw_cin = 0;
    }
    if (g.splitk > 1 || e.conv_dw_cin > 0) {
        // weight-gradient forms: linear fp32 accumulate straight from the MFMA layout (16 lanes = 64 contiguous bytes
        // per atomic / store instruction)
        float* Cw = reinterpret_cast<float*>(g.C) + zo * g.cs_o + zi * g.cs_i;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mb + a * 16 + lg * 4 + r;
                if (m >= g.M) continue;
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    int nn = nb + b * 16 + li;
                    if (nn >= g.N) continue;
                    if (e.conv_dw_cin > 0) { int jj = nn / e.conv_dw_cin; nn = (nn - jj * e.conv_dw_cin) * 5 + jj; }
                    const float v = acc[a][b][r] * e.alpha;
                    float* dst = Cw + (long)m * g.ldc + nn;
                    if (g.splitk > 1) atomicAdd(dst, v); else if (e.accumulate) *dst += v; else *dst = v;
                }
            }
        return;
    }
    __builtin_amdgcn_s_barrier();                          // every wave is done reading the operand ring
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = a * 16 + lg * 4 + r;
#pragma unroll
            for (int b = 0; b < NB; ++b) stg[row * 64 + ((b * 16 + li) ^ (((row >> 2) & 1) << 4))] = acc[a][b][r];
        }
    // (same-wave LDS hand-off: DS operations of one wave complete in order)
    __builtin_amdgcn_sched_barrier(0);                     // the accumulators die here, before the operand prefetch below claims registers
    const long cbase = zo * g.cs_o + zi * g.cs_i;
    float* Cf = reinterpret_cast<float*>(g.C);
    bf16_t* Ct = reinterpret_cast<bf16_t*>(g.C);
    const bf16_t* aux = reinterpret_cast<const bf16_t*>(e.relu_aux);
    DropCfg dcfg = e.drop;
    if (e.drop.thresh && e.drop_salt) dcfg.key ^= b2s_hash32((uint32_t)(*e.drop_salt) * 2246822519u + 3266489917u);
    const bool vec_ok = (g.ldc & 7) == 0 && (cbase & 7) == 0 && e.conv_dw_cin == 0 && g.splitk == 1 &&
                        ((reinterpret_cast<uintptr_t>(g.C) & 15) == 0);
    const int cchunk = lane & 7;
    const int n = nb + cchunk * 8;
    float bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = (e.bias && n + j < g.N) ? e.bias[n + j] : 0.f;
    // Global operands of the passes below (fp32 residual rows, the bf16 ReLU-mask rows, the C rows of an accumulate store)
    // are fetched four passes at a time, ahead of those passes: C may alias the residual (in-place h += ...), so the
    // compiler cannot lift a pass's loads above the previous pass's stores, and each pass would expose a full memory
    // latency (measured, operands cold in HBM: 37 -> 24 us on the 8148x768x768 residual GEMM; no difference when they
    // are cache-resident).  A lane reads and writes only its own addresses, so the early loads see the same data.
    const bool full = n + 8 <= g.N;
    const bool lane_ok = n < g.N && cchunk < NB * 2;
    const bool pre_res = e.residual && full && (e.ldr & 3) == 0;
    const bool pre_acc = !e.residual && vec_ok && full && g.c_fp32 && e.accumulate;
    const bool pre_aux = aux && full && (e.ld_aux & 7) == 0;
    constexpr int PG = 4;                                  // passes per prefetch group (8 at once does not fit the register budget)
#pragma unroll
    for (int pg = 0; pg < 8; pg += PG) {
    float4 q0[PG], q1[PG];
    uint4 qa[PG];
#pragma unroll
    for (int pp = 0; pp < PG; ++pp) {
        const int m = mb + (pg + pp) * 8 + (lane >> 3);
        q0[pp] = q1[pp] = make_float4(0.f, 0.f, 0.f, 0.f);
        qa[pp] = make_uint4(0, 0, 0, 0);
        if (m >= g.M || !lane_ok) continue;
        if (pre_res) {
            q0[pp] = *reinterpret_cast<const float4*>(e.residual + (long)m * e.ldr + n);
            q1[pp] = *reinterpret_cast<const float4*>(e.residual + (long)m * e.ldr + n + 4);
        } else if (pre_acc) {
            q0[pp] = *reinterpret_cast<const float4*>(Cf + cbase + (long)m * g.ldc + n);
            q1[pp] = *reinterpret_cast<const float4*>(Cf + cbase + (long)m * g.ldc + n + 4);
        }
        if (pre_aux) qa[pp] = *reinterpret_cast<const uint4*>(aux + (long)m * e.ld_aux + n);
    }
#pragma unroll
    for (int pp = 0; pp < PG; ++pp) {
        const int p = pg + pp;
        const int row = p * 8 + (lane >> 3);
        const int m = mb + row;
        const float4 v0 = *reinterpret_cast<const float4*>(stg + row * 64 + ((cchunk * 8) ^ (((row >> 2) & 1) << 4)));
        const float4 v1 = *reinterpret_cast<const float4*>(stg + row * 64 + ((cchunk * 8) ^ (((row >> 2) & 1) << 4)) + 4);
        if (m >= g.M || n >= g.N || cchunk >= NB * 2) continue;
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * e.alpha + bv[j];
        if (e.relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        if (aux) {
            if (pre_aux) {
                const uint4 u = qa[pp];
                const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t bits = (w[j >> 1] >> ((j & 1) * 16)) & 0xffffu;      // bf16 > 0  <=>  sign clear and non-zero
                    v[j] = (bits != 0 && !(bits & 0x8000u)) ? v[j] * e.aux_scale : 0.f;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (n + j < g.N) v[j] = bf2f(aux[(long)m * e.ld_aux + n + j]) > 0.f ? v[j] * e.aux_scale : 0.f;
            }
        }
        if (e.drop.thresh) {
            const uint32_t idx = (uint32_t)(((long)z * g.M + m) * g.N + n);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = b2s_keep(dcfg, idx + j) ? v[j] * dcfg.scale : 0.f;
        }
        if (e.residual) {
            if (pre_res) {
                const float4 r0 = q0[pp], r1 = q1[pp];
                v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) if (n + j < g.N) v[j] += e.residual[(long)m * e.ldr + n + j];
            }
        }
        if (e.row_len) {
            const int bb = m / e.rows_per_batch, t = m - bb * e.rows_per_batch;
            if (t >= e.row_len[bb]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = 0.f;
            }
        }
        const long off = cbase + (long)m * g.ldc + n;
        if (vec_ok && full) {
            if (g.c_fp32) {
                float4* dst = reinterpret_cast<float4*>(Cf + off);
                float4 o0 = make_float4(v[0], v[1], v[2], v[3]), o1 = make_float4(v[4], v[5], v[6], v[7]);
                if (e.accumulate) {
                    const float4 c0 = pre_acc ? q0[pp] : dst[0], c1 = pre_acc ? q1[pp] : dst[1];
                    o0.x += c0.x; o0.y += c0.y; o0.z += c0.z; o0.w += c0.w; o1.x += c1.x; o1.y += c1.y; o1.z += c1.z; o1.w += c1.w;
                }
                epi_store16(dst, make_uint4(__float_as_uint(o0.x), __float_as_uint(o0.y), __float_as_uint(o0.z), __float_as_uint(o0.w)));
                epi_store16(dst + 1, make_uint4(__float_as_uint(o1.x), __float_as_uint(o1.y), __float_as_uint(o1.z), __float_as_uint(o1.w)));
            } else {
                uint4 o;
                o.x = f2bf2(v[0], v[1]); o.y = f2bf2(v[2], v[3]); o.z = f2bf2(v[4], v[5]); o.w = f2bf2(v[6], v[7]);
                epi_store16(Ct + off, o);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (n + j >= g.N) continue;
                int nn = n + j;
                if (e.conv_dw_cin > 0) { int jj = nn / e.conv_dw_cin; nn = (nn - jj * e.conv_dw_cin) * 5 + jj; }
                const long o = cbase + (long)m * g.ldc + nn;
                if (g.c_fp32) { if (g.splitk > 1) atomicAdd(Cf + o, v[j]); else if (e.accumulate) Cf[o] += v[j]; else Cf[o] = v[j]; }
                else Ct[o] = f2bf(v[j]);
            }
        }
    }
    }
}
