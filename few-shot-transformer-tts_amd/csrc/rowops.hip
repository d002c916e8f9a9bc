// Row-wise / element-wise kernels (HBM-bound) for the Transformer-TTS path on gfx950.
// Every kernel reads its rows with 16-byte (fp32x4) or 8/16-byte (bf16) coalesced accesses and reduces
// with 64-lane wavefront shuffles; sequence masks come from the per-utterance lengths (no dense
// bias tensors are built, unlike transformer/common.py:32-48).
#include <algorithm>
#include <cstdlib>
#include "rowops.h"

namespace {

#define RO_DISPATCH(dtype, CALL) do { if (dtype) { typedef bf16_t TY; CALL; } else { typedef float TY; CALL; } } while (0)

__device__ inline float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ inline void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ inline float4 ld4(const bf16_t* p) {
    uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(bf2f(u.x & 0xffff), bf2f(u.x >> 16), bf2f(u.y & 0xffff), bf2f(u.y >> 16));
}
__device__ inline void st4(bf16_t* p, float4 v) {
    uint2 u;
    u.x = f2bf2(v.x, v.y);
    u.y = f2bf2(v.z, v.w);
    *reinterpret_cast<uint2*>(p) = u;
}
__device__ inline float4 drop4(float4 v, const DropCfg& d, uint32_t idx) {
    if (!d.thresh) return v;
    v.x = b2s_keep(d, idx) ? v.x * d.scale : 0.f;
    v.y = b2s_keep(d, idx + 1) ? v.y * d.scale : 0.f;
    v.z = b2s_keep(d, idx + 2) ? v.z * d.scale : 0.f;
    v.w = b2s_keep(d, idx + 3) ? v.w * d.scale : 0.f;
    return v;
}
__device__ inline float block_sum_256(float v, float* sh) {   // sh: >= 4 floats
    v = wave_sum(v);
    int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    return r;
}

// ---------------------------------------------------------------------------------- embedding prep
__global__ void k_embed_prep_fwd(const long* ids, const int* lens, const float* embed, const float* pe,
                                 const float* pe_scale, float* x, int S, int D, DropCfg drop) {
    const int row = blockIdx.x, b = row / S, s = row - b * S;
    const bool valid = s < lens[b];
    const long id = ids[row];
    const float sc = *pe_scale;
    for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4) {
        float4 e = valid ? ld4(embed + id * D + c) : make_float4(0, 0, 0, 0);
        float4 p = ld4(pe + (long)s * D + c);
        float4 v = make_float4(e.x + p.x * sc, e.y + p.y * sc, e.z + p.z * sc, e.w + p.w * sc);
        st4(x + (long)row * D + c, drop4(v, drop, (uint32_t)((long)row * D + c)));
    }
}
template <typename TX>
__global__ void k_embed_prep_bwd(const TX* dx, const long* ids, const int* lens, const float* pe, float* d_embed,
                                 float* d_pe_scale, int S, int D, DropCfg drop) {
    __shared__ float sh[4];
    const int row = blockIdx.x, b = row / S, s = row - b * S;
    const bool valid = s < lens[b];
    const long id = ids[row];
    float acc = 0.f;
    for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4) {
        float4 g = drop4(ld4(dx + (long)row * D + c), drop, (uint32_t)((long)row * D + c));
        float4 p = ld4(pe + (long)s * D + c);
        acc += g.x * p.x + g.y * p.y + g.z * p.z + g.w * p.w;
        if (valid) {
            float* de = d_embed + id * D + c;
            atomicAdd(de, g.x); atomicAdd(de + 1, g.y); atomicAdd(de + 2, g.z); atomicAdd(de + 3, g.w);
        }
    }
    acc = block_sum_256(acc, sh);
    if (threadIdx.x == 0) atomicAdd(d_pe_scale, acc);
}

// ---------------------------------------------------------------------------------- LayerNorm
constexpr int LN_MAXC = 4;   // float4 chunks per lane -> D <= 1024
template <typename T>
__global__ __launch_bounds__(256) void k_ln_fwd(const float* x, const float* gamma, const float* beta, T* y, int ldy,
                                                float* y32, int ldy32, float* mean, float* rstd, int M, int D,
                                                float eps, const int* row_len, int rpb) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nch = D >> 2;
    float4 v[LN_MAXC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        int ci = lane + 64 * i;
        if (ci < nch) { v[i] = ld4(x + (long)row * D + ci * 4); s += v[i].x + v[i].y + v[i].z + v[i].w; }
    }
    const float mu = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        int ci = lane + 64 * i;
        if (ci < nch) {
            float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
            q += a * a + b * b + c * c + d * d;
        }
    }
    const float rs = 1.f / sqrtf(wave_sum(q) / D + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    bool zero = false;
    if (row_len) { int b = row / rpb; zero = (row - b * rpb) >= row_len[b]; }
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        int ci = lane + 64 * i;
        if (ci < nch) {
            float4 g = ld4(gamma + ci * 4), be = ld4(beta + ci * 4), o;
            o.x = (v[i].x - mu) * rs * g.x + be.x; o.y = (v[i].y - mu) * rs * g.y + be.y;
            o.z = (v[i].z - mu) * rs * g.z + be.z; o.w = (v[i].w - mu) * rs * g.w + be.w;
            if (zero) o = make_float4(0, 0, 0, 0);
            if (y) st4(y + (long)row * ldy + ci * 4, o);
            if (y32) st4(y32 + (long)row * ldy32 + ci * 4, o);
        }
    }
}

template <typename TD>
__global__ __launch_bounds__(256) void k_ln_bwd(const TD* dy, int lddy, const float* x, const float* gamma,
                                                const float* mean, const float* rstd, float* dx, int accumulate,
                                                float* dgamma, float* dbeta, int M, int D, const int* row_len,
                                                int rpb, float* ws, bf16_t* dy2, DropCfg drop2) {
    __shared__ float sacc[4 * 2 * LN_MAXC * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = D >> 2;
    const int NW = blockDim.x >> 6;
    for (int i = threadIdx.x; i < 2 * D; i += blockDim.x) sacc[i] = 0.f;
    __syncthreads();
    float4 pg[LN_MAXC], pb[LN_MAXC], gm[LN_MAXC];
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        pg[i] = pb[i] = make_float4(0, 0, 0, 0);
        int ci = lane + 64 * i;
        gm[i] = ci < nch ? ld4(gamma + ci * 4) : make_float4(0, 0, 0, 0);
    }
    for (int row = blockIdx.x * NW + wave; row < M; row += gridDim.x * NW) {
        bool zero = false;
        if (row_len) { int b = row / rpb; zero = (row - b * rpb) >= row_len[b]; }
        const float mu = mean[row], rs = rstd[row];
        float4 g[LN_MAXC], xh[LN_MAXC], prev[LN_MAXC];
        float s1 = 0.f, s2 = 0.f;
        // the accumulate operand (dx itself) is fetched with the row's other loads, not after the row reductions: one
        // memory round trip per row instead of two (each wave walks its rows one at a time)
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            int ci = lane + 64 * i;
            prev[i] = (accumulate && ci < nch) ? ld4(dx + (long)row * D + ci * 4) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            int ci = lane + 64 * i;
            if (ci < nch) {
                float4 d = zero ? make_float4(0, 0, 0, 0) : ld4(dy + (long)row * lddy + ci * 4);
                float4 xv = ld4(x + (long)row * D + ci * 4);
                xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                pb[i].x += d.x; pb[i].y += d.y; pb[i].z += d.z; pb[i].w += d.w;
                pg[i].x += d.x * xh[i].x; pg[i].y += d.y * xh[i].y; pg[i].z += d.z * xh[i].z; pg[i].w += d.w * xh[i].w;
                g[i] = make_float4(d.x * gm[i].x, d.y * gm[i].y, d.z * gm[i].z, d.w * gm[i].w);
                s1 += g[i].x + g[i].y + g[i].z + g[i].w;
                s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
            }
        }
        s1 = wave_sum(s1) / D; s2 = wave_sum(s2) / D;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            int ci = lane + 64 * i;
            if (ci < nch) {
                float4 o;
                o.x = rs * (g[i].x - s1 - xh[i].x * s2); o.y = rs * (g[i].y - s1 - xh[i].y * s2);
                o.z = rs * (g[i].z - s1 - xh[i].z * s2); o.w = rs * (g[i].w - s1 - xh[i].w * s2);
                float* p = dx + (long)row * D + ci * 4;
                o.x += prev[i].x; o.y += prev[i].y; o.z += prev[i].z; o.w += prev[i].w;
                st4(p, o);
                // the next backward op consumes bf16(dropout(dx)): write it here instead of a separate cast pass
                if (dy2) st4(dy2 + (long)row * D + ci * 4, drop4(o, drop2, (uint32_t)((long)row * D + ci * 4)));
            }
        }
    }
    __syncthreads();
    // per-wave partial rows in LDS (plain stores), then a cross-wave sum: LDS float atomics measured ~2x the whole kernel
    float* mine = sacc + wave * 2 * D;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        int ci = lane + 64 * i;
        if (ci < nch) { st4(mine + ci * 4, pg[i]); st4(mine + D + ci * 4, pb[i]); }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * D; i += blockDim.x) {
        float v = 0.f;
        for (int w = 0; w < NW; ++w) v += sacc[w * 2 * D + i];
        if (ws) ws[(long)blockIdx.x * 2 * D + i] = v; else atomicAdd(i < D ? dgamma + i : dbeta + (i - D), v);
    }
}
// ---- compile-time widths (D = 256 * NCH: 768 -> 3, 512 -> 2 -- the decoder / encoder of the default model).
// Every load of a row is UNCONDITIONAL and issued before anything waits: in the generic kernels above each `if (ci < nch)` chunk is a
// branch around its loads, and hipcc waits vmcnt(0) at the join of every such block -- a row became ~8 dependent memory round trips.
// Here a wave issues all loads of its NEXT row before it reduces the current one (two rows in flight per wave).
template <int NCH> struct LnRow { float4 d[NCH], xv[NCH], pv[NCH]; float mu, rs; };
template <typename TD, int NCH, bool ACC, typename TX>
__device__ __forceinline__ void ln_row_load(LnRow<NCH>& r, const TD* __restrict__ dy, int lddy, const float* __restrict__ x, const TX* dx,
                                            const float* __restrict__ mean, const float* __restrict__ rstd, int row, int lane) {
    constexpr int D = NCH * 256;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 4;
        r.d[i] = ld4(dy + (long)row * lddy + c);
        r.xv[i] = ld4(x + (long)row * D + c);
        if (ACC) r.pv[i] = ld4(dx + (long)row * D + c);
    }
    r.mu = mean[row]; r.rs = rstd[row];
}
#ifndef B2S_LN_BWD_WAVES
#define B2S_LN_BWD_WAVES 3      // waves per SIMD the backward kernel is compiled for (4 = 128 VGPRs spills 12 dwords: 8.59 vs 8.50 ms per step)
#endif
// TX: type of the residual gradient dx (fp32, or bf16: 12 instead of 16 bytes per element and launch -- the kernel runs at the rate of
// its read / write mix, tools/mix_lab.hip)
template <typename TD, int NCH, bool ACC, bool DY2, typename TX>
__global__ __launch_bounds__(256, B2S_LN_BWD_WAVES) void k_ln_bwd_fast(const TD* __restrict__ dy, int lddy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd, TX* dx, int M,
                                                     const int* __restrict__ row_len, int rpb, float* __restrict__ ws, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, bf16_t* __restrict__ dy2, DropCfg drop2) {
    constexpr int D = NCH * 256;
    __shared__ float sacc[4 * 2 * D];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // (scalar: row indices, the row_len lookup and the row bases stay off the vector unit)
    float4 pg[NCH], pb[NCH], gm[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) { pg[i] = pb[i] = make_float4(0, 0, 0, 0); gm[i] = ld4(gamma + (lane + 64 * i) * 4); }
    const int nw = gridDim.x * 4;
    int row = blockIdx.x * 4 + wave;
    LnRow<NCH> cur, nxt;
    ln_row_load<TD, NCH, ACC, TX>(cur, dy, lddy, x, dx, mean, rstd, min(row, M - 1), lane);
    for (; row < M; row += nw) {
        ln_row_load<TD, NCH, ACC, TX>(nxt, dy, lddy, x, dx, mean, rstd, min(row + nw, M - 1), lane);      // (past the end: the last row again, unused)
        bool zero = false;
        if (row_len) { const int b = row / rpb; zero = (row - b * rpb) >= row_len[b]; }
        const float mu = cur.mu, rs = cur.rs;
        float4 g[NCH], xh[NCH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const float4 d = zero ? make_float4(0, 0, 0, 0) : cur.d[i];
            const float4 xv = cur.xv[i];
            xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
            pb[i].x += d.x; pb[i].y += d.y; pb[i].z += d.z; pb[i].w += d.w;
            pg[i].x += d.x * xh[i].x; pg[i].y += d.y * xh[i].y; pg[i].z += d.z * xh[i].z; pg[i].w += d.w * xh[i].w;
            g[i] = make_float4(d.x * gm[i].x, d.y * gm[i].y, d.z * gm[i].z, d.w * gm[i].w);
            s1 += g[i].x + g[i].y + g[i].z + g[i].w;
            s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
        }
        s1 = wave_sum(s1) * (1.f / D); s2 = wave_sum(s2) * (1.f / D);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = (lane + 64 * i) * 4;
            float4 o;
            o.x = rs * (g[i].x - s1 - xh[i].x * s2); o.y = rs * (g[i].y - s1 - xh[i].y * s2);
            o.z = rs * (g[i].z - s1 - xh[i].z * s2); o.w = rs * (g[i].w - s1 - xh[i].w * s2);
            if (ACC) { o.x += cur.pv[i].x; o.y += cur.pv[i].y; o.z += cur.pv[i].z; o.w += cur.pv[i].w; }
            st4(dx + (long)row * D + c, o);
            if (DY2) st4(dy2 + (long)row * D + c, drop4(o, drop2, (uint32_t)((long)row * D + c)));
        }
        cur = nxt;
    }
    float* mine = sacc + wave * 2 * D;
#pragma unroll
    for (int i = 0; i < NCH; ++i) { st4(mine + (lane + 64 * i) * 4, pg[i]); st4(mine + D + (lane + 64 * i) * 4, pb[i]); }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * D; i += 256) {
        const float v = sacc[i] + sacc[2 * D + i] + sacc[4 * D + i] + sacc[6 * D + i];
        if (ws) ws[(long)blockIdx.x * 2 * D + i] = v; else atomicAdd(i < D ? dgamma + i : dbeta + (i - D), v);
    }
}
template <typename T, int NCH>
__global__ __launch_bounds__(256) void k_ln_fwd_fast(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     T* __restrict__ y, int ldy, float* __restrict__ y32, int ldy32, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int M, float eps, const int* __restrict__ row_len, int rpb) {
    constexpr int D = NCH * 256;
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= M) return;                                 // (wave-uniform)
    float4 v[NCH], g[NCH], be[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 4;
        v[i] = ld4(x + (long)row * D + c); g[i] = ld4(gamma + c); be[i] = ld4(beta + c);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
    const float mu = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
        q += a * a + b * b + c * c + d * d;
    }
    const float rs = 1.f / sqrtf(wave_sum(q) / D + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    bool zero = false;
    if (row_len) { const int b = row / rpb; zero = (row - b * rpb) >= row_len[b]; }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 4;
        float4 o;
        o.x = (v[i].x - mu) * rs * g[i].x + be[i].x; o.y = (v[i].y - mu) * rs * g[i].y + be[i].y;
        o.z = (v[i].z - mu) * rs * g[i].z + be[i].z; o.w = (v[i].w - mu) * rs * g[i].w + be[i].w;
        if (zero) o = make_float4(0, 0, 0, 0);
        if (y) st4(y + (long)row * ldy + c, o);
        if (y32) st4(y32 + (long)row * ldy32 + c, o);
    }
}

__global__ __launch_bounds__(256) void k_ln_param_reduce(const float* ws, int nblk, int D, float* dgamma, float* dbeta) {
    // column sums of ws [nblk, 2D]: 64 columns per workgroup, rows split over 4 row-lanes and gridDim.y workgroups
    __shared__ float sh[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    float acc = 0.f;
    if (c < 2 * D)
        for (int b = blockIdx.y * 4 + ry; b < nblk; b += gridDim.y * 4) acc += ws[(long)b * 2 * D + c];
    sh[ry][cx] = acc;
    __syncthreads();
    if (ry == 0 && c < 2 * D) {
        const float s = sh[0][cx] + sh[1][cx] + sh[2][cx] + sh[3][cx];
        atomicAdd(c < D ? dgamma + c : dbeta + (c - D), s);
    }
}

// the same reduction for up to RO_LN_BATCH LayerNorms in one launch (blockIdx.z = job): a backward stage's 2-3 LayerNorms
// cost one ~5 us dispatch instead of one each
__global__ __launch_bounds__(256) void k_ln_param_reduce_batch(LnReduceBatch bt) {
    __shared__ float sh[4][64];
    const LnReduceJob jb = bt.j[blockIdx.z];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    float acc = 0.f;
    if (c < 2 * jb.D)
        for (int b = blockIdx.y * 4 + ry; b < jb.nblk; b += gridDim.y * 4) acc += jb.ws[(long)b * 2 * jb.D + c];
    sh[ry][cx] = acc;
    __syncthreads();
    if (ry == 0 && c < 2 * jb.D) {
        const float s = sh[0][cx] + sh[1][cx] + sh[2][cx] + sh[3][cx];
        atomicAdd(c < jb.D ? jb.dgamma + c : jb.dbeta + (c - jb.D), s);
    }
}

// ---------------------------------------------------------------------------------- softmax
template <typename T>
__global__ __launch_bounds__(256) void k_softmax_fwd(const float* S, T* P, T* Pd, int H, int Lq, int Lk, int ldp,
                                                     long rows, float scale, int mask_mode, const int* klen,
                                                     const float* bias, long bias_sb, long bias_sq, DropCfg drop) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int q = (int)(r % Lq);
    const int b = (int)(r / ((long)Lq * H));
    int kend = Lk;
    if (mask_mode & 1) kend = min(kend, klen[b]);
    if (mask_mode & 2) kend = min(kend, q + 1);
    const float* s = S + r * ldp;
    const float* bi = bias ? bias + b * bias_sb + q * bias_sq : nullptr;
    float mx = -INFINITY;
    for (int k = lane; k < kend; k += 64) mx = fmaxf(mx, s[k] * scale + (bi ? bi[k] : 0.f));
    mx = wave_max(mx);
    float sum = 0.f;
    for (int k = lane; k < kend; k += 64) sum += __expf(s[k] * scale + (bi ? bi[k] : 0.f) - mx);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    T* p = P + r * ldp;
    T* pd = Pd ? Pd + r * ldp : nullptr;
    for (int k = lane; k < ldp; k += 64) {
        float v = 0.f;
        if (k < kend) v = __expf(s[k] * scale + (bi ? bi[k] : 0.f) - mx) * inv;
        TT<T>::st(p + k, v);
        if (pd) {
            float w = v;
            if (drop.thresh && k < kend) w = b2s_keep_w(drop, (uint32_t)r, (uint32_t)k) ? v * drop.scale : 0.f;
            TT<T>::st(pd + k, w);
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_softmax_bwd(const T* P, const float* dPraw, T* dS, int Lk, int ldp,
                                                     long rows, float scale, DropCfg drop) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const T* p = P + r * ldp;
    const float* dp = dPraw + r * ldp;
    float acc = 0.f;
    for (int k = lane; k < Lk; k += 64) {
        float d = dp[k];
        if (drop.thresh) d = b2s_keep_w(drop, (uint32_t)r, (uint32_t)k) ? d * drop.scale : 0.f;
        acc += TT<T>::ld(p + k) * d;
    }
    acc = wave_sum(acc);
    T* o = dS + r * ldp;
    for (int k = lane; k < ldp; k += 64) {
        float v = 0.f;
        if (k < Lk) {
            float d = dp[k];
            if (drop.thresh) d = b2s_keep_w(drop, (uint32_t)r, (uint32_t)k) ? d * drop.scale : 0.f;
            v = TT<T>::ld(p + k) * (d - acc) * scale;
        }
        TT<T>::st(o + k, v);
    }
}
template <typename T>
__global__ void k_align_transpose(const T* P, float* align, int Lq, int Lk, int ldp) {
    __shared__ float tile[32][33];
    const int z = blockIdx.z, q0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;    // 256 threads: ty 0..7
    for (int j = ty; j < 32; j += 8) {
        int q = q0 + j, k = k0 + tx;
        tile[j][tx] = (q < Lq && k < Lk) ? TT<T>::ld(P + ((long)z * Lq + q) * ldp + k) : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        int k = k0 + j, q = q0 + tx;
        if (k < Lk && q < Lq) align[((long)z * Lk + k) * Lq + q] = tile[tx][j];
    }
}

// ---------------------------------------------------------------------------------- casts
template <typename T>
__global__ void k_cast_drop(const float* in, int ldin, T* out, int ldo, int M, int N, DropCfg drop) {
    const int nq = N >> 2;
    const long total = (long)M * nq;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int m = (int)(i / nq), c = (int)(i - (long)m * nq) * 4;
        float4 v = drop4(ld4(in + (long)m * ldin + c), drop, (uint32_t)((long)m * N + c));
        st4(out + (long)m * ldo + c, v);
    }
}
template <typename T>
__global__ void k_cast(const float* in, T* out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        TT<T>::st(out + i, in[i]);
}
template <typename T>
__global__ void k_cast_back(const T* in, float* out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = TT<T>::ld(in + i);
}
// bf16 forms on 16-byte-aligned buffers, 8 elements per thread and iteration (the gradient exchange's pack / unpack of 32 MB buckets, the
// step's mel casts): 2 x 16-byte loads + one 16-byte store instead of 4-byte loads + 2-byte stores; the last n % 8 elements by thread 0..7
__global__ __launch_bounds__(256) void k_cast_bf16_v(const float* __restrict__ in, bf16_t* __restrict__ out, long n) {
    const long n8 = n >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const float4 a = ld4(in + i * 8), b = ld4(in + i * 8 + 4);
        uint4 o;
        o.x = (uint32_t)f2bf(a.x) | ((uint32_t)f2bf(a.y) << 16); o.y = (uint32_t)f2bf(a.z) | ((uint32_t)f2bf(a.w) << 16);
        o.z = (uint32_t)f2bf(b.x) | ((uint32_t)f2bf(b.y) << 16); o.w = (uint32_t)f2bf(b.z) | ((uint32_t)f2bf(b.w) << 16);
        *reinterpret_cast<uint4*>(out + i * 8) = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) out[(n8 << 3) + threadIdx.x] = f2bf(in[(n8 << 3) + threadIdx.x]);
}
__global__ __launch_bounds__(256) void k_cast_back_bf16_v(const bf16_t* __restrict__ in, float* __restrict__ out, long n) {
    const long n8 = n >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const uint4 v = *reinterpret_cast<const uint4*>(in + i * 8);
        st4(out + i * 8, make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)));
        st4(out + i * 8 + 4, make_float4(__uint_as_float(v.z << 16), __uint_as_float(v.z & 0xffff0000u), __uint_as_float(v.w << 16), __uint_as_float(v.w & 0xffff0000u)));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) out[(n8 << 3) + threadIdx.x] = TT<bf16_t>::ld(in + (n8 << 3) + threadIdx.x);
}

// ---------------------------------------------------------------------------------- ragged (compact) decoder rows
// The decoder segment of a training step may keep its token rows RAGGED: utterance b owns rows [off[b], off[b + 1]) = frames t < target_lengths[b]
// (engine.hip: b2s_decoder_compact_rows), so that every row-wise kernel of the segment runs over sum(target_lengths) rows instead of B x T.
// These kernels sit at the segment's boundaries: padded [B, T, C] fp32 tensors in, padded out (zeros on padded rows).
__device__ inline int ragged_batch(const int* __restrict__ off, int B, int row) { int b = 0; while (b + 1 < B && off[b + 1] <= row) ++b; return b; }
struct RowOff64 { int off[65]; };
__global__ void k_set_rowoff(RowOff64 h, int* dst, int n) { if ((int)threadIdx.x < n) dst[threadIdx.x] = h.off[threadIdx.x]; }
// (in2 / out2, optional: a second fp32 tensor [B, T, C2] gathered by the same launch -- the decoder backward's d_stop beside d_mels)
template <typename TO>
__global__ void k_rows_gather(const float* __restrict__ in, TO* __restrict__ out, const int* __restrict__ off, int T, int C,
                              const float* __restrict__ in2, float* __restrict__ out2, int C2) {
    const int b = blockIdx.y, t = blockIdx.x, r0 = off[b];
    if (t >= off[b + 1] - r0) return;
    for (int c = threadIdx.x; c < C; c += blockDim.x) TT<TO>::st(out + (long)(r0 + t) * C + c, in[((long)b * T + t) * C + c]);
    if (in2) for (int c = threadIdx.x; c < C2; c += blockDim.x) out2[(long)(r0 + t) * C2 + c] = in2[((long)b * T + t) * C2 + c];
}
// decoder heads on ragged rows, one launch: the stop projection of a frame (k_rowdot's arithmetic, term for term) + the scatter of its mel row and its
// stop logit into the caller's padded [B, T, *] tensors (zeros where the frame does not exist) -- instead of k_rowdot + two k_rows_scatter
template <typename TX>
__global__ __launch_bounds__(64) void k_heads_scatter(const TX* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ bias,
                                                      const float* __restrict__ mel_in, float* __restrict__ mel_out, float* __restrict__ stop_out,
                                                      const int* __restrict__ off, int T, int C, int D) {
    const int b = blockIdx.y, t = blockIdx.x, r0 = off[b], lane = threadIdx.x;
    const bool live = t < off[b + 1] - r0;
    const long row = r0 + t, prow = (long)b * T + t;
    for (int c = lane; c < C; c += 64) mel_out[prow * C + c] = live ? mel_in[row * C + c] : 0.f;
    float acc = 0.f;
    if (live) {
        for (int c = lane * 4; c < D; c += 256) {
            float4 v = ld4(x + row * ldx + c), ww = ld4(w + c);
            acc += v.x * ww.x + v.y * ww.y + v.z * ww.z + v.w * ww.w;
        }
        acc = wave_sum(acc);
    }
    if (lane == 0) stop_out[prow] = live ? acc + bias[0] : 0.f;
}

// ---------------------------------------------------------------------------------- decoder input prep
// (off != nullptr: ragged rows -- `row` is the ragged row index, also the row of the dropout index)
__global__ void k_shift_pe_fwd(const float* a, const int* lens, const float* pe, const float* pe_scale, float* x,
                               int T, int D, DropCfg drop, const int* off, int B) {
    const int row = blockIdx.x, b = off ? ragged_batch(off, B, row) : row / T, t = off ? row - off[b] : row - b * T;
    const bool have = t > 0 && (t - 1) < lens[b];
    const float sc = *pe_scale;
    for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4) {
        float4 e = have ? ld4(a + (long)(row - 1) * D + c) : make_float4(0, 0, 0, 0);
        float4 p = ld4(pe + (long)t * D + c);
        float4 v = make_float4(e.x + p.x * sc, e.y + p.y * sc, e.z + p.z * sc, e.w + p.w * sc);
        st4(x + (long)row * D + c, drop4(v, drop, (uint32_t)((long)row * D + c)));
    }
}
template <typename T_, typename TX>
__global__ __launch_bounds__(512) void k_shift_pe_bwd(const TX* dx, const int* lens, const float* pe, T_* da, float* d_pe_scale, int T,
                                                      int D, DropCfg drop, int rows, const int* off, int B) {
    // A wave takes two rows per pass (every load of both issued before the first is used).  At most 256 workgroups: the kernel ends
    // with one float atomic per workgroup on d_pe_scale, and same-address atomics retire at ~7 ns each (2048 workgroups: 35 us of
    // which 15 were the atomics).
    __shared__ float sh[8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, stride = gridDim.x * 8;
    float acc = 0.f;
    for (int r0 = blockIdx.x * 8 + wave; r0 < rows; r0 += 2 * stride) {
        int row[2]; bool live[2], have[2]; long nxt[2]; int tt[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            live[q] = r0 + q * stride < rows;
            row[q] = live[q] ? r0 + q * stride : r0;
            const int b = off ? ragged_batch(off, B, row[q]) : row[q] / T;
            tt[q] = off ? row[q] - off[b] : row[q] - b * T;
            // da[b,t] = g[b,t+1] if t+1 < T and t < len[b]  (ragged rows: row t + 1 = len[b] does not exist -- its gradient is zero in the padded layout)
            have[q] = (tt[q] + 1) < (off ? lens[b] : T) && tt[q] < lens[b];
            nxt[q] = have[q] ? row[q] + 1 : row[q];                   // (the clamped row is loaded and discarded)
        }
#pragma unroll 3
        for (int c = lane * 4; c < D; c += 256) {
            float4 g[2], p[2], o[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                g[q] = ld4(dx + (long)row[q] * D + c);
                p[q] = ld4(pe + (long)tt[q] * D + c);
                o[q] = ld4(dx + nxt[q] * D + c);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (!live[q]) continue;
                const float4 gd = drop4(g[q], drop, (uint32_t)((long)row[q] * D + c));
                acc += gd.x * p[q].x + gd.y * p[q].y + gd.z * p[q].z + gd.w * p[q].w;
                float4 od = drop4(o[q], drop, (uint32_t)(nxt[q] * D + c));
                if (!have[q]) od = make_float4(0, 0, 0, 0);
                st4(da + (long)row[q] * D + c, od);
            }
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) sh[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(d_pe_scale, ((sh[0] + sh[1]) + (sh[2] + sh[3])) + ((sh[4] + sh[5]) + (sh[6] + sh[7])));
}

// ---------------------------------------------------------------------------------- speaker / language nets
template <typename T>
__global__ void k_embed_net_fwd(const long* ids, const float* table, const float* vecs, int L, const float* Wl,
                                const float* W, const float* bias, float* e_raw, float* h_pre, float* mem32, T* memT,
                                int ldm, int col0, int S, int E) {
    extern __shared__ float sh[];        // E floats
    const int b = blockIdx.x, n = threadIdx.x;
    if (n < E) {
        float e;
        if (ids) e = table[ids[b] * E + n];
        else { e = 0.f; for (int l = 0; l < L; ++l) e += Wl[n * L + l] * vecs[b * L + l]; }
        sh[n] = e; e_raw[b * E + n] = e;
    }
    __syncthreads();
    if (n < E) {
        float h = bias[n];
        for (int k = 0; k < E; ++k) h += W[n * E + k] * sh[k];
        h_pre[b * E + n] = h;
        float o = h / (1.f + fabsf(h));
        for (int s = 0; s < S; ++s) {
            long off = ((long)b * S + s) * ldm + col0 + n;
            if (mem32) mem32[off] = o;
            if (memT) TT<T>::st(memT + off, o);
        }
    }
}
// phase 1 (one workgroup per utterance): dh[b,n] = softsign'(h) * sum_s dmem[b,s,col0+n]
__global__ __launch_bounds__(1024) void k_embed_net_bwd_dh(const float* dmem, int ldm, int col0, const float* h_pre, float* dh, int S, int E) {
    // threads = E columns x G row groups (the serial walk over S rows was latency-bound); partial sums meet in LDS
    __shared__ float part[1024];
    const int b = blockIdx.x, n = threadIdx.x % E, grp = threadIdx.x / E, G = blockDim.x / E;
    float d = 0.f;
    for (int s = grp; s < S; s += G) d += dmem[((long)b * S + s) * ldm + col0 + n];
    part[threadIdx.x] = d;
    __syncthreads();
    if (grp == 0) {
        for (int g = 1; g < G; ++g) d += part[g * E + n];
        const float h = h_pre[b * E + n], den = 1.f + fabsf(h);
        dh[b * E + n] = d / (den * den);
    }
}
// phase 2 (one workgroup per utterance): de[b,n] = sum_k W[k][n] dh[b,k]; embedding-table rows get it by atomics
__global__ void k_embed_net_bwd_de(const float* dh, const long* ids, const float* W, float* de, float* d_table, int E) {
    extern __shared__ float sh[];
    const int b = blockIdx.x, n = threadIdx.x;
    if (n < E) sh[n] = dh[b * E + n];
    __syncthreads();
    if (n >= E) return;
    float v = 0.f;
    for (int k = 0; k < E; ++k) v += W[k * E + n] * sh[k];
    de[b * E + n] = v;
    if (ids) atomicAdd(d_table + ids[b] * E + n, v);                       // rows of different utterances may coincide
}
// phase 3 (one workgroup per output row n): dW[n,:] += sum_b dh[b,n] e[b,:], db[n], dWl[n,:] += sum_b de[b,n] vec[b,:]
__global__ void k_embed_net_bwd_w(const float* dh, const float* de, const float* e_raw, const float* vecs, int L, float* dW,
                                  float* db, float* dWl, int B, int E) {
    const int n = blockIdx.x, k = threadIdx.x;
    if (k < E) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += dh[b * E + n] * e_raw[b * E + k];
        dW[n * E + k] += acc;
    }
    if (k == 0) { float s = 0.f; for (int b = 0; b < B; ++b) s += dh[b * E + n]; db[n] += s; }
    if (dWl && k < L) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += de[b * E + n] * vecs[b * L + k];
        dWl[n * L + k] += acc;
    }
}

// ---------------------------------------------------------------------------------- rowdot / colsum
template <typename T>
__global__ __launch_bounds__(256) void k_rowdot(const T* x, int ldx, const float* w, const float* b, float* out, int M,
                                                int D, const int* row_len, int rpb) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float acc = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        float4 v = ld4(x + (long)row * ldx + c), ww = ld4(w + c);
        acc += v.x * ww.x + v.y * ww.y + v.z * ww.z + v.w * ww.w;
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        bool zero = false;
        if (row_len) { int bb = row / rpb; zero = (row - bb * rpb) >= row_len[bb]; }
        out[row] = zero ? 0.f : acc + b[0];
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_colsum(const T* X, int ldx, const float* wgt, float* out, int M, int C) {
    __shared__ float sh[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    float acc = 0.f;
    if (c < C)
        for (int m = blockIdx.y * 4 + ry; m < M; m += gridDim.y * 4)
            acc += (wgt ? wgt[m] : 1.f) * TT<T>::ld(X + (long)m * ldx + c);
    sh[ry][cx] = acc;
    __syncthreads();
    if (ry == 0 && c < C) atomicAdd(out + c, sh[0][cx] + sh[1][cx] + sh[2][cx] + sh[3][cx]);
}

// ---------------------------------------------------------------------------------- BatchNorm
__global__ __launch_bounds__(256) void k_bn_colred(const float* y, int M, int C, const float* mu_sum, float* out) {
    // mu_sum == nullptr: out[c] += sum_m y ; else out[c] += sum_m (y - mu_sum[c]/M)^2
    __shared__ float sh[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    float acc = 0.f;
    if (c < C) {
        const float mu = mu_sum ? mu_sum[c] / M : 0.f;
        for (int m = blockIdx.y * 4 + ry; m < M; m += gridDim.y * 4) {
            float v = y[(long)m * C + c];
            if (mu_sum) { v -= mu; v *= v; }
            acc += v;
        }
    }
    sh[ry][cx] = acc;
    __syncthreads();
    if (ry == 0 && c < C) atomicAdd(out + c, sh[0][cx] + sh[1][cx] + sh[2][cx] + sh[3][cx]);
}
__global__ void k_bn_finalize(const float* scratch, int M, int C, float* mean, float* rstd, float eps, float* rm,
                              float* rv, long* nbt, float mom) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        float mu = scratch[c] / M, var = scratch[C + c] / M;
        mean[c] = mu; rstd[c] = 1.f / sqrtf(var + eps);
        if (rm) {
            rm[c] = (1.f - mom) * rm[c] + mom * mu;
            rv[c] = (1.f - mom) * rv[c] + mom * var * ((float)M / (float)(M - 1));
        }
    }
    if (c == 0 && nbt) *nbt += 1;
}
__global__ void k_bn_eval_stats(const float* rm, const float* rv, float* mean, float* rstd, float eps, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) { mean[c] = rm[c]; rstd[c] = 1.f / sqrtf(rv[c] + eps); }
}
template <typename T>
__global__ void k_bn_apply(const float* y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                           int use_tanh, T* outT, float* out32, const float* add32, int M, int C, DropCfg drop) {
    const int nq = C >> 2;
    const long total = (long)M * nq;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int m = (int)(i / nq), c = (int)(i - (long)m * nq) * 4;
        float4 v = ld4(y + (long)m * C + c), mu = ld4(mean + c), rs = ld4(rstd + c), g = ld4(gamma + c), be = ld4(beta + c);
        float4 o;
        o.x = (v.x - mu.x) * rs.x * g.x + be.x; o.y = (v.y - mu.y) * rs.y * g.y + be.y;
        o.z = (v.z - mu.z) * rs.z * g.z + be.z; o.w = (v.w - mu.w) * rs.w * g.w + be.w;
        if (use_tanh) { o.x = tanhf(o.x); o.y = tanhf(o.y); o.z = tanhf(o.z); o.w = tanhf(o.w); }
        o = drop4(o, drop, (uint32_t)((long)m * C + c));
        if (outT) st4(outT + (long)m * C + c, o);
        if (out32) {
            if (add32) { float4 a = ld4(add32 + (long)m * C + c); o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w; }
            st4(out32 + (long)m * C + c, o);
        }
    }
}
template <typename TD>
__device__ inline float bn_dz(const TD* dout, const float* y, long off, int c, const float* mean, const float* rstd,
                              const float* gamma, const float* beta, int use_tanh, const DropCfg& drop, float* xhat) {
    float d = TT<TD>::ld(dout + off);
    if (drop.thresh) d = b2s_keep(drop, (uint32_t)off) ? d * drop.scale : 0.f;
    float xh = (y[off] - mean[c]) * rstd[c];
    *xhat = xh;
    if (use_tanh) { float a = tanhf(gamma[c] * xh + beta[c]); d *= (1.f - a * a); }
    return d;
}
template <typename TD>
__global__ __launch_bounds__(256) void k_bn_bwd_red(const TD* dout, const float* y, const float* mean,
                                                    const float* rstd, const float* gamma, const float* beta,
                                                    int use_tanh, float* dgamma, float* dbeta, int M, int C,
                                                    DropCfg drop) {
    __shared__ float sh[2][4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    float a1 = 0.f, a2 = 0.f;
    if (c < C)
        for (int m = blockIdx.y * 4 + ry; m < M; m += gridDim.y * 4) {
            float xh;
            float dz = bn_dz<TD>(dout, y, (long)m * C + c, c, mean, rstd, gamma, beta, use_tanh, drop, &xh);
            a1 += dz; a2 += dz * xh;
        }
    sh[0][ry][cx] = a1; sh[1][ry][cx] = a2;
    __syncthreads();
    if (ry == 0 && c < C) {
        atomicAdd(dbeta + c, sh[0][0][cx] + sh[0][1][cx] + sh[0][2][cx] + sh[0][3][cx]);
        atomicAdd(dgamma + c, sh[1][0][cx] + sh[1][1][cx] + sh[1][2][cx] + sh[1][3][cx]);
    }
}
template <typename TD, typename T>
__global__ void k_bn_bwd_apply(const TD* dout, const float* y, const float* mean, const float* rstd,
                               const float* gamma, const float* beta, int use_tanh, const float* dgamma,
                               const float* dbeta, T* dy, int M, int C, DropCfg drop) {
    const long total = (long)M * C;
    const float invM = 1.f / M;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        float xh;
        float dz = bn_dz<TD>(dout, y, i, c, mean, rstd, gamma, beta, use_tanh, drop, &xh);
        TT<T>::st(dy + i, gamma[c] * rstd[c] * (dz - dbeta[c] * invM - xh * dgamma[c] * invM));
    }
}

// ---- vectorised BatchNorm kernels (C % 4 == 0): a thread owns FOUR adjacent channels and RU rows whose loads are all issued before
// the first use (one memory round trip per thread), 4 row lanes per workgroup.  The scalar kernels above read 4 bytes per lane and
// walked their rows one dependent round trip at a time (1-2 TB/s).
constexpr int BN_RU = 8;
__device__ __forceinline__ float fast_tanh(float x) { return 1.f - 2.f / (__expf(2.f * x) + 1.f); }      // (|err| ~ 1e-7: exp overflow -> 1, underflow -> -1)
struct BnStat {                   // how the forward kernel gets mean / rstd
    const float* sums;            // train: [2C] column sums of y and y^2 (written by the conv GEMM's epilogue); null: mean / rstd given
    float *mean, *rstd;           // train: outputs (saved for the backward pass); eval: inputs
    float *rm, *rv; long* nbt;    // running statistics to update (train; may be null)
    float eps, mom;
};
// out = dropout(tanh?(gamma * (y - mean) * rstd + beta)) [+ add32]; in training mode mean / rstd come from the column sums and the
// workgroups of grid row 0 also store them and update the running statistics (k_bn_finalize's job, without its launch)
template <typename T>
__global__ __launch_bounds__(256) void k_bn_apply_v(const float* __restrict__ y, BnStat bs, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    int use_tanh, T* __restrict__ outT, float* __restrict__ out32, const float* __restrict__ add32, int M,
                                                    int C, DropCfg drop) {
    const int cq = blockIdx.x * 64 + (threadIdx.x & 63), ry = threadIdx.x >> 6;
    const int c = min(cq * 4, C - 4);                    // (threads past the last channel group re-do it: no divergent loads; they do not store)
    const bool own = cq * 4 < C;
    float4 mu, rs;
    if (bs.sums) {
        const float4 s1 = ld4(bs.sums + c), s2 = ld4(bs.sums + C + c);
        const float im = 1.f / M;
        mu = make_float4(s1.x * im, s1.y * im, s1.z * im, s1.w * im);
        const float4 var = make_float4(fmaxf(s2.x * im - mu.x * mu.x, 0.f), fmaxf(s2.y * im - mu.y * mu.y, 0.f), fmaxf(s2.z * im - mu.z * mu.z, 0.f),
                                       fmaxf(s2.w * im - mu.w * mu.w, 0.f));
        rs = make_float4(1.f / sqrtf(var.x + bs.eps), 1.f / sqrtf(var.y + bs.eps), 1.f / sqrtf(var.z + bs.eps), 1.f / sqrtf(var.w + bs.eps));
        if (blockIdx.y == 0 && ry == 0 && own) {
            st4(bs.mean + c, mu); st4(bs.rstd + c, rs);
            if (bs.rm) {
                const float unb = (float)M / (float)(M - 1), a = 1.f - bs.mom, b = bs.mom;
                const float4 m0 = ld4(bs.rm + c), v0 = ld4(bs.rv + c);
                st4(bs.rm + c, make_float4(a * m0.x + b * mu.x, a * m0.y + b * mu.y, a * m0.z + b * mu.z, a * m0.w + b * mu.w));
                st4(bs.rv + c, make_float4(a * v0.x + b * var.x * unb, a * v0.y + b * var.y * unb, a * v0.z + b * var.z * unb, a * v0.w + b * var.w * unb));
            }
            if (bs.nbt && cq == 0) *bs.nbt += 1;
        }
    } else { mu = ld4(bs.mean + c); rs = ld4(bs.rstd + c); }
    const float4 g = ld4(gamma + c), be = ld4(beta + c);
    const int m0 = (blockIdx.y * 4 + ry) * BN_RU;
    float4 v[BN_RU], ad[BN_RU];
#pragma unroll
    for (int j = 0; j < BN_RU; ++j) {
        const long o = (long)min(m0 + j, M - 1) * C + c;
        v[j] = ld4(y + o);
        if (add32) ad[j] = ld4(add32 + o);
    }
#pragma unroll
    for (int j = 0; j < BN_RU; ++j) {
        const int m = m0 + j;
        float4 o;
        o.x = (v[j].x - mu.x) * rs.x * g.x + be.x; o.y = (v[j].y - mu.y) * rs.y * g.y + be.y;
        o.z = (v[j].z - mu.z) * rs.z * g.z + be.z; o.w = (v[j].w - mu.w) * rs.w * g.w + be.w;
        if (use_tanh) { o.x = fast_tanh(o.x); o.y = fast_tanh(o.y); o.z = fast_tanh(o.z); o.w = fast_tanh(o.w); }
        o = drop4(o, drop, (uint32_t)((long)m * C + c));
        if (m < M && own) {
            if (outT) st4(outT + (long)m * C + c, o);
            if (out32) {
                if (add32) { o.x += ad[j].x; o.y += ad[j].y; o.z += ad[j].z; o.w += ad[j].w; }
                st4(out32 + (long)m * C + c, o);
            }
        }
    }
}
// dz = dropout-mask(dout) * tanh'(gamma xhat + beta) for 4 channels of one row
template <typename TD>
__device__ __forceinline__ float4 bn_dz4(float4 d, float4 yv, float4 mu, float4 rs, float4 g, float4 be, int use_tanh, const DropCfg& drop, uint32_t off,
                                         float4* xh) {
    d = drop4(d, drop, off);
    *xh = make_float4((yv.x - mu.x) * rs.x, (yv.y - mu.y) * rs.y, (yv.z - mu.z) * rs.z, (yv.w - mu.w) * rs.w);
    if (use_tanh) {
        const float a0 = fast_tanh(g.x * xh->x + be.x), a1 = fast_tanh(g.y * xh->y + be.y), a2 = fast_tanh(g.z * xh->z + be.z), a3 = fast_tanh(g.w * xh->w + be.w);
        d.x *= 1.f - a0 * a0; d.y *= 1.f - a1 * a1; d.z *= 1.f - a2 * a2; d.w *= 1.f - a3 * a3;
    }
    return d;
}
// APPLY = false: dgamma[c] += sum_m dz xhat, dbeta[c] += sum_m dz.   APPLY = true: dy = gamma rstd (dz - dbeta / M - xhat dgamma / M)
template <typename TD, typename T, bool APPLY>
__global__ __launch_bounds__(256) void k_bn_bwd_v(const TD* __restrict__ dout, const float* __restrict__ y, const float* __restrict__ mean,
                                                  const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  int use_tanh, float* dgamma, float* dbeta, T* __restrict__ dy, int M, int C, DropCfg drop) {
    __shared__ float sh[2][4][256];
    const int cx = threadIdx.x & 63, cq = blockIdx.x * 64 + cx, ry = threadIdx.x >> 6;
    const int c = min(cq * 4, C - 4);
    const bool own = cq * 4 < C;
    const float4 mu = ld4(mean + c), rs = ld4(rstd + c), g = ld4(gamma + c), be = ld4(beta + c);
    float4 sg = make_float4(0, 0, 0, 0), sb = make_float4(0, 0, 0, 0);
    if (APPLY) { sg = ld4(dgamma + c); sb = ld4(dbeta + c); }
    const int m0 = (blockIdx.y * 4 + ry) * BN_RU;
    float4 dv[BN_RU], yv[BN_RU];
#pragma unroll
    for (int j = 0; j < BN_RU; ++j) {
        const long o = (long)min(m0 + j, M - 1) * C + c;
        dv[j] = ld4(dout + o); yv[j] = ld4(y + o);
    }
    const float invM = 1.f / M;
    float4 a1 = make_float4(0, 0, 0, 0), a2 = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < BN_RU; ++j) {
        const int m = m0 + j;
        float4 xh;
        float4 dz = bn_dz4<TD>(dv[j], yv[j], mu, rs, g, be, use_tanh, drop, (uint32_t)((long)min(m, M - 1) * C + c), &xh);
        if (APPLY) {
            float4 o;
            o.x = g.x * rs.x * (dz.x - sb.x * invM - xh.x * sg.x * invM); o.y = g.y * rs.y * (dz.y - sb.y * invM - xh.y * sg.y * invM);
            o.z = g.z * rs.z * (dz.z - sb.z * invM - xh.z * sg.z * invM); o.w = g.w * rs.w * (dz.w - sb.w * invM - xh.w * sg.w * invM);
            if (m < M && own) st4(dy + (long)m * C + c, o);
        } else if (m < M) {
            a1.x += dz.x; a1.y += dz.y; a1.z += dz.z; a1.w += dz.w;
            a2.x += dz.x * xh.x; a2.y += dz.y * xh.y; a2.z += dz.z * xh.z; a2.w += dz.w * xh.w;
        }
    }
    if (!APPLY) {
        st4(&sh[0][ry][cx * 4], a1); st4(&sh[1][ry][cx * 4], a2);
        __syncthreads();
        const int cc = blockIdx.x * 256 + threadIdx.x;        // one thread per channel of the workgroup's 256
        if (cc < C) {
            const int i = threadIdx.x;
            atomicAdd(dbeta + cc, sh[0][0][i] + sh[0][1][i] + sh[0][2][i] + sh[0][3][i]);
            atomicAdd(dgamma + cc, sh[1][0][i] + sh[1][1][i] + sh[1][2][i] + sh[1][3][i]);
        }
    }
}

// ---------------------------------------------------------------------------------- loss
__device__ inline float softplusf(float x) { return fmaxf(x, 0.f) + log1pf(__expf(-fabsf(x))); }
__global__ __launch_bounds__(256) void k_loss_partial(const float* bef, const float* aft, const float* stop,
                                                      const float* tgt, const int* lens, float* scratch, int B, int T,
                                                      int C, float pw) {
    // grid (x, B): one wave per frame of utterance blockIdx.y, strided over x; block-level sums, then 4 atomics per
    // workgroup (bef, aft, stop, per-sample aft) -- same-address atomics serialise, so keep them to a few hundred
    __shared__ float sh[4];
    const int lane = threadIdx.x & 63, b = blockIdx.y;
    const int len = min(lens[b], T);
    float tb = 0.f, ta = 0.f, tc = 0.f;
    for (int t = blockIdx.x * 4 + (threadIdx.x >> 6); t < len; t += gridDim.x * 4) {
        const long row = (long)b * T + t;
        float sb = 0.f, sa = 0.f;
        for (int c = lane; c < C; c += 64) {
            float y = tgt[row * C + c];
            float d1 = bef[row * C + c] - y, d2 = aft[row * C + c] - y;
            sb += d1 * d1; sa += d2 * d2;
        }
        tb += sb; ta += sa;                       // lanes keep their partial sums; reduced once at the end
        if (lane == 0) {
            float x = stop[row];
            tc += (t == len - 1) ? pw * softplusf(-x) : softplusf(x);
        }
    }
    tb = block_sum_256(tb, sh) / C;
    ta = block_sum_256(ta, sh) / C;
    tc = block_sum_256(tc, sh);
    if (threadIdx.x == 0) {
        atomicAdd(scratch + 0, tb); atomicAdd(scratch + 1, ta); atomicAdd(scratch + 2, tc);
        atomicAdd(scratch + 3 + b, ta);
    }
}
// The same sums, vectorised: the valid frames of an utterance are one contiguous prefix of len * C floats of its [T, C] block, so the squared
// errors stream as float4 with four independent positions per thread in flight (the kernel above walks rows one dependent round trip at a time)
__global__ __launch_bounds__(256) void k_loss_partial_v(const float* __restrict__ bef, const float* __restrict__ aft, const float* __restrict__ stop,
                                                        const float* __restrict__ tgt, const int* __restrict__ lens, float* scratch, int T, int C, float pw) {
    __shared__ float sh[4];
    const int b = blockIdx.y, len = min(lens[b], T);
    const long base = (long)b * T * C;
    const int n4 = len * C / 4, stride = gridDim.x * 256;
    float tb = 0.f, ta = 0.f, tc = 0.f;
    for (int i0 = blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += 4 * stride) {
        float4 y[4], p[4], q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long o = base + 4L * min(i0 + k * stride, n4 - 1);
            y[k] = ld4(tgt + o); p[k] = ld4(bef + o); q[k] = ld4(aft + o);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k * stride >= n4) continue;
            const float a0 = p[k].x - y[k].x, a1 = p[k].y - y[k].y, a2 = p[k].z - y[k].z, a3 = p[k].w - y[k].w;
            const float c0 = q[k].x - y[k].x, c1 = q[k].y - y[k].y, c2 = q[k].z - y[k].z, c3 = q[k].w - y[k].w;
            tb += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3; ta += c0 * c0 + c1 * c1 + c2 * c2 + c3 * c3;
        }
    }
    if (blockIdx.x == 0)
        for (int t = threadIdx.x; t < len; t += 256) {
            const float x = stop[(long)b * T + t];
            tc += (t == len - 1) ? pw * softplusf(-x) : softplusf(x);
        }
    tb = block_sum_256(tb, sh) / C;
    ta = block_sum_256(ta, sh) / C;
    tc = block_sum_256(tc, sh);
    if (threadIdx.x == 0) {
        atomicAdd(scratch + 0, tb); atomicAdd(scratch + 1, ta); if (blockIdx.x == 0) atomicAdd(scratch + 2, tc);
        atomicAdd(scratch + 3 + b, ta);
    }
}
__global__ void k_loss_finalize(const float* scratch, const int* lens, const float* l2, float* out, float* aft_losses,
                                int B) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float sl = 0.f;
        for (int b = 0; b < B; ++b) sl += (float)lens[b];
        float bef = scratch[0] / sl, aft = scratch[1] / sl, ce = scratch[2] / sl, l = l2 ? *l2 : 0.f;
        out[0] = bef + aft + l + ce; out[1] = bef; out[2] = aft; out[3] = (bef + aft) * 0.5f; out[4] = l; out[5] = ce;
        out[6] = sl;
        for (int b = 0; b < B; ++b) aft_losses[b] = scratch[3 + b] / (float)lens[b];
    }
}
__global__ __launch_bounds__(256) void k_loss_bwd(const float* bef, const float* aft, const float* stop,
                                                  const float* tgt, const int* lens, const float* gscale, float* d_bef,
                                                  float* d_aft, float* d_stop, int B, int T, int C, float pw) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * T) return;
    const int b = row / T, t = row - b * T, len = lens[b];
    float sl = 0.f;
    for (int i = 0; i < B; ++i) sl += (float)lens[i];
    const float wb = (gscale ? gscale[0] : 1.f) / sl, wa = (gscale ? gscale[1] : 1.f) / sl, ws = (gscale ? gscale[2] : 1.f) / sl;
    const bool valid = t < len;
    const float k = valid ? 2.f / C : 0.f;
    for (int c = lane; c < C; c += 64) {
        long o = (long)row * C + c;
        float y = tgt[o];
        d_bef[o] = k * wb * (bef[o] - y);
        d_aft[o] = k * wa * (aft[o] - y);
    }
    if (lane == 0) {
        float g = 0.f;
        if (valid) {
            float x = stop[row], sg = 1.f / (1.f + __expf(-x));
            g = (t == len - 1) ? -pw * (1.f - sg) : sg;
            g *= ws;
        }
        d_stop[row] = g;
    }
}

// ---------------------------------------------------------------------------------- multi-tensor
__global__ __launch_bounds__(256) void k_mt_sumsq(const MtChunk* ch, float* out, float scale) {
    __shared__ float sh[4];
    const MtChunk c = ch[blockIdx.x];
    float acc = 0.f;
    for (int i = threadIdx.x; i < c.n; i += 256) { float v = c.a[i]; acc += v * v; }
    acc = block_sum_256(acc, sh);
    if (threadIdx.x == 0) atomicAdd(out, acc * scale);
}
__global__ __launch_bounds__(256) void k_mt_axpy(const MtChunk* ch, float alpha, const float* gscale) {
    const MtChunk c = ch[blockIdx.x];
    const float a = alpha * (gscale ? *gscale : 1.f);
    for (int i = threadIdx.x; i < c.n; i += 256) c.b[i] += a * c.a[i];
}
__device__ __forceinline__ float adam_elem(float& p, float g_raw, float& m, float& v, float gs, float l2p, float b1, float b2, float step, float sbc2,
                                           float eps) {
    // no fused multiply-add contraction: the update of an element must not depend on which loop of which chunking reaches it (the unrolled
    // body, the tail loop, a clipped chunk of the sharded optimizer) -- the two virtual ranks of tests/test_gpu_rs_ag.py reproduce the unsharded
    // step bit for bit.  The kernel is HBM-bound; the few extra instructions are free.
#pragma clang fp contract(off)
    const float g = g_raw * gs + l2p * p;
    m = b1 * m + (1.f - b1) * g;
    v = b2 * v + (1.f - b2) * g * g;
    p -= step * m / (sqrtf(v) / sbc2 + eps);
    return p * p;
}
__global__ __launch_bounds__(256) void k_mt_adam(const MtChunk* ch, AdamHyper hp, float b1, float b2, float eps,
                                                 float l2, float gs, float* sumsq_part) {
    __shared__ float sh[4];
    const MtChunk c = ch[blockIdx.x];
    const float lr = hp.lr, bc1 = hp.bc1, sbc2 = hp.sbc2;     // sbc2 = sqrt(1 - beta2^t)
    const float step = lr / bc1, l2p = c.pad ? l2 : 0.f;
    float ss = 0.f;
    int i0 = 0;
    // 16-byte accesses where the chunk allows it (plain tensors whose four streams are 16-byte aligned): 30 bytes per parameter is all
    // this kernel does, and dword accesses leave ~10 % of the HBM rate on the table
    const bool vec = !c.cin && ((((size_t)c.a | (size_t)c.b | (size_t)c.c | (size_t)c.d) & 15) == 0) && (((size_t)c.s & 7) == 0);
    if (vec) {
        const int n4 = c.n >> 2;
        for (int i = threadIdx.x; i < n4; i += 256) {
            float4 p = reinterpret_cast<const float4*>(c.a)[i], m = reinterpret_cast<const float4*>(c.c)[i], v = reinterpret_cast<const float4*>(c.d)[i];
            const float4 g = reinterpret_cast<const float4*>(c.b)[i];
            ss += adam_elem(p.x, g.x, m.x, v.x, gs, l2p, b1, b2, step, sbc2, eps);
            ss += adam_elem(p.y, g.y, m.y, v.y, gs, l2p, b1, b2, step, sbc2, eps);
            ss += adam_elem(p.z, g.z, m.z, v.z, gs, l2p, b1, b2, step, sbc2, eps);
            ss += adam_elem(p.w, g.w, m.w, v.w, gs, l2p, b1, b2, step, sbc2, eps);
            reinterpret_cast<float4*>(c.c)[i] = m; reinterpret_cast<float4*>(c.d)[i] = v; reinterpret_cast<float4*>(c.a)[i] = p;
            if (c.s) {                                   // refresh the compute-dtype shadow in the same pass
                uint2 o;
                o.x = (uint32_t)f2bf(p.x) | ((uint32_t)f2bf(p.y) << 16);
                o.y = (uint32_t)f2bf(p.z) | ((uint32_t)f2bf(p.w) << 16);
                reinterpret_cast<uint2*>(c.s)[i] = o;
            }
        }
        i0 = n4 << 2;
    }
    for (int i = i0 + threadIdx.x; i < c.n; i += 256) {
        float p = c.a[i], m = c.c[i], v = c.d[i];
        ss += adam_elem(p, c.b[i], m, v, gs, l2p, b1, b2, step, sbc2, eps);
        c.c[i] = m; c.d[i] = v;
        c.a[i] = p;
        if (c.cin) {                                     // conv weight: both re-laid-out images, no separate relayout pass
            const long gi = c.off + i, r = gi / 5;
            const int j = (int)(gi - r * 5), co = (int)(r / c.cin), ci = (int)(r - (long)co * c.cin);
            const bf16_t pb = f2bf(p);
            c.s[(long)co * 5 * c.cin + (long)j * c.cin + ci] = pb;
            c.s2[(long)ci * 5 * c.cout + (long)(4 - j) * c.cout + co] = pb;
        } else if (c.s) c.s[i] = f2bf(p);
    }
    if (sumsq_part) {            // sum of squares of the UPDATED L2 members: the next step's regulariser value for free
        ss = block_sum_256(ss, sh);
        if (threadIdx.x == 0) sumsq_part[blockIdx.x] = c.pad ? ss : 0.f;
    }
}
// Second form of the same update (default; B2S_ADAM_V1 selects the one above): the chunk kind (no shadow / bf16 shadow / conv images) is
// workgroup-uniform, so each kind gets its own straight-line loop -- a store under a condition makes the compiler wait for ALL outstanding
// memory operations at the join (s_waitcnt vmcnt(0)), which serialised every iteration's loads behind the previous iteration's stores --
// and four float4 iterations of loads are in flight before the first is used.
typedef float f4v __attribute__((ext_vector_type(4)));
template <int KIND>
__device__ __forceinline__ void adam_store_shadow(const MtChunk& c, int i4, f4v p) {
    if (KIND == 1) {
        uint2 o;
        o.x = (uint32_t)f2bf(p[0]) | ((uint32_t)f2bf(p[1]) << 16);
        o.y = (uint32_t)f2bf(p[2]) | ((uint32_t)f2bf(p[3]) << 16);
        reinterpret_cast<uint2*>(c.s)[i4] = o;
    } else if (KIND == 2) {
        const uint32_t g0 = (uint32_t)c.off + 4u * (uint32_t)i4;
        const float inv_cin = __builtin_amdgcn_rcpf((float)c.cin);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t gi = g0 + e, r = __umulhi(gi, 0xCCCCCCCDu) >> 2;          // gi / 5
            const int j = (int)(gi - r * 5u);
            int co = (int)(((float)r + 0.5f) * inv_cin);                             // r / cin (r < 2^24: one correction step is enough)
            int ci = (int)r - co * c.cin;
            if (ci < 0) { --co; ci += c.cin; } else if (ci >= c.cin) { ++co; ci -= c.cin; }
            const bf16_t pb = f2bf(p[e]);
            c.s[(long)co * 5 * c.cin + (long)j * c.cin + ci] = pb;
            c.s2[(long)ci * 5 * c.cout + (long)(4 - j) * c.cout + co] = pb;
        }
    }
}
// gradient of 4 consecutive elements: the fp32 gradient buffer, or (gw != null) the bf16 wire buffer of the gradient exchange at the same
// element offsets -- the all-reduced sum is consumed where RCCL left it, no unpack pass and no fp32 re-read (b2s_adam_set_grad_wire)
template <bool W16>
__device__ __forceinline__ f4v adam_grad4(const f4v* G, const uint2* GW, int i4) {
    if (!W16) return __builtin_nontemporal_load(G + i4);
    const uint2 u = GW[i4];
    return (f4v){bf2f((bf16_t)(u.x & 0xffff)), bf2f((bf16_t)(u.x >> 16)), bf2f((bf16_t)(u.y & 0xffff)), bf2f((bf16_t)(u.y >> 16))};
}
// The optimizer streams 2.5 GB that nothing reads again before the next step: non-temporal loads / stores for the masters and both moments keep it from
// evicting what the kernels running beside it (the encoder backward) live on.  Six interleaved pairs, same box: 7.148 vs 7.179 ms per step, all six faster.
#ifndef B2S_ADAM_NT
#define B2S_ADAM_NT 1
#endif
#if B2S_ADAM_NT
#define B2S_NT_LD(p) __builtin_nontemporal_load(p)
#define B2S_NT_ST(v, p) __builtin_nontemporal_store(v, p)
#else
#define B2S_NT_LD(p) (*(p))
#define B2S_NT_ST(v, p) (*(p) = (v))
#endif
template <int KIND, bool W16>
__device__ __forceinline__ float adam_chunk(const MtChunk& c, const bf16_t* gw, float gs, float l2p, float b1, float b2, float step, float sbc2, float eps) {
    float ss = 0.f;
    const int n4 = c.n >> 2;
    f4v* P = reinterpret_cast<f4v*>(c.a); f4v* M = reinterpret_cast<f4v*>(c.c); f4v* V = reinterpret_cast<f4v*>(c.d);
    const f4v* G = reinterpret_cast<const f4v*>(c.b);
    const uint2* GW = reinterpret_cast<const uint2*>(gw);
    int i = threadIdx.x;
    for (; i + 768 < n4; i += 1024) {
        f4v p[4], g[4], m[4], v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { p[u] = B2S_NT_LD(P + i + u * 256); g[u] = adam_grad4<W16>(G, GW, i + u * 256); m[u] = B2S_NT_LD(M + i + u * 256); v[u] = B2S_NT_LD(V + i + u * 256); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { float pe = p[u][e], me = m[u][e], ve = v[u][e]; ss += adam_elem(pe, g[u][e], me, ve, gs, l2p, b1, b2, step, sbc2, eps); p[u][e] = pe; m[u][e] = me; v[u][e] = ve; }
            B2S_NT_ST(m[u], M + i + u * 256); B2S_NT_ST(v[u], V + i + u * 256); B2S_NT_ST(p[u], P + i + u * 256);
            adam_store_shadow<KIND>(c, i + u * 256, p[u]);
        }
    }
    for (; i < n4; i += 256) {
        f4v p = B2S_NT_LD(P + i), g = adam_grad4<W16>(G, GW, i), m = B2S_NT_LD(M + i), v = B2S_NT_LD(V + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) { float pe = p[e], me = m[e], ve = v[e]; ss += adam_elem(pe, g[e], me, ve, gs, l2p, b1, b2, step, sbc2, eps); p[e] = pe; m[e] = me; v[e] = ve; }
        B2S_NT_ST(m, M + i); B2S_NT_ST(v, V + i); B2S_NT_ST(p, P + i);
        adam_store_shadow<KIND>(c, i, p);
    }
    return ss;
}
template <bool W16>
__global__ __launch_bounds__(256) void k_mt_adam2(const MtChunk* __restrict__ ch, int nchunks, AdamHyper hp, float b1, float b2, float eps,
                                                  float l2, float gs, float* sumsq_part, const bf16_t* __restrict__ wire, const float* gbase) {
    __shared__ float sh[4];
    const float lr = hp.lr, bc1 = hp.bc1, sbc2 = hp.sbc2;
    // one chunk per workgroup, or (a capped grid: ro_mt_adam's max_wg) every gridDim.x-th chunk
    for (int cix = blockIdx.x; cix < nchunks; cix += gridDim.x) {
        const MtChunk c = ch[cix];
        const bf16_t* gw = W16 ? wire + (c.b - gbase) : nullptr;
        const float step = lr / bc1, l2p = c.pad ? l2 : 0.f;
        float ss = 0.f;
        int i0 = 0;
        const bool vec = ((((size_t)c.a | (size_t)c.b | (size_t)c.c | (size_t)c.d) & 15) == 0) && (c.cin || ((size_t)c.s & 7) == 0) && (!W16 || ((size_t)gw & 7) == 0);
        if (vec) {
            if (c.cin) ss = adam_chunk<2, W16>(c, gw, gs, l2p, b1, b2, step, sbc2, eps);
            else if (c.s) ss = adam_chunk<1, W16>(c, gw, gs, l2p, b1, b2, step, sbc2, eps);
            else ss = adam_chunk<0, W16>(c, gw, gs, l2p, b1, b2, step, sbc2, eps);
            i0 = (c.n >> 2) << 2;
        }
        for (int i = i0 + threadIdx.x; i < c.n; i += 256) {       // unaligned tensors and the last 0-3 elements of a chunk
            float p = c.a[i], m = c.c[i], v = c.d[i];
            ss += adam_elem(p, W16 ? bf2f(gw[i]) : c.b[i], m, v, gs, l2p, b1, b2, step, sbc2, eps);
            c.c[i] = m; c.d[i] = v;
            c.a[i] = p;
            if (c.cin) {
                const long gi = c.off + i, r = gi / 5;
                const int j = (int)(gi - r * 5), co = (int)(r / c.cin), ci = (int)(r - (long)co * c.cin);
                const bf16_t pb = f2bf(p);
                c.s[(long)co * 5 * c.cin + (long)j * c.cin + ci] = pb;
                c.s2[(long)ci * 5 * c.cout + (long)(4 - j) * c.cout + co] = pb;
            } else if (c.s) c.s[i] = f2bf(p);
        }
        if (sumsq_part) {
            ss = block_sum_256(ss, sh);
            if (threadIdx.x == 0) sumsq_part[cix] = c.pad ? ss : 0.f;
        }
    }
}
// out[0] = scale * sum(part[0..n))
// (+ clears nzero floats at zero: the loss kernels' accumulators -- a hipMemsetAsync of an odd-sized, 4-byte-aligned range is up to three
// 5 us fill kernels on the critical path)
__global__ __launch_bounds__(1024) void k_sum_scaled(const float* part, int n, float scale, float* out, float* zero, int nzero) {
    __shared__ float sh[16];
    for (int i = threadIdx.x; i < nzero; i += 1024) zero[i] = 0.f;
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) acc += part[i];
    acc = block_sum_256(acc, sh);
    if (threadIdx.x == 0) out[0] = acc * scale;
}

template <typename T>
__global__ void k_conv_w_relayout(const float* w, T* wf, T* wb, int Cout, int Cin) {
    const long total = (long)Cout * Cin * 5;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int j = (int)(i % 5); long r = i / 5; int ci = (int)(r % Cin), co = (int)(r / Cin);
        float v = w[i];
        TT<T>::st(wf + (long)co * 5 * Cin + (long)j * Cin + ci, v);
        TT<T>::st(wb + (long)ci * 5 * Cout + (long)(4 - j) * Cout + co, v);
    }
}
__global__ void k_add(const float* a, const float* b, float* o, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) o[i] = a[i] + b[i];
}
__global__ void k_add3(const float* a, const float* b, const float* c, float* o, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) o[i] = (a[i] + b[i]) + c[i];
}
__global__ void k_fill(float* p, float v, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}

inline int ew_grid(long n, int per = 256) { long g = (n + per - 1) / per; return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }

}  // namespace

// ====================================================================================== host wrappers
int ro_embed_prep_fwd(const long* ids, const int* lens, const float* embed, const float* pe, const float* pe_scale,
                      float* x, int B, int S, int D, DropCfg drop, hipStream_t st) {
    B2S_CHECK(D % 4 == 0, "embed_prep: D=%d must be a multiple of 4", D);
    hipLaunchKernelGGL(k_embed_prep_fwd, dim3(B * S), dim3(128), 0, st, ids, lens, embed, pe, pe_scale, x, S, D, drop);
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_embed_prep_bwd(const float* dx, const long* ids, const int* lens, const float* pe, float* d_embed,
                      float* d_pe_scale, int B, int S, int D, DropCfg drop, hipStream_t st, int dx_bf16) {
    if (dx_bf16) hipLaunchKernelGGL(k_embed_prep_bwd<bf16_t>, dim3(B * S), dim3(128), 0, st, (const bf16_t*)dx, ids, lens, pe, d_embed, d_pe_scale, S, D, drop);
    else hipLaunchKernelGGL(k_embed_prep_bwd<float>, dim3(B * S), dim3(128), 0, st, dx, ids, lens, pe, d_embed, d_pe_scale, S, D, drop);
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_layernorm_fwd(int dtype, const float* x, const float* gamma, const float* beta, void* y, int ldy, float* y32,
                     int ldy32, float* mean, float* rstd, int M, int D, float eps, const int* row_len,
                     int rows_per_batch, hipStream_t st) {
    B2S_CHECK(D % 4 == 0 && D <= 1024, "layernorm: D=%d must be a multiple of 4 and <= 1024", D);
    constexpr bool no_fast = false;             // A/B switch
    if (!no_fast && (D == 768 || D == 512) && M > 0) {
        if (D == 768) RO_DISPATCH(dtype, hipLaunchKernelGGL((k_ln_fwd_fast<TY, 3>), dim3(cdiv(M, 4)), dim3(256), 0, st, x, gamma, beta, (TY*)y, ldy,
                                                            y32, ldy32, mean, rstd, M, eps, row_len, rows_per_batch));
        else RO_DISPATCH(dtype, hipLaunchKernelGGL((k_ln_fwd_fast<TY, 2>), dim3(cdiv(M, 4)), dim3(256), 0, st, x, gamma, beta, (TY*)y, ldy,
                                                   y32, ldy32, mean, rstd, M, eps, row_len, rows_per_batch));
        B2S_LAUNCH_CHECK(); return 0;
    }
    RO_DISPATCH(dtype, hipLaunchKernelGGL((k_ln_fwd<TY>), dim3(cdiv(M, 4)), dim3(256), 0, st, x, gamma, beta, (TY*)y, ldy,
                                          y32, ldy32, mean, rstd, M, D, eps, row_len, rows_per_batch));
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_layernorm_bwd(int dtype, const void* dy, int dy_fp32, int lddy, const float* x, const float* gamma,
                     const float* mean, const float* rstd, float* dx, int accumulate, float* dgamma, float* dbeta,
                     int M, int D, const int* row_len, int rows_per_batch, hipStream_t st, float* ws, void* dy2, DropCfg drop2,
                     int* defer_nblk, int dx_bf16) {
    B2S_CHECK(D % 4 == 0 && D <= 1024, "layernorm: D=%d must be a multiple of 4 and <= 1024", D);
    B2S_CHECK(!dx_bf16 || ((D == 768 || D == 512) && (lddy & 3) == 0 && true), "layernorm backward: a bf16 residual gradient needs the D = 512 / 768 kernels");
    int grid = cdiv(M, 4); if (grid > (ws ? RO_LN_WS_ROWS : 512)) grid = ws ? RO_LN_WS_ROWS : 512;
    constexpr bool no_fast = false;             // A/B switch
    const bool f32 = dy_fp32 || !dtype;
    if (!no_fast && (D == 768 || D == 512) && M > 0 && (lddy & 3) == 0) {
        // ~3 rows per wave (the next row's loads fly under the current row's reductions); at most RO_LN_WS_ROWS partial rows
        constexpr int rows_per_wg = 12;
        grid = std::max(1, std::min(cdiv(M, rows_per_wg), ws ? RO_LN_WS_ROWS : 512));
#define B2S_LN_FAST(TD, NCH, ACC, DY2) do { if (dx_bf16) hipLaunchKernelGGL((k_ln_bwd_fast<TD, NCH, ACC, DY2, bf16_t>), dim3(grid), dim3(256), 0, st, (const TD*)dy, lddy, x, \
            gamma, mean, rstd, (bf16_t*)dx, M, row_len, rows_per_batch, ws, dgamma, dbeta, (bf16_t*)dy2, drop2); \
        else hipLaunchKernelGGL((k_ln_bwd_fast<TD, NCH, ACC, DY2, float>), dim3(grid), dim3(256), 0, st, (const TD*)dy, lddy, x, \
            gamma, mean, rstd, dx, M, row_len, rows_per_batch, ws, dgamma, dbeta, (bf16_t*)dy2, drop2); } while (0)
#define B2S_LN_FAST_AD(TD, NCH) do { if (accumulate) { if (dy2) B2S_LN_FAST(TD, NCH, true, true); else B2S_LN_FAST(TD, NCH, true, false); } \
                                     else { if (dy2) B2S_LN_FAST(TD, NCH, false, true); else B2S_LN_FAST(TD, NCH, false, false); } } while (0)
        if (D == 768) { if (f32) B2S_LN_FAST_AD(float, 3); else B2S_LN_FAST_AD(bf16_t, 3); }
        else          { if (f32) B2S_LN_FAST_AD(float, 2); else B2S_LN_FAST_AD(bf16_t, 2); }
#undef B2S_LN_FAST_AD
#undef B2S_LN_FAST
    } else if (dy_fp32 || !dtype)
        hipLaunchKernelGGL((k_ln_bwd<float>), dim3(grid), dim3(256), 0, st, (const float*)dy, lddy, x, gamma, mean, rstd,
                           dx, accumulate, dgamma, dbeta, M, D, row_len, rows_per_batch, ws, (bf16_t*)dy2, drop2);
    else
        hipLaunchKernelGGL((k_ln_bwd<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)dy, lddy, x, gamma, mean,
                           rstd, dx, accumulate, dgamma, dbeta, M, D, row_len, rows_per_batch, ws, (bf16_t*)dy2, drop2);
    if (ws && defer_nblk) *defer_nblk = grid;
    else if (ws) {
        int gy = cdiv(grid, 32); if (gy < 1) gy = 1;
        hipLaunchKernelGGL(k_ln_param_reduce, dim3(cdiv(2 * D, 64), gy), dim3(256), 0, st, (const float*)ws, grid, D, dgamma, dbeta);
    }
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_ln_param_reduce_batch(const LnReduceBatch& b, hipStream_t st) {
    if (b.n <= 0) return 0;
    int dmax = 0, nmax = 0;
    for (int i = 0; i < b.n; ++i) { dmax = std::max(dmax, b.j[i].D); nmax = std::max(nmax, b.j[i].nblk); }
    int gy = cdiv(nmax, 32); if (gy < 1) gy = 1;
    hipLaunchKernelGGL(k_ln_param_reduce_batch, dim3(cdiv(2 * dmax, 64), gy, b.n), dim3(256), 0, st, b);
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_softmax_fwd(int dtype, const float* S, void* P, void* Pd, int B, int H, int Lq, int Lk, int ldp, float scale,
                   int mask_mode, const int* klen, const float* bias, long bias_sb, long bias_sq, DropCfg drop,
                   hipStream_t st) {
    const long rows = (long)B * H * Lq;
    B2S_CHECK(!(mask_mode & 1) || klen, "softmax: key-length mask needs klen");
    RO_DISPATCH(dtype, hipLaunchKernelGGL((k_softmax_fwd<TY>), dim3(cdiv(rows, 4)), dim3(256), 0, st, S, (TY*)P, (TY*)Pd, H,
                                          Lq, Lk, ldp, rows, scale, mask_mode, klen, bias, bias_sb, bias_sq, drop));
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_softmax_bwd(int dtype, const void* P, const float* dPraw, void* dS, int B, int H, int Lq, int Lk, int ldp,
                   float scale, DropCfg drop, hipStream_t st) {
    const long rows = (long)B * H * Lq;
    RO_DISPATCH(dtype, hipLaunchKernelGGL((k_softmax_bwd<TY>), dim3(cdiv(rows, 4)), dim3(256), 0, st, (const TY*)P, dPraw,
                                          (TY*)dS, Lk, ldp, rows, scale, drop));
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_align_transpose(int dtype, const void* P, float* align, int Z, int Lq, int Lk, int ldp, hipStream_t st) {
    dim3 grid(cdiv(Lk, 32), cdiv(Lq, 32), Z);
    RO_DISPATCH(dtype, hipLaunchKernelGGL((k_align_transpose<TY>), grid, dim3(256), 0, st, (const TY*)P, align, Lq, Lk, ldp));
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_cast_drop(int dtype, const float* in, int ldin, void* out, int ldo, int M, int N, DropCfg drop, hipStream_t st) {
    B2S_CHECK(N % 4 == 0, "cast_drop: N=%d must be a multiple of 4", N);
    RO_DISPATCH(dtype, hipLaunchKernelGGL((k_cast_drop<TY>), dim3(ew_grid((long)M * N / 4)), dim3(256), 0, st, in, ldin,
                                          (TY*)out, ldo, M, N, drop));
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_cast(int dtype, const float* in, void* out, long n, hipStream_t st) {
    if (dtype == 1 && n >= 8 && (((size_t)in | (size_t)out) & 15) == 0) {
        hipLaunchKernelGGL(k_cast_bf16_v, dim3(ew_grid(n >> 3)), dim3(256), 0, st, in, (bf16_t*)out, n);
        B2S_LAUNCH_CHECK(); return 0;
    }
    RO_DISPATCH(dtype, hipLaunchKernelGGL((k_cast<TY>), dim3(ew_grid(n)), dim3(256), 0, st, in, (TY*)out, n));
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_cast_back(int dtype, const void* in, float* out, long n, hipStream_t st) {
    if (dtype == 1 && n >= 8 && (((size_t)in | (size_t)out) & 15) == 0) {
        hipLaunchKernelGGL(k_cast_back_bf16_v, dim3(ew_grid(n >> 3)), dim3(256), 0, st, (const bf16_t*)in, out, n);
        B2S_LAUNCH_CHECK(); return 0;
    }
    RO_DISPATCH(dtype, hipLaunchKernelGGL((k_cast_back<TY>), dim3(ew_grid(n)), dim3(256), 0, st, (const TY*)in, out, n));
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_shift_pe_fwd(const float* a, const int* lens, const float* pe, const float* pe_scale, float* x, int B, int T,
                    int D, DropCfg drop, hipStream_t st, const int* off, int rows) {
    B2S_CHECK(D % 4 == 0, "shift_pe: D=%d must be a multiple of 4", D);
    hipLaunchKernelGGL(k_shift_pe_fwd, dim3(off ? rows : B * T), dim3(128), 0, st, a, lens, pe, pe_scale, x, T, D, drop, off, B);
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_shift_pe_bwd(int dtype, const float* dx, const int* lens, const float* pe, void* da, float* d_pe_scale, int B,
                    int T, int D, DropCfg drop, hipStream_t st, int dx_bf16, const int* off, int rows) {
    const int n = off ? rows : B * T;
    if (dx_bf16) RO_DISPATCH(dtype, hipLaunchKernelGGL((k_shift_pe_bwd<TY, bf16_t>), dim3(std::min(cdiv(n, 8), 256)), dim3(512), 0, st, (const bf16_t*)dx, lens,
                                                       pe, (TY*)da, d_pe_scale, T, D, drop, n, off, B));
    else RO_DISPATCH(dtype, hipLaunchKernelGGL((k_shift_pe_bwd<TY, float>), dim3(std::min(cdiv(n, 8), 256)), dim3(512), 0, st, dx, lens, pe, (TY*)da,
                                          d_pe_scale, T, D, drop, n, off, B));
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_set_rowoff(const int* off_host, int n, int* dst, hipStream_t st) {
    B2S_CHECK(n >= 2 && n <= 65, "ragged rows: 1 .. 64 utterances (got %d)", n - 1);
    RowOff64 h;
    for (int i = 0; i < 65; ++i) h.off[i] = i < n ? off_host[i] : 0;
    hipLaunchKernelGGL(k_set_rowoff, dim3(1), dim3(128), 0, st, h, dst, n);
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_rows_gather(int out_dtype, const float* in, void* out, const int* off, int B, int T, int C, hipStream_t st, const float* in2, float* out2, int C2) {
    if (out_dtype == 1) hipLaunchKernelGGL((k_rows_gather<bf16_t>), dim3(T, B), dim3(C >= 128 ? 128 : 64), 0, st, in, (bf16_t*)out, off, T, C, in2, out2, C2);
    else hipLaunchKernelGGL((k_rows_gather<float>), dim3(T, B), dim3(C >= 128 ? 128 : 64), 0, st, in, (float*)out, off, T, C, in2, out2, C2);
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_heads_scatter(int dtype, const void* x, int ldx, const float* w, const float* bias, const float* mel_in, float* mel_out, float* stop_out,
                     const int* off, int B, int T, int C, int D, hipStream_t st) {
    B2S_CHECK(D % 4 == 0, "heads: D=%d must be a multiple of 4", D);
    RO_DISPATCH(dtype, hipLaunchKernelGGL((k_heads_scatter<TY>), dim3(T, B), dim3(64), 0, st, (const TY*)x, ldx, w, bias, mel_in, mel_out, stop_out, off, T, C, D));
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_spk_embed_fwd(const long* spk_ids, const float* table, const float* W, const float* b, float* e_raw,
                     float* h_pre, float* mem32, void* memT, int dtype, int ldm, int col0, int B, int S, int E,
                     hipStream_t st) {
    B2S_CHECK(E <= 1024, "embedding size %d too large", E);
    int th = ((E + 63) / 64) * 64;
    RO_DISPATCH(dtype, hipLaunchKernelGGL((k_embed_net_fwd<TY>), dim3(B), dim3(th), E * sizeof(float), st, spk_ids, table,
                                          (const float*)nullptr, 0, (const float*)nullptr, W, b, e_raw, h_pre, mem32,
                                          (TY*)memT, ldm, col0, S, E));
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_lang_embed_fwd(const float* vecs, int L, const float* Wl, const float* W, const float* b, float* e_raw,
                      float* h_pre, float* mem32, void* memT, int dtype, int ldm, int col0, int B, int S, int E,
                      hipStream_t st) {
    B2S_CHECK(E <= 1024, "embedding size %d too large", E);
    int th = ((E + 63) / 64) * 64;
    RO_DISPATCH(dtype, hipLaunchKernelGGL((k_embed_net_fwd<TY>), dim3(B), dim3(th), E * sizeof(float), st,
                                          (const long*)nullptr, (const float*)nullptr, vecs, L, Wl, W, b, e_raw, h_pre,
                                          mem32, (TY*)memT, ldm, col0, S, E));
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_spk_embed_bwd(const float* dmem, int ldm, int col0, const long* spk_ids, const float* e_raw, const float* h_pre,
                     const float* W, float* d_table, float* dW, float* db, float* dh_scratch, int B, int S, int E, hipStream_t st) {
    int th = ((E + 63) / 64) * 64;
    float* de = dh_scratch + (long)B * E;
    hipLaunchKernelGGL(k_embed_net_bwd_dh, dim3(B), dim3(E * std::max(1, 1024 / E)), 0, st, dmem, ldm, col0, h_pre, dh_scratch, S, E);
    hipLaunchKernelGGL(k_embed_net_bwd_de, dim3(B), dim3(th), E * sizeof(float), st, (const float*)dh_scratch, spk_ids, W, de, d_table, E);
    hipLaunchKernelGGL(k_embed_net_bwd_w, dim3(E), dim3(th), 0, st, (const float*)dh_scratch, (const float*)de, e_raw,
                       (const float*)nullptr, 0, dW, db, (float*)nullptr, B, E);
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_lang_embed_bwd(const float* dmem, int ldm, int col0, const float* vecs, int L, const float* e_raw,
                      const float* h_pre, const float* Wl, const float* W, float* dWl, float* dW, float* db, float* dh_scratch,
                      int B, int S, int E, hipStream_t st) {
    (void)Wl;
    int th = ((std::max(E, L) + 63) / 64) * 64;
    float* de = dh_scratch + (long)B * E;
    hipLaunchKernelGGL(k_embed_net_bwd_dh, dim3(B), dim3(E * std::max(1, 1024 / E)), 0, st, dmem, ldm, col0, h_pre, dh_scratch, S, E);
    hipLaunchKernelGGL(k_embed_net_bwd_de, dim3(B), dim3(th), E * sizeof(float), st, (const float*)dh_scratch, (const long*)nullptr, W, de,
                       (float*)nullptr, E);
    hipLaunchKernelGGL(k_embed_net_bwd_w, dim3(E), dim3(th), 0, st, (const float*)dh_scratch, (const float*)de, e_raw, vecs, L, dW, db, dWl,
                       B, E);
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_rowdot_fwd(int dtype, const void* x, int ldx, const float* w, const float* b, float* out, int M, int D,
                  const int* row_len, int rows_per_batch, hipStream_t st) {
    B2S_CHECK(D % 4 == 0, "rowdot: D=%d must be a multiple of 4", D);
    RO_DISPATCH(dtype, hipLaunchKernelGGL((k_rowdot<TY>), dim3(cdiv(M, 4)), dim3(256), 0, st, (const TY*)x, ldx, w, b, out,
                                          M, D, row_len, rows_per_batch));
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_colsum(int dtype, const void* X, int x_fp32, int ldx, const float* wgt, float* out, int accumulate, int M,
              int C, hipStream_t st) {
    if (!accumulate) B2S_HIP(hipMemsetAsync(out, 0, sizeof(float) * C, st));
    int gy = cdiv(M, 4 * 16); if (gy > 128) gy = 128; if (gy < 1) gy = 1;
    dim3 grid(cdiv(C, 64), gy);
    if (x_fp32 || !dtype)
        hipLaunchKernelGGL((k_colsum<float>), grid, dim3(256), 0, st, (const float*)X, ldx, wgt, out, M, C);
    else
        hipLaunchKernelGGL((k_colsum<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)X, ldx, wgt, out, M, C);
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_bn_stats(const float* y, int M, int C, float* mean, float* rstd, float eps, float* running_mean,
                float* running_var, long* num_batches_tracked, float momentum, float* scratch, hipStream_t st) {
    B2S_HIP(hipMemsetAsync(scratch, 0, sizeof(float) * 2 * C, st));
    int gy = cdiv(M, 4 * 16); if (gy > 128) gy = 128; if (gy < 1) gy = 1;
    dim3 grid(cdiv(C, 64), gy);
    hipLaunchKernelGGL(k_bn_colred, grid, dim3(256), 0, st, y, M, C, (const float*)nullptr, scratch);
    hipLaunchKernelGGL(k_bn_colred, grid, dim3(256), 0, st, y, M, C, (const float*)scratch, scratch + C);
    hipLaunchKernelGGL(k_bn_finalize, dim3(cdiv(C, 256)), dim3(256), 0, st, (const float*)scratch, M, C, mean, rstd, eps,
                       running_mean, running_var, num_batches_tracked, momentum);
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_bn_eval_stats(const float* running_mean, const float* running_var, float* mean, float* rstd, float eps, int C,
                     hipStream_t st) {
    hipLaunchKernelGGL(k_bn_eval_stats, dim3(cdiv(C, 256)), dim3(256), 0, st, running_mean, running_var, mean, rstd, eps, C);
    B2S_LAUNCH_CHECK(); return 0;
}
// training forward from the column sums the conv GEMM's epilogue left in sums[2C] (GemmEpilogue::colstat): statistics, running-statistics
// update and the normalisation in one launch
int ro_bn_apply_train(int dtype, const float* y, const float* sums, float* mean, float* rstd, float eps, float* running_mean, float* running_var,
                      long* num_batches_tracked, float momentum, const float* gamma, const float* beta, int use_tanh, void* outT, float* out32,
                      const float* add32, int M, int C, DropCfg drop, hipStream_t st) {
    B2S_CHECK(C % 4 == 0 && M > 1, "bn_apply_train: C=%d must be a multiple of 4, M=%d > 1", C, M);
    BnStat bs = {sums, mean, rstd, running_mean, running_var, num_batches_tracked, eps, momentum};
    dim3 grid(cdiv(C / 4, 64), cdiv(M, 4 * BN_RU));
    RO_DISPATCH(dtype, hipLaunchKernelGGL((k_bn_apply_v<TY>), grid, dim3(256), 0, st, y, bs, gamma, beta, use_tanh, (TY*)outT, out32, add32, M, C, drop));
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_bn_apply(int dtype, const float* y, const float* mean, const float* rstd, const float* gamma,
                const float* beta, int use_tanh, void* outT, float* out32, const float* add32, int M, int C,
                DropCfg drop, hipStream_t st) {
    B2S_CHECK(C % 4 == 0, "bn_apply: C=%d must be a multiple of 4", C);
    constexpr bool scalar = false;               // A/B switch: the round-2 kernels
    if (!scalar) {
        BnStat bs = {nullptr, const_cast<float*>(mean), const_cast<float*>(rstd), nullptr, nullptr, nullptr, 0.f, 0.f};
        dim3 grid(cdiv(C / 4, 64), cdiv(M, 4 * BN_RU));
        RO_DISPATCH(dtype, hipLaunchKernelGGL((k_bn_apply_v<TY>), grid, dim3(256), 0, st, y, bs, gamma, beta, use_tanh, (TY*)outT, out32, add32, M, C, drop));
        B2S_LAUNCH_CHECK(); return 0;
    }
    RO_DISPATCH(dtype, hipLaunchKernelGGL((k_bn_apply<TY>), dim3(ew_grid((long)M * C / 4)), dim3(256), 0, st, y, mean, rstd,
                                          gamma, beta, use_tanh, (TY*)outT, out32, add32, M, C, drop));
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_bn_bwd(int dtype, const void* dout, int dout_fp32, const float* y, const float* mean, const float* rstd,
              const float* gamma, const float* beta, int use_tanh, float* dgamma, float* dbeta, void* dyT, int M,
              int C, DropCfg drop, hipStream_t st) {
    constexpr bool scalar = false;               // A/B switch: the round-2 kernels
    if (!scalar && C % 4 == 0) {
        dim3 gv(cdiv(C / 4, 64), cdiv(M, 4 * BN_RU));
#define B2S_BN_BWD(TD, T) do { \
        hipLaunchKernelGGL((k_bn_bwd_v<TD, T, false>), gv, dim3(256), 0, st, (const TD*)dout, y, mean, rstd, gamma, beta, use_tanh, dgamma, dbeta, (T*)dyT, M, C, drop); \
        hipLaunchKernelGGL((k_bn_bwd_v<TD, T, true>), gv, dim3(256), 0, st, (const TD*)dout, y, mean, rstd, gamma, beta, use_tanh, dgamma, dbeta, (T*)dyT, M, C, drop); } while (0)
        if (dout_fp32 || !dtype) { if (dtype) B2S_BN_BWD(float, bf16_t); else B2S_BN_BWD(float, float); }
        else B2S_BN_BWD(bf16_t, bf16_t);
#undef B2S_BN_BWD
        B2S_LAUNCH_CHECK(); return 0;
    }
    int gy = cdiv(M, 4 * 16); if (gy > 128) gy = 128; if (gy < 1) gy = 1;
    dim3 grid(cdiv(C, 64), gy);
    const int g2 = ew_grid((long)M * C);
    if (dout_fp32 || !dtype) {
        hipLaunchKernelGGL((k_bn_bwd_red<float>), grid, dim3(256), 0, st, (const float*)dout, y, mean, rstd, gamma, beta,
                           use_tanh, dgamma, dbeta, M, C, drop);
        RO_DISPATCH(dtype, hipLaunchKernelGGL((k_bn_bwd_apply<float, TY>), dim3(g2), dim3(256), 0, st, (const float*)dout,
                                              y, mean, rstd, gamma, beta, use_tanh, (const float*)dgamma,
                                              (const float*)dbeta, (TY*)dyT, M, C, drop));
    } else {
        hipLaunchKernelGGL((k_bn_bwd_red<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)dout, y, mean, rstd, gamma,
                           beta, use_tanh, dgamma, dbeta, M, C, drop);
        hipLaunchKernelGGL((k_bn_bwd_apply<bf16_t, bf16_t>), dim3(g2), dim3(256), 0, st, (const bf16_t*)dout, y, mean,
                           rstd, gamma, beta, use_tanh, (const float*)dgamma, (const float*)dbeta, (bf16_t*)dyT, M, C,
                           drop);
    }
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_loss_fwd(const float* bef, const float* aft, const float* stop, const float* tgt, const int* lens,
                const float* l2, float* out, float* aft_losses, int B, int T, int C, float pos_weight, float* scratch,
                hipStream_t st, bool scratch_zeroed) {
    if (!scratch_zeroed) B2S_HIP(hipMemsetAsync(scratch, 0, sizeof(float) * (3 + B), st));
    if (C % 4 == 0)
        hipLaunchKernelGGL(k_loss_partial_v, dim3(std::max(1, std::min(cdiv((long)T * C / 4, 1024), 16)), B), dim3(256), 0, st, bef, aft, stop, tgt, lens,
                           scratch, T, C, pos_weight);
    else
        hipLaunchKernelGGL(k_loss_partial, dim3(std::min(cdiv(T, 4), 32), B), dim3(256), 0, st, bef, aft, stop, tgt, lens, scratch, B,
                           T, C, pos_weight);
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(64), 0, st, (const float*)scratch, lens, l2, out, aft_losses, B);
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_loss_bwd(const float* bef, const float* aft, const float* stop, const float* tgt, const int* lens,
                const float* gscale, float* d_bef, float* d_aft, float* d_stop, int B, int T, int C, float pos_weight,
                hipStream_t st) {
    hipLaunchKernelGGL(k_loss_bwd, dim3(cdiv((long)B * T, 4)), dim3(256), 0, st, bef, aft, stop, tgt, lens, gscale, d_bef,
                       d_aft, d_stop, B, T, C, pos_weight);
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_mt_sumsq(const MtChunk* chunks, int nchunks, float* out, float scale, hipStream_t st) {
    if (nchunks > 0) hipLaunchKernelGGL(k_mt_sumsq, dim3(nchunks), dim3(256), 0, st, chunks, out, scale);
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_mt_axpy(const MtChunk* chunks, int nchunks, float alpha, const float* gscale, hipStream_t st) {
    if (nchunks > 0) hipLaunchKernelGGL(k_mt_axpy, dim3(nchunks), dim3(256), 0, st, chunks, alpha, gscale);
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_mt_adam(const MtChunk* chunks, int nchunks, AdamHyper hp, float beta1, float beta2, float eps, float l2,
               float grad_scale, float* sumsq_part, hipStream_t st, const void* wire, const float* gbase, int max_wg) {
    constexpr bool v1 = false;
    const int grid = max_wg > 0 ? std::min(max_wg, nchunks) : nchunks;
    if (nchunks > 0 && wire)
        hipLaunchKernelGGL(k_mt_adam2<true>, dim3(grid), dim3(256), 0, st, chunks, nchunks, hp, beta1, beta2, eps, l2, grad_scale, sumsq_part, (const bf16_t*)wire, gbase);
    else if (nchunks > 0 && v1)
        hipLaunchKernelGGL(k_mt_adam, dim3(nchunks), dim3(256), 0, st, chunks, hp, beta1, beta2, eps, l2, grad_scale, sumsq_part);
    else if (nchunks > 0)
        hipLaunchKernelGGL(k_mt_adam2<false>, dim3(grid), dim3(256), 0, st, chunks, nchunks, hp, beta1, beta2, eps, l2, grad_scale, sumsq_part, (const bf16_t*)nullptr,
                           (const float*)nullptr);
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_sum_scaled(const float* part, int n, float scale, float* out, hipStream_t st, float* zero, int nzero) {
    hipLaunchKernelGGL(k_sum_scaled, dim3(1), dim3(1024), 0, st, part, n, scale, out, zero, nzero);
    B2S_LAUNCH_CHECK(); return 0;
}
__global__ __launch_bounds__(256) void k_im2col5(const uint4* __restrict__ x, const int* __restrict__ lens, int T, int c8, uint4* __restrict__ out, long M) {
    const long total = M * 5 * c8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / (5 * c8);
        const int rem = (int)(i - m * 5 * c8), j = rem / c8, c = rem - j * c8;
        const int b = (int)(m / T), t = (int)(m - (long)b * T), ts = t + j - 2;
        const int lim = lens ? min(lens[b], T) : T;
        const bool ok = (unsigned)ts < (unsigned)lim;
        const uint4 v = x[(ok ? m + j - 2 : m) * c8 + c];          // (unconditional load on a clamped row)
        out[i] = ok ? v : make_uint4(0, 0, 0, 0);
    }
}
int ro_im2col5(int dtype, const void* x, const int* lens, int T, int cin, void* out, long M, hipStream_t st) {
    B2S_CHECK(dtype == 1 && cin % 8 == 0 && T > 0, "im2col: bf16 rows of a multiple of 8 channels only");
    const long total = M * 5 * (cin / 8);
    hipLaunchKernelGGL(k_im2col5, dim3((int)std::min<long>(cdiv(total, 256), 4096)), dim3(256), 0, st, (const uint4*)x, lens, T, cin / 8, (uint4*)out, M);
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_conv_w_relayout(int dtype, const float* w, void* wf, void* wb, int Cout, int Cin, hipStream_t st) {
    RO_DISPATCH(dtype, hipLaunchKernelGGL((k_conv_w_relayout<TY>), dim3(ew_grid((long)Cout * Cin * 5)), dim3(256), 0, st, w,
                                          (TY*)wf, (TY*)wb, Cout, Cin));
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_add3(const float* a, const float* b, const float* c, float* out, long n, hipStream_t st) {
    hipLaunchKernelGGL(k_add3, dim3(ew_grid(n)), dim3(256), 0, st, a, b, c, out, n);
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_add(const float* a, const float* b, float* out, long n, hipStream_t st) {
    hipLaunchKernelGGL(k_add, dim3(ew_grid(n)), dim3(256), 0, st, a, b, out, n);
    B2S_LAUNCH_CHECK(); return 0;
}
// s[0..n) = bf16(a[0..n)) for every chunk of a table: the bf16 shadows of all GEMM weights in ONE launch (b2s_model_sync_weights after the parameters
// changed behind the optimizer's back -- every step of the reference's own loop, where torch.optim.Adam owns the update: 74 cast launches before)
__global__ __launch_bounds__(256) void k_mt_cast(const MtChunk* __restrict__ ch) {
    const MtChunk c = ch[blockIdx.x];
    int i0 = 0;
    if (((((size_t)c.a) & 15) == 0) && ((((size_t)c.s) & 7) == 0)) {
        const float4* P = reinterpret_cast<const float4*>(c.a);
        uint2* S = reinterpret_cast<uint2*>(c.s);
        for (int i = threadIdx.x; i < (c.n >> 2); i += 256) {
            const float4 v = P[i];
            uint2 u; u.x = f2bf2(v.x, v.y); u.y = f2bf2(v.z, v.w);
            S[i] = u;
        }
        i0 = (c.n >> 2) << 2;
    }
    for (int i = i0 + threadIdx.x; i < c.n; i += 256) c.s[i] = f2bf(c.a[i]);
}
int ro_mt_cast(const MtChunk* chunks, int nchunks, hipStream_t st) {
    if (nchunks > 0) hipLaunchKernelGGL(k_mt_cast, dim3(nchunks), dim3(256), 0, st, chunks);
    B2S_LAUNCH_CHECK(); return 0;
}
// out-of-line: zero the `a` ranges of a chunk table (a, n) in one launch (the gradient ranges that still accumulate: a memset per range would
// be ~40 fill kernels of 3-5 us)
__global__ __launch_bounds__(256) void k_mt_zero(const MtChunk* __restrict__ ch) {
    const MtChunk c = ch[blockIdx.x];
    const bool vec = (((size_t)c.a) & 15) == 0;
    int i0 = 0;
    if (vec) {
        float4* p = reinterpret_cast<float4*>(c.a);
        for (int i = threadIdx.x; i < (c.n >> 2); i += 256) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        i0 = (c.n >> 2) << 2;
    }
    for (int i = i0 + threadIdx.x; i < c.n; i += 256) c.a[i] = 0.f;
}
// Parameter wire of the sharded optimizer (engine.hip: b2s_param_wire): a flat fp32 buffer laid out like the flat gradient buffer that starts at gbase.
//   PACK     wire[c.b - gbase + i] = c.a[i]                (the parameters this rank just updated, ahead of the all-gather)
//   SCATTER  c.a[i] = wire[...] + the compute-dtype shadow / the two conv images, exactly as the Adam kernel writes them
//            (the parameters the OTHER ranks updated, after the all-gather)
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_mt_param_wire(const MtChunk* __restrict__ ch, float* __restrict__ wire, const float* gbase) {
    const MtChunk c = ch[blockIdx.x];
    float* w = wire + (c.b - gbase);
    int i0 = 0;
    const bool vec = ((((size_t)c.a | (size_t)w) & 15) == 0) && (c.cin || ((size_t)c.s & 7) == 0);
    if (vec) {
        f4v* P = reinterpret_cast<f4v*>(c.a); f4v* W = reinterpret_cast<f4v*>(w);
        for (int i = threadIdx.x; i < (c.n >> 2); i += 256) {
            if (!SCATTER) { W[i] = P[i]; continue; }
            const f4v p = W[i];
            P[i] = p;
            if (c.cin) adam_store_shadow<2>(c, i, p); else if (c.s) adam_store_shadow<1>(c, i, p);
        }
        i0 = (c.n >> 2) << 2;
    }
    for (int i = i0 + threadIdx.x; i < c.n; i += 256) {
        if (!SCATTER) { w[i] = c.a[i]; continue; }
        const float p = w[i];
        c.a[i] = p;
        if (c.cin) {
            const long gi = c.off + i, r = gi / 5;
            const int j = (int)(gi - r * 5), co = (int)(r / c.cin), ci = (int)(r - (long)co * c.cin);
            const bf16_t pb = f2bf(p);
            c.s[(long)co * 5 * c.cin + (long)j * c.cin + ci] = pb;
            c.s2[(long)ci * 5 * c.cout + (long)(4 - j) * c.cout + co] = pb;
        } else if (c.s) c.s[i] = f2bf(p);
    }
}
int ro_mt_param_wire(const MtChunk* chunks, int nchunks, float* wire, const float* gbase, bool scatter, hipStream_t st) {
    if (nchunks > 0) {
        if (scatter) hipLaunchKernelGGL(k_mt_param_wire<true>, dim3(nchunks), dim3(256), 0, st, chunks, wire, gbase);
        else hipLaunchKernelGGL(k_mt_param_wire<false>, dim3(nchunks), dim3(256), 0, st, chunks, wire, gbase);
    }
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_mt_zero(const MtChunk* chunks, int nchunks, hipStream_t st) {
    if (nchunks > 0) hipLaunchKernelGGL(k_mt_zero, dim3(nchunks), dim3(256), 0, st, chunks);
    B2S_LAUNCH_CHECK(); return 0;
}
int ro_fill(float* p, float v, long n, hipStream_t st) {
    hipLaunchKernelGGL(k_fill, dim3(ew_grid(n)), dim3(256), 0, st, p, v, n);
    B2S_LAUNCH_CHECK(); return 0;
}
