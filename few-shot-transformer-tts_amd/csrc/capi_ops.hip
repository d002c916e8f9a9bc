// Op-level C ABI: thin wrappers used by the standalone modules (MultiheadAttention, FFNLayer, ...) and by
// the op parity tests.  See include/b2s_hip.h.
#include "engine.h"
#include "attention.h"
#include "enc_fused.h"

namespace {
inline hipStream_t S_(void* s) { return (hipStream_t)s; }
inline int rup8(int x) { return (x + 7) & ~7; }
__global__ void k_dropmask(DropCfg d, uint8_t* out, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = b2s_keep(d, (uint32_t)i) ? 1 : 0;
}
}  // namespace

// defined in engine.hip
int b2s_attn_core_fwd_export(int dtype, hipStream_t st, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv,
                             void* ctx, int ldc, int B, int H, int Lq, int Lk, int dh, int mask_mode, const int* klen,
                             const float* bias, long bias_sb, long bias_sq, DropCfg drop, float* S, void* P, void* Pd);
int b2s_attn_core_bwd_export(int dtype, hipStream_t st, const void* dctx, int ldc, const void* q, int ldq, const void* k, int ldk,
                             const void* v, int ldv, const void* P, const void* Pd, void* dq, int lddq, void* dk, int lddk,
                             void* dv, int lddv, int B, int H, int Lq, int Lk, int dh, DropCfg drop, float* dP, void* dS);

extern "C" int b2s_gemm(const b2s_gemm_desc* d, const void* A, const void* B, void* C, const float* bias,
                        const float* residual, const int32_t* row_len, const int32_t* conv_len, void* stream) {
    B2S_CHECK(d && A && B && C, "null argument");
    GemmArgs g;
    g.M = d->M; g.N = d->N; g.K = d->K; g.batch = d->batch > 0 ? d->batch : 1; g.batch_inner = d->batch_inner > 0 ? d->batch_inner : 1;
    g.A.p = A; g.A.ld = d->lda; g.A.bs_o = d->a_bs_o; g.A.bs_i = d->a_bs_i;
    g.B.p = B; g.B.ld = d->ldb; g.B.bs_o = d->b_bs_o; g.B.bs_i = d->b_bs_i;
    if (d->trans_a) { g.A.R = d->K; g.A.C = d->M; } else { g.A.R = d->M; g.A.C = d->K; }
    if (d->trans_b) { g.B.R = d->K; g.B.C = d->N; } else { g.B.R = d->N; g.B.C = d->K; }
    if (d->conv_cin_a > 0) {
        B2S_CHECK(!d->trans_a, "conv gather is defined on the row-major A operand");
        g.A.g_cin = d->conv_cin_a; g.A.g_T = d->conv_T; g.A.g_len = conv_len; g.A.R = d->M; g.A.C = d->K;
    }
    g.C = C; g.c_fp32 = d->c_fp32; g.ldc = d->ldc; g.cs_o = d->c_bs_o; g.cs_i = d->c_bs_i;
    g.epi.alpha = d->alpha; g.epi.bias = bias; g.epi.relu = d->relu; g.epi.accumulate = d->accumulate;
    g.epi.drop = make_drop(d->drop_p, d->seed, 7u);
    g.epi.residual = residual; g.epi.ldr = d->ldc;
    g.epi.row_len = row_len; g.epi.rows_per_batch = d->rows_per_batch > 0 ? d->rows_per_batch : 1;
    g.epi.conv_dw_cin = d->conv_dw_cin;
    return b2s_gemm_launch(g, d->dtype, d->trans_a != 0, d->trans_b != 0, S_(stream));
}
extern "C" int b2s_gemm_splitk(const b2s_gemm_desc* d, int splitk, const void* A, const void* B, float* C, float* ws, size_t ws_floats,
                               void* stream) {
    B2S_CHECK(d && A && B && C && splitk >= 1, "null argument / bad split");
    B2S_CHECK(d->c_fp32 && d->accumulate && !d->relu && d->drop_p == 0.f && d->conv_cin_a == 0, "split-K needs a linear fp32 accumulate epilogue");
    GemmArgs g;
    g.M = d->M; g.N = d->N; g.K = d->K; g.batch = d->batch > 0 ? d->batch : 1; g.batch_inner = d->batch_inner > 0 ? d->batch_inner : 1;
    g.A.p = A; g.A.ld = d->lda; g.A.bs_o = d->a_bs_o; g.A.bs_i = d->a_bs_i;
    g.B.p = B; g.B.ld = d->ldb; g.B.bs_o = d->b_bs_o; g.B.bs_i = d->b_bs_i;
    if (d->trans_a) { g.A.R = d->K; g.A.C = d->M; } else { g.A.R = d->M; g.A.C = d->K; }
    if (d->trans_b) { g.B.R = d->K; g.B.C = d->N; } else { g.B.R = d->N; g.B.C = d->K; }
    g.C = C; g.c_fp32 = 1; g.ldc = d->ldc; g.cs_o = d->c_bs_o; g.cs_i = d->c_bs_i;
    g.epi.alpha = d->alpha; g.epi.accumulate = 1; g.epi.conv_dw_cin = d->conv_dw_cin;
    g.splitk = splitk; g.ws = ws; g.ws_floats = ws ? ws_floats : 0;
    return b2s_gemm_launch(g, d->dtype, d->trans_a != 0, d->trans_b != 0, S_(stream));
}
extern "C" int b2s_layernorm_forward(int dtype, const float* x, const float* gamma, const float* beta, void* y, float* mean,
                                     float* rstd, int M, int D, float eps, void* stream) {
    B2S_CHECK(x && gamma && beta && y && mean && rstd, "null argument");
    return ro_layernorm_fwd(dtype, x, gamma, beta, y, D, nullptr, 0, mean, rstd, M, D, eps, nullptr, 1, S_(stream));
}
extern "C" int b2s_layernorm_backward(int dtype, const void* dy, const float* x, const float* gamma, const float* mean,
                                      const float* rstd, float* dx, float* dgamma, float* dbeta, int M, int D, void* stream) {
    B2S_CHECK(dy && x && gamma && mean && rstd && dx && dgamma && dbeta, "null argument");
    B2S_HIP(hipMemsetAsync(dgamma, 0, sizeof(float) * D, S_(stream)));
    B2S_HIP(hipMemsetAsync(dbeta, 0, sizeof(float) * D, S_(stream)));
    return ro_layernorm_bwd(dtype, dy, 0, D, x, gamma, mean, rstd, dx, 0, dgamma, dbeta, M, D, nullptr, 1, S_(stream));
}
extern "C" size_t b2s_attention_ws_bytes(int dtype, int B, int H, int Lq, int Lk) {
    const size_t pn = (size_t)B * H * Lq * rup8(Lk);
    return 2 * pn * 4 + pn * (dtype ? 2 : 4) + 1024;      // S / dP (fp32) + dS (T)
}
extern "C" int b2s_attention_forward(int dtype, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* ctx,
                                     int ldc, int B, int H, int Lq, int Lk, int dh, int mask_mode, const int32_t* klen,
                                     const float* bias, int64_t bias_sb, int64_t bias_sq, float drop_p, uint64_t seed, void* ws,
                                     void* P_out, void* Pd_out, void* stream) {
    B2S_CHECK(q && k && v && ctx && ws && P_out, "null argument");
    B2S_CHECK(drop_p <= 0.f || Pd_out, "Pd_out is required when dropout is on");
    return b2s_attn_core_fwd_export(dtype, S_(stream), q, ldq, k, ldk, v, ldv, ctx, ldc, B, H, Lq, Lk, dh, mask_mode, klen, bias,
                                    bias_sb, bias_sq, make_drop(drop_p, seed, 11u), (float*)ws, P_out, Pd_out);
}
extern "C" int b2s_attention_backward(int dtype, const void* dctx, int ldc, const void* q, int ldq, const void* k, int ldk,
                                      const void* v, int ldv, const void* P, const void* Pd, void* dq, int lddq, void* dk, int lddk,
                                      void* dv, int lddv, int B, int H, int Lq, int Lk, int dh, float drop_p, uint64_t seed, void* ws,
                                      void* stream) {
    B2S_CHECK(dctx && q && k && v && P && dq && dk && dv && ws, "null argument");
    const size_t pn = (size_t)B * H * Lq * rup8(Lk);
    float* dP = (float*)ws + pn;
    void* dS = (void*)((float*)ws + 2 * pn);
    return b2s_attn_core_bwd_export(dtype, S_(stream), dctx, ldc, q, ldq, k, ldk, v, ldv, P, Pd, dq, lddq, dk, lddk, dv, lddv, B, H,
                                    Lq, Lk, dh, make_drop(drop_p, seed, 11u), dP, dS);
}
// ---- fused attention (attention.hip)
namespace {
AttnArgs mk_args(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int B, int H, int Lq, int Lk, int dh, int mask_mode,
                 const int32_t* klen, float drop_p, uint64_t seed, float* lse) {
    AttnArgs a;
    a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk;
    a.scale = 1.f / sqrtf((float)dh); a.mask_mode = mask_mode; a.klen = klen; a.drop = make_drop(drop_p, seed, 11u); a.lse = lse;
    return a;
}
}  // namespace
extern "C" int b2s_flash_attention_forward(int dtype, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* ctx, int ldc,
                                           int B, int H, int Lq, int Lk, int dh, int mask_mode, const int32_t* klen, float drop_p,
                                           uint64_t seed, float* lse_out, void* stream) {
    B2S_CHECK(ctx && lse_out, "null argument");
    AttnArgs a = mk_args(q, ldq, k, ldk, v, ldv, B, H, Lq, Lk, dh, mask_mode, klen, drop_p, seed, lse_out);
    a.out = ctx; a.ldo = ldc;
    return b2s_flash_fwd(dtype, a, dh, S_(stream));
}
extern "C" int b2s_flash_attention_backward(int dtype, const void* dctx, const void* ctx, int ldc, const void* q, int ldq, const void* k, int ldk,
                                            const void* v, int ldv, const float* lse, float* dsum_scratch, void* dq, int lddq, void* dk,
                                            int lddk, void* dv, int lddv, int B, int H, int Lq, int Lk, int dh, int mask_mode,
                                            const int32_t* klen, float drop_p, uint64_t seed, void* stream) {
    B2S_CHECK(dctx && ctx && lse && dsum_scratch, "null argument");
    AttnArgs a = mk_args(q, ldq, k, ldk, v, ldv, B, H, Lq, Lk, dh, mask_mode, klen, drop_p, seed, const_cast<float*>(lse));
    a.dout = dctx; a.ldo = ldc; a.dsum = dsum_scratch; a.dq = dq; a.dk = dk; a.dv = dv; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
    return b2s_flash_bwd(dtype, a, dh, ctx, S_(stream));
}
extern "C" int b2s_flash_attention_align(int dtype, const void* q, int ldq, const void* k, int ldk, const float* lse, int B, int H, int Lq, int Lk,
                                         int dh, int mask_mode, const int32_t* klen, float* align_out, void* stream) {
    AttnArgs a = mk_args(q, ldq, k, ldk, nullptr, 0, B, H, Lq, Lk, dh, mask_mode, klen, 0.f, 0, const_cast<float*>(lse));
    return b2s_flash_align(dtype, a, dh, align_out, S_(stream));
}
extern "C" int b2s_align_from_probs(int dtype, const void* P, float* align, int B, int H, int Lq, int Lk, void* stream) {
    B2S_CHECK(P && align, "null argument");
    return ro_align_transpose(dtype, P, align, B * H, Lq, Lk, rup8(Lk), S_(stream));
}
extern "C" int b2s_add3(const float* a, const float* b, const float* c, float* out, int64_t n, void* stream) {
    B2S_CHECK(a && b && out, "null argument");
    return c ? ro_add3(a, b, c, out, n, S_(stream)) : ro_add(a, b, out, n, S_(stream));
}
extern "C" int b2s_cast(int dtype, const float* in, void* out, int64_t n, void* stream) {
    B2S_CHECK(in && out, "null argument");
    return ro_cast(dtype, in, out, n, S_(stream));
}
extern "C" int b2s_cast_back(int dtype, const void* in, float* out, int64_t n, void* stream) {
    B2S_CHECK(in && out, "null argument");
    return ro_cast_back(dtype, in, out, n, S_(stream));
}
extern "C" int b2s_dropout_mask(float p, uint64_t seed, uint32_t op_id, uint8_t* out, int64_t n, void* stream) {
    B2S_CHECK(out && n >= 0, "bad argument");
    DropCfg d = make_drop(p, seed, op_id);
    if (p <= 0.f) { B2S_HIP(hipMemsetAsync(out, 1, n, S_(stream))); return 0; }
    hipLaunchKernelGGL(k_dropmask, dim3(cdiv(n, 256)), dim3(256), 0, S_(stream), d, out, (long)n);
    B2S_LAUNCH_CHECK();
    return 0;
}

__global__ void k_dropmask_attn(DropCfg d, uint8_t* out, long rows, int Lk) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * Lk) out[i] = b2s_keep_w(d, (uint32_t)(i / Lk), (uint32_t)(i % Lk)) ? 1 : 0;
}
extern "C" int b2s_dropout_mask_attn(float p, uint64_t seed, uint32_t op_id, uint8_t* out, int64_t rows, int Lk, void* stream) {
    B2S_CHECK(out && rows >= 0 && Lk > 0, "bad argument");
    DropCfg d = make_drop(p, seed, op_id);
    if (p <= 0.f) { B2S_HIP(hipMemsetAsync(out, 1, rows * Lk, S_(stream))); return 0; }
    hipLaunchKernelGGL(k_dropmask_attn, dim3(cdiv(rows * Lk, 256)), dim3(256), 0, S_(stream), d, out, (long)rows, Lk);
    B2S_LAUNCH_CHECK();
    return 0;
}

// ---- fused encoder sublayer kernels (enc_fused.h), op level: what the engine launches per sublayer, for the parity tests
extern "C" int b2s_encf_attention_forward(const void* hN, const void* Wqkv, const void* Wo, const int32_t* klen, int B, int S, float drop_p,
                                          uint64_t seed, uint32_t op_id, void* qkv, void* ctx, float* lse, void* slabs, int slab_bf16, void* stream) {
    EncfAttnFwd a;
    a.hN = (const bf16_t*)hN; a.Wqkv = (const bf16_t*)Wqkv; a.Wo = (const bf16_t*)Wo; a.klen = klen; a.B = B; a.S = S;
    a.datt = make_drop(drop_p, seed, op_id); a.qkv = (bf16_t*)qkv; a.ctx = (bf16_t*)ctx; a.lse = lse; a.slabs = slabs;
    return b2s_encf_attn_fwd(a, slab_bf16, S_(stream));
}
extern "C" int b2s_encf_attention_backward(const void* dY, const void* qkv, const void* ctx, const float* lse, const void* WoT, const void* WqkvT,
                                           const int32_t* klen, int B, int S, float drop_p, uint64_t seed, uint32_t op_id, void* dqkv, void* slabs,
                                           int slab_bf16, void* stream) {
    EncfAttnBwd a;
    a.dY = (const bf16_t*)dY; a.qkv = (const bf16_t*)qkv; a.ctx = (const bf16_t*)ctx; a.lse = lse; a.WoT = (const bf16_t*)WoT;
    a.WqkvT = (const bf16_t*)WqkvT; a.klen = klen; a.B = B; a.S = S; a.datt = make_drop(drop_p, seed, op_id); a.dqkv = (bf16_t*)dqkv; a.slabs = slabs;
    return b2s_encf_attn_bwd(a, slab_bf16, S_(stream));
}
// backward = 0: X = LN(x), Wa = W1, Wb = W2, f_io receives f;  backward = 1: X = dY, Wa = W2^T, Wb = W1^T, f_io holds the saved f, dz receives dz
extern "C" int b2s_encf_ffn_sublayer(int backward, const void* X, const void* Wa, const void* Wb, void* f_io, void* dz, int B, int S, float drop_p,
                                     uint64_t seed, uint32_t op_id, void* slabs, int slab_bf16, void* stream) {
    EncfFfn a;
    a.X = (const bf16_t*)X; a.Wa = (const bf16_t*)Wa; a.Wb = (const bf16_t*)Wb; a.F = (bf16_t*)f_io; a.dz = (bf16_t*)dz; a.slabs = slabs; a.B = B; a.S = S;
    a.dhid = backward ? DropCfg{0, 0, 1.f} : make_drop(drop_p, seed, op_id);
    a.aux_scale = make_drop(drop_p, seed, op_id).scale;
    return b2s_encf_ffn(a, backward != 0, slab_bf16, S_(stream));
}
extern "C" int b2s_encf_reduce_layernorm_forward(const float* x_in, const void* slabs, int ns, int slab_bf16, float drop_p, uint64_t seed, uint32_t op_id,
                                                 const float* gamma, const float* beta, float* x_out, void* h, float* h32, int ldh32, float* mean,
                                                 float* rstd, int M, void* stream) {
    return b2s_encf_reduce_ln_fwd(x_in, slabs, ns, slab_bf16, make_drop(drop_p, seed, op_id), gamma, beta, x_out, (bf16_t*)h, encf::D, h32, ldh32, mean, rstd,
                                  M, S_(stream));
}
extern "C" int b2s_encf_reduce_layernorm_backward(const void* slabs, int ns, int slab_bf16, const float* x_in, const float* gamma, const float* mean,
                                                  const float* rstd, float* dx, float* dgamma, float* dbeta, float* ws, void* dy2, float drop_p,
                                                  uint64_t seed, uint32_t op_id, int M, void* stream) {
    B2S_CHECK(dgamma && dbeta, "null argument");
    LnReduceBatch bt; bt.n = 1;
    bt.j[0].ws = ws; bt.j[0].D = encf::D; bt.j[0].dgamma = dgamma; bt.j[0].dbeta = dbeta;
    B2S_TRY(b2s_encf_reduce_ln_bwd(slabs, ns, slab_bf16, x_in, gamma, mean, rstd, dx, ws, &bt.j[0].nblk, (bf16_t*)dy2, make_drop(drop_p, seed, op_id), M,
                                   S_(stream)));
    return ro_ln_param_reduce_batch(bt, S_(stream));
}
extern "C" int b2s_transpose_bf16(const void* src, void* dst, int R, int C, void* stream) {
    const EncfTransposeJob jb = {(const bf16_t*)src, (bf16_t*)dst, R, C};
    return b2s_encf_transpose(&jb, 1, S_(stream));
}
