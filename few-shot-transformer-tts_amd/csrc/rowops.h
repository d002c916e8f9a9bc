// Row-wise / element-wise HBM-bound kernels of the Transformer-TTS path (see rowops.hip).
// dtype: 0 = fp32 compute tensors, 1 = bf16 compute tensors ("T" below).  Residual stream, statistics,
// parameters and gradients of parameters are always fp32.
#pragma once
#include "b2s_common.h"

// x[b,s,:] = embed[id]*(s<len[b]) + pe[s,:]*pe_scale ; dropout            (modules.py:49-56, tacotron.py:34)
int ro_embed_prep_fwd(const long* ids, const int* lens, const float* embed, const float* pe, const float* pe_scale,
                      float* x, int B, int S, int D, DropCfg drop, hipStream_t st);
int ro_embed_prep_bwd(const float* dx, const long* ids, const int* lens, const float* pe, float* d_embed,
                      float* d_pe_scale, int B, int S, int D, DropCfg drop, hipStream_t st, int dx_bf16 = 0);      // dx_bf16: dx holds bf16

// y = LN(x) (eps), rows of width D; y (T) with leading dim ldy; optional fp32 copy y32 (ld ldy32);
// optional row mask (rows t >= row_len[b] -> 0)                                    (modules.py:36-47,88-106)
int ro_layernorm_fwd(int dtype, const float* x, const float* gamma, const float* beta, void* y, int ldy, float* y32,
                     int ldy32, float* mean, float* rstd, int M, int D, float eps, const int* row_len,
                     int rows_per_batch, hipStream_t st);
// dx (+)= LN'(dy); dgamma += ..., dbeta += ... (atomic; caller zeroes).  dy is T (dy_fp32=0) or fp32, ld lddy.
int ro_layernorm_bwd(int dtype, const void* dy, int dy_fp32, int lddy, const float* x, const float* gamma,
                     const float* mean, const float* rstd, float* dx, int accumulate, float* dgamma, float* dbeta,
                     int M, int D, const int* row_len, int rows_per_batch, hipStream_t st, float* ws = nullptr,
                     void* dy2 = nullptr, DropCfg drop2 = DropCfg{0, 0, 1.f}, int* defer_nblk = nullptr, int dx_bf16 = 0);
// dx_bf16: the residual gradient dx is held in bf16 (D = 512 / 768 only)
// ws (optional): RO_LN_WS_ROWS * 2 * D floats of scratch for the parameter-gradient partials (avoids global atomics)
// defer_nblk (with ws): the partial-sum reduction into dgamma / dbeta is NOT launched; *defer_nblk receives the number of
// partial rows and the caller reduces several LayerNorms' partials in one launch (ro_ln_param_reduce_batch)
constexpr int RO_LN_WS_ROWS = 768;
struct LnReduceJob { const float* ws; int nblk, D; float* dgamma; float* dbeta; };
constexpr int RO_LN_BATCH = 8;
struct LnReduceBatch { int n; LnReduceJob j[RO_LN_BATCH]; };
int ro_ln_param_reduce_batch(const LnReduceBatch& b, hipStream_t st);

// P = softmax(scale*S + mask) per row; rows [Z=B*H][Lq][ldp].  mask_mode bit0: keys >= klen[b] masked,
// bit1: causal (key > query masked); bias: optional dense additive fp32 bias [bias_sb*b + bias_sq*q + k].
// Writes P (T) and, if drop.thresh, Pd = dropout(P) (T); pad columns [Lk, ldp) are written as zeros.
int ro_softmax_fwd(int dtype, const float* S, void* P, void* Pd, int B, int H, int Lq, int Lk, int ldp, float scale,
                   int mask_mode, const int* klen, const float* bias, long bias_sb, long bias_sq, DropCfg drop,
                   hipStream_t st);
// dS = scale * P * (dPeff - sum_k P*dPeff), dPeff = dropmask(dPraw)                     (autograd of attention.py:83-91)
int ro_softmax_bwd(int dtype, const void* P, const float* dPraw, void* dS, int B, int H, int Lq, int Lk, int ldp,
                   float scale, DropCfg drop, hipStream_t st);
// align[z][k][q] = P[z][q][k] (fp32)                                                   (attention.py:88)
int ro_align_transpose(int dtype, const void* P, float* align, int Z, int Lq, int Lk, int ldp, hipStream_t st);

// out(T)[i] = dropout(in[i]) ; n elements, 2-D index (row*ncols + col) with input ld
int ro_cast_drop(int dtype, const float* in, int ldin, void* out, int ldo, int M, int N, DropCfg drop, hipStream_t st);
// fp32 -> T cast of a flat array (weight shadows)
int ro_cast(int dtype, const float* in, void* out, long n, hipStream_t st);
int ro_cast_back(int dtype, const void* in, float* out, long n, hipStream_t st);

// decoder input prep: x[b,t,:] = (t>0 && t-1<len[b] ? a[b,t-1,:] : 0) + pe[t,:]*pe_scale ; dropout  (modules.py:108-121)
// (off != nullptr: ragged decoder rows, `rows` of them -- see rowops.hip: "ragged (compact) decoder rows")
int ro_shift_pe_fwd(const float* a, const int* lens, const float* pe, const float* pe_scale, float* x, int B, int T,
                    int D, DropCfg drop, hipStream_t st, const int* off = nullptr, int rows = 0);
int ro_shift_pe_bwd(int dtype, const float* dx, const int* lens, const float* pe, void* da, float* d_pe_scale, int B,
                    int T, int D, DropCfg drop, hipStream_t st, int dx_bf16 = 0, const int* off = nullptr, int rows = 0);
int ro_set_rowoff(const int* off_host, int n, int* dst, hipStream_t st);                       // n = B + 1 <= 65 offsets through the kernel arguments
int ro_rows_gather(int out_dtype, const float* in, void* out, const int* off, int B, int T, int C, hipStream_t st, const float* in2 = nullptr, float* out2 = nullptr, int C2 = 0);     // padded [B, T, C] fp32 -> ragged rows
int ro_heads_scatter(int dtype, const void* x, int ldx, const float* w, const float* bias, const float* mel_in, float* mel_out, float* stop_out,
                     const int* off, int B, int T, int C, int D, hipStream_t st);

// speaker / language embeddings (tacotron.py:21-31), written into memory[:, :, col0 : col0+E] for all S
int ro_spk_embed_fwd(const long* spk_ids, const float* table, const float* W, const float* b, float* e_raw,
                     float* h_pre, float* mem32, void* memT, int dtype, int ldm, int col0, int B, int S, int E,
                     hipStream_t st);
int ro_lang_embed_fwd(const float* vecs, int L, const float* Wl, const float* W, const float* b, float* e_raw,
                      float* h_pre, float* mem32, void* memT, int dtype, int ldm, int col0, int B, int S, int E,
                      hipStream_t st);
// dmem32 [B*S, ldm] -> gradients of the tiny embedding nets (atomic adds; caller zeroes)
int ro_spk_embed_bwd(const float* dmem, int ldm, int col0, const long* spk_ids, const float* e_raw, const float* h_pre,
                     const float* W, float* d_table, float* dW, float* db, float* dh_scratch, int B, int S, int E, hipStream_t st);
int ro_lang_embed_bwd(const float* dmem, int ldm, int col0, const float* vecs, int L, const float* e_raw,
                      const float* h_pre, const float* Wl, const float* W, float* dWl, float* dW, float* db, float* dh_scratch,
                      int B, int S, int E, hipStream_t st);

// out[m] = (t<len ? x[m,:].w + b : 0)     (stop_net on the detached decoder output, tacotron.py:114-115)
int ro_rowdot_fwd(int dtype, const void* x, int ldx, const float* w, const float* b, float* out, int M, int D,
                  const int* row_len, int rows_per_batch, hipStream_t st);
// generic column reduction: out[c] (+)= sum_m wgt[m] * X[m,c]   (X is T or fp32)
int ro_colsum(int dtype, const void* X, int x_fp32, int ldx, const float* wgt, float* out, int accumulate, int M,
              int C, hipStream_t st);

// BatchNorm1d over all M = B*T rows (tacotron.py:83-89)
int ro_bn_stats(const float* y, int M, int C, float* mean, float* rstd, float eps, float* running_mean,
                float* running_var, long* num_batches_tracked, float momentum, float* scratch, hipStream_t st);
int ro_bn_apply_train(int dtype, const float* y, const float* sums, float* mean, float* rstd, float eps, float* running_mean, float* running_var,
                      long* num_batches_tracked, float momentum, const float* gamma, const float* beta, int use_tanh, void* outT, float* out32,
                      const float* add32, int M, int C, DropCfg drop, hipStream_t st);
int ro_bn_eval_stats(const float* running_mean, const float* running_var, float* mean, float* rstd, float eps, int C,
                     hipStream_t st);
// u = dropout(act(gamma*(y-mean)*rstd+beta)); act = tanh if use_tanh.  out T (ldo) or, if out32, fp32 = add32 + u
int ro_bn_apply(int dtype, const float* y, const float* mean, const float* rstd, const float* gamma,
                const float* beta, int use_tanh, void* outT, float* out32, const float* add32, int M, int C,
                DropCfg drop, hipStream_t st);
// backward: dout (T or fp32) -> dgamma, dbeta (atomic, caller zeroes) then dy (T)
int ro_bn_bwd(int dtype, const void* dout, int dout_fp32, const float* y, const float* mean, const float* rstd,
              const float* gamma, const float* beta, int use_tanh, float* dgamma, float* dbeta, void* dyT, int M,
              int C, DropCfg drop, hipStream_t st);

// losses (tacotron.py:136-158).  out[0..6] = loss,bef,aft,mse,l2(unchanged),stop,sumlen ; aft_losses[B]
// grads of (w0*bef_loss + w1*aft_loss + w2*stop_loss): d_bef[m,c], d_aft[m,c], d_stop[m]; gscale = device float[3] (null -> 1,1,1)
int ro_loss_fwd(const float* bef, const float* aft, const float* stop, const float* tgt, const int* lens,
                const float* l2, float* out, float* aft_losses, int B, int T, int C, float pos_weight, float* scratch,
                hipStream_t st, bool scratch_zeroed = false);       // scratch: 3 + B floats, zeroed here unless the caller did
int ro_loss_bwd(const float* bef, const float* aft, const float* stop, const float* tgt, const int* lens,
                const float* gscale, float* d_bef, float* d_aft, float* d_stop, int B, int T, int C, float pos_weight,
                hipStream_t st);

// multi-tensor ops over a chunk table (device array of MtChunk)
struct MtChunk {            // pad: 1 = member of the L2 set; s: bf16 shadow (or null)
    float* a; float* b; float* c; float* d; bf16_t* s; int n; int pad;
    // conv weights (cin > 0): the shadows are the two re-laid-out bf16 images of the whole tensor (see ro_conv_w_relayout),
    // s = forward image, s2 = backward-data image; off = index of the chunk's first element in the [Cout][Cin][5] master
    bf16_t* s2; int cin, cout; long off;
};
int ro_mt_sumsq(const MtChunk* chunks, int nchunks, float* out, float scale, hipStream_t st);
int ro_mt_param_wire(const MtChunk* chunks, int nchunks, float* wire, const float* gbase, bool scatter, hipStream_t st);      // sharded optimizer: pack / scatter
int ro_mt_cast(const MtChunk* chunks, int nchunks, hipStream_t st);                                        // s[0..n) = bf16(a[0..n)) for every chunk
int ro_mt_zero(const MtChunk* chunks, int nchunks, hipStream_t st);                                        // a[0..n) = 0 for every chunk         // out += scale*sum a^2
int ro_mt_axpy(const MtChunk* chunks, int nchunks, float alpha, const float* gscale, hipStream_t st);  // b += alpha*gscale*a
// Adam: a=param b=grad c=m d=v ; lr and step read from device (hp[0]=lr, hp[1]=bias_corr1, hp[2]=bias_corr2)
// wire (optional): bf16 array laid out like the fp32 gradient buffer that starts at gbase -- the gradients are read from there instead
// learning rate and the step's bias corrections (bc1 = 1 - beta1^t, sbc2 = sqrt(1 - beta2^t)): kernel arguments -- a device copy cost a 5 us copy kernel in
// front of every optimizer launch, the encoder group's on the critical path at the end of the step
struct AdamHyper { float lr, bc1, sbc2; };
int ro_mt_adam(const MtChunk* chunks, int nchunks, AdamHyper hp, float beta1, float beta2, float eps, float l2,
               float grad_scale, float* sumsq_part, hipStream_t st, const void* wire = nullptr, const float* gbase = nullptr,
               int max_wg = 0);                                                             // max_wg > 0: at most that many workgroups walk the chunk list
// out[m][j*cin + ci] = x[m + j - 2][ci] if 0 <= t + j - 2 < min(lens[b], T) (m = b*T + t) else 0: the conv1d k=5 p=2 input of every token
// with its five taps side by side (bf16 only; cin % 8 == 0)
int ro_im2col5(int dtype, const void* x, const int* lens, int T, int cin, void* out, long M, hipStream_t st);
int ro_sum_scaled(const float* part, int n, float scale, float* out, hipStream_t st, float* zero = nullptr, int nzero = 0);   // (+ clears zero[0..nzero))

// conv weight relayouts (fp32 master [Cout][Cin][5] -> T):  fwd[co][j*Cin+ci] ; bwd[ci][j*Cout+co] = w[co][ci][4-j]
int ro_conv_w_relayout(int dtype, const float* w, void* wf, void* wb, int Cout, int Cin, hipStream_t st);
int ro_add(const float* a, const float* b, float* out, long n, hipStream_t st);
int ro_add3(const float* a, const float* b, const float* c, float* out, long n, hipStream_t st);     // out = (a + b) + c
int ro_fill(float* p, float v, long n, hipStream_t st);
