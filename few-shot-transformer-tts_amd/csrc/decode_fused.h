// Fused per-sublayer kernels of the autoregressive decode step (decode_fused.hip); launched by decode.hip.
#pragma once
#include "b2s_common.h"

// One decoder sublayer for one frame as ONE kernel:  x = X_in + sum of the previous sublayer's partial slabs; h = LayerNorm(x);
// slice s of the sublayer's projections (a head of the attention / a 1/NS part of the FFN hidden units) for a group of
// utterances; the slice's share of the output projection goes to partial slab s of P_out (dropout mask already applied:
// the mask depends on (utterance, column, frame) only, so it distributes over the sum of the slices).
struct DfCommon {
    const float* X_in;        // [B][D] residual stream before the previous sublayer's contribution
    float* X_out;             // [B][D] = X_in + sum(P_prev): written by the slice-0 workgroups for the next kernel
    const void* P_prev;       // [np_prev][B][D] partial outputs of the previous sublayer in the compute dtype (np_prev = 0: none)
    int np_prev;
    void* P_out;              // [ns][B][D] compute dtype
    int B, D;
    const float *ln_g, *ln_b; // LayerNorm of this sublayer
    float eps;
    const int* t;             // device: frame index
    DropCfg drop_res;         // dropout on the sublayer output (keyed by the frame index in the kernel)
    // 1: the whole decoder has the reference's default widths (b2s_df_fast_model), so the default-size kernel instantiations apply: they
    // hard-code how many partial slabs the previous sublayer wrote (heads = 8, FFN slices = 32) and, in bf16, read fragment-packed weights
    int fast = 0;
};
struct DfAttn {
    DfCommon c;
    int H, dh;
    const void* Wqkv;         // self: qkv_transform [3D][D]; cross: q_transform [D][D]   (the packed copies when b2s_df_attn_packed)
    const void* Wo;           // output_transform [D][D]
    void *Kc, *Vc;            // self: head-major caches [B][H][maxT][dh] (this frame's row is appended); cross: memory K / V rows
    int ldkv;
    long kv_bstride, kv_hstride;
    int maxT;                 // cache rows per (utterance, head) (self)
    float* probs;             // alignment rows [B][H][probs_rows][probs_ld] (optional)
    int probs_rows, probs_ld;
    const int* klen;          // cross: valid memory positions per utterance
    int nmax;                 // most keys a row can have (maxT / S): sizes the LDS score buffer
    float scale;
    DropCfg drop_attn;
};
struct DfFfn {
    DfCommon c;
    int F, ns;                // hidden width (4 D) and number of slices
    const void *W1, *W2;      // input_layer [F][D], output_layer [D][F]   (packed copies when b2s_df_ffn_packed)
    DropCfg drop_hid;
};
struct DfPrenet {
    const float* mels;        // [B][maxT][NM] generated so far
    int maxT, NM, HP, D, B;
    const void *W0, *W1, *Wf; // dense0 [HP][NM], dense1 [HP][HP], dense_final [D][HP] (compute dtype; W1, Wf packed when b2s_df_prenet_packed)
    const float *b0, *b1;
    const float *pe, *pe_scale;
    const int* lengths;
    const int* t;
    float* X;                 // [B][D]
    DropCfg drop0, drop1, drop_x;
};
struct DfFinal {
    const float* X_in; const void* P_prev; int np_prev;
    int fast = 0;             // see DfCommon::fast
    int B, D, NM, maxT;
    const float *ln_g, *ln_b; float eps;
    const void* Wmel;         // mel_net [NM][D] (compute dtype; packed when b2s_df_final_packed)
    const float *wstop, *bstop;
    float* mels;              // [B][maxT][NM]
    int *t, *finished, *lengths, *status, *done_cnt;
};

// dtype 0 = fp32, 1 = bf16.  lds_bytes: dynamic shared memory of the launch (b2s_df_*_lds).
size_t b2s_df_attn_lds(int dtype, int D, int dh, int nmax);
size_t b2s_df_ffn_lds(int dtype, int D, int F, int ns);
int b2s_df_prenet(int dtype, const DfPrenet& a, hipStream_t st);
int b2s_df_attn(int dtype, bool self, const DfAttn& a, hipStream_t st);
int b2s_df_ffn(int dtype, const DfFfn& a, hipStream_t st);
int b2s_df_final(int dtype, const DfFinal& a, hipStream_t st);
bool b2s_df_supported(int dtype, int D, int H, int F, int NM, int HP, int nmax);
// bf16 at the default widths: the kernels take their projection weights as MFMA-fragment-packed copies -- for every block of 16 rows
// and every 32-deep k step the 64 lanes' 16-byte fragments are contiguous, so each load instruction of a wave is one coalesced KB.
// b2s_df_pack writes such a copy of a row-major [N][K] bf16 matrix (N % 16 == 0, K % 32 == 0; same size).
// The default-size attention / FFN / heads kernels assume the WHOLE decoder stack has the default widths: each of them sums a
// compile-time number of partial slabs written by the sublayer before it (8 heads, 32 FFN slices).  One predicate for all three.
bool b2s_df_fast_model(int D, int H, int F);
bool b2s_df_attn_packed(int dtype, int D, int H, int F);
bool b2s_df_ffn_packed(int dtype, int D, int H, int F);
bool b2s_df_prenet_packed(int dtype, int HP, int NM, int D);
bool b2s_df_final_packed(int dtype, int D, int H, int F);
int b2s_df_pack(const void* W, int N, int K, void* out, hipStream_t st);
int b2s_df_ffn_slices(int dtype, int D, int H, int F);
