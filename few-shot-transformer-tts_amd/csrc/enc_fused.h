// Cluster-fused encoder sublayers (enc_fused.hip): one kernel per Transformer-encoder sublayer instead of 3-4 GEMM / attention launches.
// Reference: transformer/modules.py:49-69 (TransformerEncoder.forward), transformer/attention.py:53-122, modules.py:8-20 (FFNLayer).
//
// The encoder is 5 % of the training step's FLOPs in ~106 latency-bound launches (1596 rows: 52-208 GEMM tiles on 256 CUs).  One utterance
// is <= 128 rows x 512, and every encoder op is row-wise or per-utterance, so a sublayer is cut by (utterance, head) or (utterance, hidden
// slice): a workgroup streams its slice of the weights through LDS once (LDS-DMA ring), keeps the utterance's rows on chip across the
// chained products and leaves a PARTIAL [rows, 512] slab; a row kernel sums the slabs in fixed order (deterministic), adds the residual
// and applies the next LayerNorm (forward) / the LayerNorm backward (backward).
//
//   forward  attention sublayer: q/k/v projection of the head (K = 512) -> 128 x 128 attention on chip -> head's columns of the output projection
//            FFN sublayer:       ReLU(h W1[slice]^T) (dropout) -> . W2[:, slice]^T   (two hidden slices of 128 per workgroup)
//   backward the same two shapes with the transposed weight copies (engine: enc_wT), so every streamed operand is K-contiguous:
//            FFN:       dz = (dy W2[:, slice]) * relu' -> dh partial = dz W1[slice]
//            attention: d ctx = dy Wo[:, head] -> attention backward on chip -> dh partial = [dq dk dv] Wqkv[head rows]
//   weight gradients stay on the grouped TN GEMM (saved activations + the dz / dqkv these kernels write).
//
// Compile-time model dims (the default encoder): D = 512, 8 heads of 64, FFN 2048; bf16 compute mode; S <= 128 rows per utterance.
// Everything else takes the unfused path (engine.hip).
#pragma once
#include "b2s_common.h"

namespace encf {
constexpr int D = 512, NH = 8, DH = 64, FF = 2048, HS = 128, NSF = 8, MAXS = 128;      // NSF: FFN slabs (each workgroup sums 2 hidden slices of 128)
}

// slabs: [ns][M][512] partial sublayer outputs (fp32, or bf16 when slab_bf16); M = B * S token rows
struct EncfAttnFwd {
    const bf16_t* hN;        // [M,512] LayerNorm output
    const bf16_t* Wqkv;      // [1536,512]
    const bf16_t* Wo;        // [512,512]
    const int* klen;         // [B] valid keys per utterance
    int B, S;
    DropCfg datt;            // dropout on the attention weights (index ((b*8+h)*S + q)*S + k, as attention.hip)
    bf16_t* qkv;             // out [M,1536]
    bf16_t* ctx;             // out [M,512]
    float* lse;              // out [B*8, S]
    void* slabs;             // out [8][M][512]
};
struct EncfAttnBwd {
    const bf16_t* dY;        // [M,512] bf16(dropout mask * d x_out)
    const bf16_t* qkv;       // [M,1536] saved
    const bf16_t* ctx;       // [M,512] saved
    const float* lse;        // [B*8, S] saved
    const bf16_t* WoT;       // [512,512]  = Wo^T   (rows: input feature = head*64 + d)
    const bf16_t* WqkvT;     // [512,1536] = Wqkv^T
    const int* klen;
    int B, S;
    DropCfg datt;
    bf16_t* dqkv;            // out [M,1536]
    void* slabs;             // out [8][M][512]: partial d h
};
struct EncfFfn {
    const bf16_t* X;         // [M,512]: forward h = LN(x); backward dY
    const bf16_t* Wa;        // [2048,512]: forward W1 (input_layer.weight); backward W2^T
    const bf16_t* Wb;        // [512,2048]: forward W2 (output_layer.weight); backward W1^T
    bf16_t* F;               // [M,2048]: forward OUT f = dropout(relu(.)); backward IN (the saved f: relu / dropout mask)
    bf16_t* dz;              // backward OUT [M,2048]
    void* slabs;             // out [8][M][512]
    int B, S;
    DropCfg dhid;            // forward: hidden dropout (index row*2048 + col, as the GEMM epilogue)
    float aux_scale;         // backward: 1 / (1 - p) of the hidden dropout
};

bool b2s_encf_supported(int D, int H, int F, int S);
int b2s_encf_attn_fwd(const EncfAttnFwd& a, int slab_bf16, hipStream_t st);
int b2s_encf_attn_bwd(const EncfAttnBwd& a, int slab_bf16, hipStream_t st);
int b2s_encf_ffn(const EncfFfn& a, bool bwd, int slab_bf16, hipStream_t st);
// x_out = x_in + dropout(sum_s slabs[s]) ; (mean, rstd) of x_out ; h = LN(x_out) (bf16, ld ldh) and / or h32 (fp32, ld ldh32)
int b2s_encf_reduce_ln_fwd(const float* x_in, const void* slabs, int ns, int slab_bf16, DropCfg dres, const float* gamma, const float* beta,
                           float* x_out, bf16_t* h, int ldh, float* h32, int ldh32, float* mean, float* rstd, int M, hipStream_t st);
// dx += LN'(sum_s slabs[s]) ; partial d gamma / d beta rows -> ws [nblk][2*512] (reduced later: ro_ln_param_reduce_batch) ;
// dy2 (optional) = bf16(dropout(dx)) for the sublayer that runs next in the backward pass
int b2s_encf_reduce_ln_bwd(const void* slabs, int ns, int slab_bf16, const float* x_in, const float* gamma, const float* mean, const float* rstd,
                           float* dx, float* ws, int* nblk, bf16_t* dy2, DropCfg drop2, int M, hipStream_t st, int dx_bf16 = 0);     // dx_bf16: dx holds bf16
// dst[c][r] = src[r][c] for n matrices in one launch (bf16)
struct EncfTransposeJob { const bf16_t* src; bf16_t* dst; int R, C; };
int b2s_encf_transpose(const EncfTransposeJob* jobs, int n, hipStream_t st);
