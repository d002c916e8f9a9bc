// Fused (flash-style) attention kernels, see attention.hip.
#pragma once
#include "b2s_common.h"

struct AttnArgs {
    const void *q = nullptr, *k = nullptr, *v = nullptr;   // [B, L, H*dh] rows with leading dimensions ldq / ldk / ldv
    int ldq = 0, ldk = 0, ldv = 0;
    void* out = nullptr;                                   // forward: context [B, Lq, H*dh] (ld ldo)
    int ldo = 0;
    int B = 0, H = 0, Lq = 0, Lk = 0;
    float scale = 1.f;
    int mask_mode = 0;                                     // bit0: keys >= klen[b] masked, bit1: causal
    const int* klen = nullptr;
    DropCfg drop = {0, 0, 1.f};                            // dropout on the weights, element index ((z*Lq + q)*Lk + k)
    float* lse = nullptr;                                  // [B*H, Lq] log-sum-exp of the scaled, masked logits
    // backward
    const void* dout = nullptr;                            // d context (ld ldo)
    const void* oref = nullptr;                            // forward context O (ld ldo), for D = rowsum(dO * O)
    float* dsum = nullptr;                                 // [B*H, Lq] scratch: rowsum(dO * O)
    void *dq = nullptr, *dk = nullptr, *dv = nullptr;
    int lddq = 0, lddk = 0, lddv = 0;
    // guided-attention term (extension, engine.hip: decoder): loss += c * sum_{q < qlen[b]} sum_k P[q][k] * W[q][k] with
    // W = 1 - exp(-(k / klen[b] - q / qlen[b])^2 * ga_inv2s2).  Forward writes ga_rows[z][q] = sum_k P W (0 for padded
    // rows); backward adds c = *ga_scale times dW-term to dP and c * ga_rows to D.  Enabled when ga_rows != nullptr.
    float* ga_rows = nullptr;                              // [B*H, Lq]
    const int* qlen = nullptr;
    const float* ga_scale = nullptr;                       // device scalar (backward only)
    float ga_inv2s2 = 0.f;
    // padded query rows: when set, 64-row query tiles that lie entirely at or beyond qskip[b] are not computed -- their context / dQ rows
    // are written as zeros (lse = dsum = 0) and the dK/dV loop stops before them.  Only for callers whose rows >= qskip[b] are padding
    // that nothing reads and whose upstream gradient is zero (the decoder: its outputs and gradients are masked by target_lengths).
    const int* qskip = nullptr;
    // ragged rows (the decoder's compact layout, engine.hip): when set, utterance b's rows of the q-side tensors (q, out, dout, oref, dq) are
    // [qoff[b], qoff[b + 1]) instead of [b Lq, (b + 1) Lq), likewise koff for the k-side tensors (k, v, dk, dv); Lq / Lk stay the padded lengths
    // (grid size, the [B H, Lq] row indexing of lse / dsum / ga_rows and of the dropout rows).  [B + 1] entries each.
    const int *qoff = nullptr, *koff = nullptr;
};

bool b2s_flash_supported(int dh);
// bf16 kernels on the 32x32x16 MFMA (attention32.hip); which: 0 forward, 1 dQ (needs oref), 2 dK/dV
bool b2s_flash32_supported(int dh);
int b2s_flash32_launch(const AttnArgs& a, int dh, int which, hipStream_t st);
int b2s_flash_fwd(int dtype, const AttnArgs& a, int dh, hipStream_t st);
// st_dkv (optional): the dK / dV kernel is launched there, behind ev_dq recorded on st after the dQ kernel (which writes the row sums it reads)
int b2s_flash_bwd(int dtype, const AttnArgs& a, int dh, const void* O, hipStream_t st, hipStream_t st_dkv = nullptr, hipEvent_t ev_dq = nullptr);
int b2s_flash_align(int dtype, const AttnArgs& a, int dh, float* align, hipStream_t st);
