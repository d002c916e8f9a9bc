// Internal model / context structures of libb2s_hip (see engine.hip).
#pragma once
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <functional>
#include <vector>
#include "../../include/b2s_hip.h"
#include "b2s_common.h"
#include "gemm.h"
#include "rowops.h"

struct TensorInfo {
    std::string name;
    std::vector<int64_t> shape;
    int kind;      // 1 parameter, 0 fp32 buffer, 2 int64 buffer
    long numel;
    bool gemm_weight;   // gets a compute-dtype shadow
    bool l2;            // member of the L2 set (tacotron.py:144-146)
};

struct Arena {
    char* base = nullptr;
    size_t cap = 0, off = 0;
    bool overflow = false;
    void* take(size_t bytes) {
        size_t o = (off + 255) & ~(size_t)255;
        off = o + bytes;
        if (!base) return nullptr;                 // dry run: size only
        if (off > cap) { overflow = true; return base; }
        return base + o;
    }
    float* f32(long n) { return (float*)take((size_t)n * 4); }
    void* T(long n, int esz) { return take((size_t)n * esz); }
};

struct AttnSave {            // one attention sub-layer
    float* x_in = nullptr;   // residual input (fp32) = LayerNorm input
    float *mean = nullptr, *rstd = nullptr;
    void* h = nullptr;       // LN output (T)
    void* qkv = nullptr;     // self: [M,3D] ; cross: q [M,D]
    void* kv = nullptr;      // cross: [B*S,2D] (row stride ldkv: 2D, or L*2D when the layers' K/V live in one projection output)
    int ldkv = 0;
    void *P = nullptr, *Pd = nullptr;   // softmax weights (T) [B,H,Lq,ldp] ; Pd == P when no dropout
    void* ctx = nullptr;     // [M,D] (T)
    float* lse = nullptr;    // fused path: log-sum-exp [B*H, Lq]
    int mask_mode = 0;
    int Lq = 0, Lk = 0, ldp = 0;
    uint32_t op_attn = 0, op_res = 0;
    const int* qskip = nullptr;   // fused path: per-utterance row count beyond which whole query tiles were skipped (attention.h), or null
    const int *qoff = nullptr, *koff = nullptr;     // ragged rows of the q-side / k-side tensors (attention.h), or null: the padded layout
};
struct FfnSave {
    float* x_in = nullptr;
    float *mean = nullptr, *rstd = nullptr;
    void* h = nullptr;
    void* f = nullptr;       // [M,4D] post relu+dropout (T)
    uint32_t op_hid = 0, op_res = 0;
};

struct b2s_ctx {
    int kind = 0;            // 1 encoder, 2 decoder, 3 postnet
    int B = 0, S = 0, T = 0;
    int train = 0;
    uint64_t seed = 0;
    char* ws = nullptr;
    size_t ws_bytes = 0;
    Arena scratch;           // region after the saved activations (reused by backward)
    // inputs (caller keeps them alive until backward)
    const int64_t* ids = nullptr;
    const int32_t *in_len = nullptr, *tgt_len = nullptr;
    const int64_t* spk_ids = nullptr;
    const float* lang_vecs = nullptr;
    // encoder
    std::vector<AttnSave> self_attn, cross_attn;
    void* kvcat = nullptr;      // decoder: [B*S][L*2D] K/V of every layer (when the model has a kv_cat weight slab)
    std::vector<FfnSave> ffn;
    float* x_final = nullptr;           // input of the output LayerNorm
    float *mean_f = nullptr, *rstd_f = nullptr;
    float *spk_e = nullptr, *spk_h = nullptr, *lang_e = nullptr, *lang_h = nullptr, *spk_dh = nullptr, *lang_dh = nullptr;
    // decoder
    int enc_fused = 0;               // encoder: 0 = the forward ran kernel by kernel, 1 = fused sublayer kernels, 2 = fused FFN sublayers only (S > 128)
    bool enc_wT_done = false;        // the forward left the transposed weight copies behind enc_wT_ev
    void* memT = nullptr;
    void *tgtT = nullptr, *a1 = nullptr, *a2 = nullptr;
    void* outT = nullptr;               // imputed decoder output (T)
    // ragged rows (b2s_decoder_compact_rows): the segment's token-major tensors hold utterance b's frames t < target_lengths[b] at rows
    // [rowoff[b], rowoff[b + 1]) -- Mr = sum(target_lengths) rows instead of B x T; padded tensors only at the segment's boundaries
    bool ragged = false;
    long Mr = 0;                        // rows the row-wise kernels of this segment process (= B T in the padded layout)
    int* rowoff = nullptr;              // device [B + 1]
    float *melc = nullptr, *stopc = nullptr;     // ragged mel / stop outputs before they are scattered into the caller's padded tensors
    float* ga_rows = nullptr;           // guided attention: [Ld][B*H][T] rowsum(P W); ga_small: [0] fwd scale, [1] loss, [2] bwd scale
    float* ga_small = nullptr;
    // postnet
    std::vector<void*> u;               // conv inputs (T), u[0] = cast(inputs)
    std::vector<float*> y, bn_mean, bn_rstd;
};

struct b2s_model {
    b2s_config cfg;
    int dtype = 0, esz = 4;
    int Dm = 0;                                     // memory width
    std::vector<TensorInfo> tinfo;
    std::map<std::string, int> index;
    std::vector<void*> data, grad, shadow;          // per tensor
    std::vector<void*> exp_avg, exp_avg_sq;
    std::vector<void*> conv_wf, conv_wb;            // per postnet layer (T)
    std::vector<int> ragged_lens;                   // b2s_decoder_compact_rows: host copy of target_lengths for the NEXT decoder forward
    float *pe_enc = nullptr, *pe_dec = nullptr;
    int pe_len = 0;
    bool bound = false;
    size_t grad_pad_bytes = 0;                      // b2s_model_set_grad_slot_padding: largest gap between bound gradient ranges that is padding
    // multi-tensor chunk tables (device)
    MtChunk *l2_chunks = nullptr, *adam_chunks = nullptr;
    int n_l2_chunks = 0, n_adam_chunks = 0;
    MtChunk* cast_chunks = nullptr;                 // (a = fp32 master, s = bf16 shadow) of every GEMM weight: one cast launch (b2s_model_sync_weights)
    int n_cast_chunks = 0;
    float* small = nullptr;                         // device scratch: [0..15] misc scalars
    // per-chunk sum of squares of the L2 members, written by the fused Adam step for the parameters it just produced:
    // the next loss evaluation reads the regulariser from here instead of re-reading 83 M parameters.  l2_fresh is
    // cleared whenever parameters may have changed behind the optimizer's back (bind / full weight sync).
    float* l2_part = nullptr;
    bool l2_fresh = false;
    std::vector<void*> owned;                       // hipMalloc'ed buffers
    // split-K slab workspaces, one per stream this model launches split-K GEMMs on ([0] caller's stream, [1] aux stream):
    // launches of one stream are ordered, the two streams never share a slab
    float* sk_ws[2] = {nullptr, nullptr};
    size_t sk_ws_floats = 0;
    void set_ws(GemmArgs& g, hipStream_t st) const { g.ws = sk_ws[(aux && st == aux) ? 1 : 0]; g.ws_floats = g.ws ? sk_ws_floats : 0; }
    // second HIP stream for the weight-gradient GEMMs: they depend only on (dY, X) and feed nothing in the backward chain,
    // so they run concurrently with the dX / attention / LayerNorm kernels of the same layer (the 128x128-tile GEMMs leave
    // CUs idle in their last, partial wave of workgroups).  Hazards on re-used scratch buffers are tracked per buffer.
    mutable hipStream_t aux = nullptr;
    mutable std::vector<hipEvent_t> ev_pool;
    mutable size_t ev_next = 0;
    mutable std::map<const void*, hipEvent_t> aux_readers;     // scratch buffer -> event after its last aux-stream reader
    mutable bool aux_dirty = false;
    // side stream (b2s_model_set_side_stream: the caller's otherwise idle encoder stream): the decoder backward launches the dK / dV kernel of
    // every encoder-decoder attention there.  Its results feed only the memory-side gradients (the layer's kv weight gradient on the second
    // stream, the one d(memory) GEMM at the end of the call), so the main stream's chain never waits for it.  side_ev: event behind the
    // last kernel launched there (own events: the shared pool wraps around while this one is still referenced)
#ifdef B2S_LAB
    mutable bool lab_early_marked = false;
#endif
    mutable hipStream_t side = nullptr;
    mutable hipEvent_t side_ev = nullptr;
    mutable std::vector<hipEvent_t> side_evs;
    mutable size_t side_next = 0;
    hipEvent_t next_event() const { hipEvent_t e = ev_pool[ev_next % ev_pool.size()]; ++ev_next; return e; }
    // Weight-gradient GEMMs of one backward stage are deferred and launched as ONE grouped GEMM on the aux stream when the
    // stage ends (bf16 mode with an aux stream): 7 problems of 18..72 tiles each fill the chip together, no split-K slabs.
    // The stage hook of stage s then fires one stage late (when stage s+1 has been enqueued), so the host never stalls
    // the main stream on the group it has just launched.
    mutable bool dw_group = false;
    // Overwrite mode of a backward pass (b2s_zero_grads_ex, flag B2S_ZERO_GRADS_OVERWRITE_DW): the weight gradients that are written exactly
    // once per pass by a grouped launch (dw_ow: the GEMM weights of the encoder / decoder layers whose stage has >= 32 output tiles) are
    // STORED instead of accumulated, and only the other gradients (zero_chunks: biases, LayerNorm / BatchNorm, embeddings, conv weights,
    // prenet / heads -- everything that accumulates through atomics, split-K slabs or several launches) are cleared, by one kernel
    // instead of a 334 MB memset; the stores also take the GEMM's 16-byte fast epilogue and read nothing.
    std::vector<char> dw_ow;                                   // per tensor
    std::set<const float*> dw_ow_ptrs;                          // their gradient pointers
    mutable bool dw_overwrite_pass = false;
    MtChunk* zero_chunks = nullptr;
    int n_zero_chunks = 0;
    mutable std::vector<GemmArgs> dw_pending;
    mutable int dw_stages_pending = 0;                        // backward stages whose weight-gradient GEMMs are queued in dw_pending
    // Tail policy of the decoder backward when the encoder backward runs beside its end (b2s_decoder_backward_ev + B2S_DEC_BWD_FLUSH_TAIL):
    // the weight-gradient groups of the stages >= dw_hold_from are not handed over stage by stage but all at the end, as a sequence of
    // launches of at most dw_tail_cap tiles each -- they then fill the CUs the encoder chain's 78..224-workgroup kernels
    // leave idle, instead of occupying every CU for 150 us at a time while that chain waits (profiles/NOTES_r03.md)
    mutable int dw_hold_from = -1, dw_tail_cap = 0;
    mutable bool dw_flush_capped = false;
    // launches that only feed parameter gradients (bias / stop-net column sums, the speaker / language nets' backward): queued and
    // issued on the second stream with the next weight-gradient hand-over
    mutable std::vector<std::function<int(hipStream_t)>> aux_jobs;
    int adam_grp[4] = {0, 0, 0, 0};                            // chunk ranges [grp[g], grp[g+1]) of encoder / decoder / postnet (b2s_adam_step_groups)
    const void* adam_wire = nullptr;                           // b2s_adam_set_grad_wire: bf16 gradients (the exchange's wire buffer) laid out like the
    const float* adam_gbase = nullptr;                         // fp32 gradient buffer that starts at adam_gbase
    // Sharded optimizer (b2s_adam_shard: reduce-scatter / all-gather data parallelism): chunk tables of the flat-gradient element ranges this rank
    // owns (updated by b2s_adam_step, packed into the parameter wire) and of the rest (scattered back from the wire after the all-gather)
    MtChunk *shard_chunks = nullptr, *other_chunks = nullptr;
    int n_shard_chunks = 0, n_other_chunks = 0;
    std::vector<std::pair<long, long>> shard_ranges;           // owned [lo, hi) in elements of the flat gradient buffer that starts at shard_gbase
    const float* shard_gbase = nullptr;
    int adam_step_no = 0, adam_step_mask = 0;                  // b2s_adam_step_groups: groups already updated in step adam_step_no
    mutable bool dw_flush_exposed = false;                     // end_stage -> flush_dw: this hand-over is the entry point's drain (nothing overlaps it)
    mutable hipEvent_t grads_mark_ev = nullptr;                // b2s_model_mark_grads_ready: second-stream event behind the gradient work queued so far
    mutable bool grads_marked = false;
    // bf16 mode: the compute-dtype copies of the L encoder-decoder kv_transform weights are one slab [L*2D][D], so that the
    // memory K/V of all layers come from ONE GEMM (N = L*2D) and d(memory) from ONE GEMM over the concatenated dK/dV
    // (K = L*2D) instead of L launches of 56-112 tiles each
    void* kv_cat = nullptr;
    // Fused encoder sublayers (enc_fused.h; bf16 mode, default encoder dims, <= 128 rows per utterance): the backward kernels stream
    // TRANSPOSED bf16 copies of the encoder GEMM weights so that every operand they DMA is K-contiguous -- per layer {Wqkv^T, Wo^T,
    // W1^T, W2^T}, rewritten on the second stream by every training-mode encoder forward (the weights change every step) and
    // ordered before the encoder backward through enc_wT_ev.
    bool enc_fused = false;
    int enc_slab_bf16 = 0;
    // bf16 mode, hidden sizes 512 / 768: the residual gradient that the LayerNorm backward kernels read, add to and write back (3 per decoder
    // layer, 2 per encoder layer) is held in bf16 -- 12 instead of 16 bytes per element and launch for kernels that run at the rate of their
    // read / write mix (B2S_DX_BF16=0: fp32)
    int dx_bf16 = 0;
    std::vector<void*> enc_wT;
    mutable hipEvent_t enc_wT_ev = nullptr;
    mutable LnReduceBatch ln_jobs = {};                       // LayerNorm parameter-gradient reductions queued for the stage's single launch
    // stages whose gradient work has been handed to the second stream (completed by pending_ev) and whose hook has not fired yet /
    // stages that ended since the last hand-over
    mutable std::vector<int> pending_stages, unflushed_stages;
    mutable hipEvent_t pending_ev = nullptr;
    void (*stage_hook)(int, void*) = nullptr;       // called on the host after each backward stage is enqueued
    void* stage_user = nullptr;
    void stage_done(int s) const { if (stage_hook) stage_hook(s, stage_user); }   // callers order the hook's stream first (hook_after_*)
    hipStream_t hook_stream = nullptr;              // stream the hook launches its collective on (b2s_model_set_stage_hook); null: the backward's

    int id(const std::string& n) const;
    float* P(const std::string& n) const { return (float*)data[id(n)]; }
    float* G(const std::string& n) const { return (float*)grad[id(n)]; }
    void* W(const std::string& n) const { return shadow[id(n)]; }
    long numel(const std::string& n) const { return tinfo[id(n)].numel; }
};
