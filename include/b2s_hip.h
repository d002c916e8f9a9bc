/*
 * b2s_hip.h -- C ABI of libb2s_hip.so: the MI355X (gfx950) implementation of the Transformer-TTS
 * ("byte2speech") hot path of mutiann/few-shot-transformer-tts.
 *
 * The reference has no FFI: its hot path is Python calling PyTorch aten ops.  This header is the
 * boundary a maintainer binds instead (ctypes stub in INTEGRATION.md); each entry point names the
 * reference code it replaces.  Conventions:
 *   - plain C, no torch types; every pointer is DEVICE memory unless the name ends in _host;
 *   - tensors are contiguous row-major fp32 unless stated; lengths are int32, token ids int64;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued
 *     asynchronously on it, nothing synchronises;
 *   - functions return 0 on success, non-zero on error (message via b2s_last_error()); they never
 *     abort the process (reference convention: Python exceptions);
 *   - no hidden global state: all state lives in the b2s_model / b2s_ctx handles; one model handle
 *     is used from one host thread at a time (the reference calls the model from the main thread only).
 */
#ifndef B2S_HIP_H
#define B2S_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define B2S_DTYPE_FP32 0 /* parity mode: exact fp32 MFMA (v_mfma_f32_16x16x4_f32)            */
#define B2S_DTYPE_BF16 1 /* performance mode: bf16 MFMA operands, fp32 accumulate / residual */

const char* b2s_last_error(void);
int b2s_version(void);

/* ---- hyper-parameters that shape the model (hyperparams.py:4,24-35,54-61) ------------------------ */
typedef struct b2s_config {
    int32_t num_mels, vocab_size, embed_size, encoder_hidden, decoder_hidden;
    int32_t n_encoder_layer, n_decoder_layer, n_attention_head;
    int32_t prenet_hidden, postnet_hidden, n_postnet_layer;
    int32_t multi_speaker, max_num_speaker, speaker_embedding_size;
    int32_t multi_lingual, max_num_language, language_embedding_size;
    float transformer_dropout_rate, decoder_dropout_rate;
    float reg_weight;
    int32_t compute_dtype; /* B2S_DTYPE_* */
    /* Extensions named by the north-star, absent upstream (both OFF by default):
     * guided-attention loss on every head of every encoder-decoder attention (weight 0 = off), and the few-shot
     * fine-tuning mode in which every `encoder.*` parameter is frozen (no gradient, no all-reduce, no Adam update). */
    float guided_attention_weight, guided_attention_sigma;
    int32_t freeze_encoder;
} b2s_config;

typedef struct b2s_model b2s_model; /* replaces transformer.tacotron.Tacotron (tacotron.py:119-133)  */
typedef struct b2s_ctx b2s_ctx;     /* activations saved by one forward call, consumed by backward   */

/* Tacotron.__init__ (tacotron.py:119-124).  The model owns only compute-dtype weight shadows and
 * sinusoid tables; parameters stay caller-owned (state_dict names/shapes/order of the reference). */
int b2s_model_create(const b2s_config* cfg, b2s_model** out);
void b2s_model_destroy(b2s_model* m);
/* state_dict layout: number of entries, then name / shape / kind of entry i
 * (kind: 1 = parameter, 0 = fp32 buffer, 2 = int64 buffer num_batches_tracked). */
int b2s_model_num_tensors(const b2s_model* m);
int b2s_model_tensor_info(const b2s_model* m, int i, char* name, int name_cap, int64_t* shape, int* ndim, int* kind);
/* Bind device pointers of every state_dict entry (data) and of the gradient of every parameter
 * (grad[i] may be NULL for buffers).  Arrays are host arrays of device pointers. */
int b2s_model_bind(b2s_model* m, void* const* data_host, void* const* grad_host, int n);
/* Refresh the compute-dtype weight shadows (bf16 copies, conv re-layouts) after parameters changed.
 * shadows_fresh != 0: b2s_adam_step already refreshed the bf16 shadows (it writes them in the same pass); only the conv
 * weight re-layouts are redone. */
int b2s_model_sync_weights(b2s_model* m, void* stream, int shadows_fresh);

/* ---- Encoder.forward (tacotron.py:33-44; modules.py:49-69) ---------------------------------------
 * memory_out: [B, S, encoder_hidden (+spk) (+lang)].  train != 0 enables dropout (seeded by `seed`).
 * need_grad != 0 keeps activations in `ws` and returns a ctx for b2s_encoder_backward. */
size_t b2s_encoder_ws_bytes(const b2s_model* m, int B, int S);
int b2s_encoder_forward(b2s_model* m, const int64_t* inputs, const int32_t* input_lengths, const int64_t* spk_ids,
                        const float* language_vecs, int B, int S, int train, uint64_t seed, void* ws, size_t ws_bytes,
                        float* memory_out, void* stream, b2s_ctx** ctx_out);
int b2s_encoder_backward(b2s_model* m, b2s_ctx* ctx, const float* d_memory, void* stream);

/* Ragged decoder rows.  HOST copy of target_lengths ([B] int32, 1 <= length <= T, B <= 64) for the NEXT b2s_decoder_forward on this model that is
 * called with B2S_DEC_PADDED_UNOBSERVED: that segment (forward and its backward) then keeps its token rows ragged -- utterance b owns
 * sum_{b' < b} length[b'] .. + length[b] of the segment's internal [rows, C] tensors -- so every row-wise kernel (GEMMs, LayerNorms, the weight-gradient
 * K walks) runs over sum(target_lengths) rows instead of B x T (the reference masks those rows out: transformer/common.py:51-70, modules.py:142-144).
 * Inputs, outputs and gradients at the boundary stay padded [B, T, *] tensors with zeros on padded rows; results on valid rows are those of the padded
 * layout (forward outputs and activation gradients bit for bit at dropout 0; weight gradients up to the fp32 summation order of their K walk).
 * Dropout: the element index of the decoder's row sites (b2s_dropout_site kind 0) counts RAGGED rows, row = offset[b] + t, in such a segment.
 * The hand-over is consumed by that one forward call (also when it fails or does not qualify); without it the padded layout is used. */
int b2s_decoder_compact_rows(b2s_model* m, const int32_t* target_lengths_host, int B);

/* ---- Decoder.forward (tacotron.py:107-116; modules.py:108-145) -----------------------------------
 * memory [B,S,Dm], targets [B,T,num_mels] -> mels [B,T,num_mels], stop_logits [B,T]. */
size_t b2s_decoder_ws_bytes(const b2s_model* m, int B, int S, int T);
/* memory_ready (hipEvent_t as void*, or NULL): the encoder output arrives from another stream -- the event is waited for on `stream`
 * right before the first kernel that reads `memory` (the memory K / V projection ahead of the first encoder-decoder attention); the prenet
 * and the first layer's self-attention are enqueued before the wait and overlap the encoder forward running on the other stream. */
/* train: bit 0 = training mode (dropout live).  Bit 1 (B2S_DEC_PADDED_UNOBSERVED): the caller reads nothing of this forward at query rows
 * t >= target_lengths[b] except the (masked) outputs -- in particular no alignments of those rows; the attention kernels then skip whole
 * 64-row tiles of padded queries (forward and backward).  Losses, outputs and every gradient are unchanged: the heads mask those rows and
 * the backward zeroes their gradient whatever d_mels / d_stop hold there. */
#define B2S_DEC_PADDED_UNOBSERVED 2
int b2s_decoder_forward(b2s_model* m, const float* memory, const int32_t* input_lengths, const float* targets,
                        const int32_t* target_lengths, int B, int S, int T, int train, uint64_t seed, void* ws,
                        size_t ws_bytes, float* mels_out, float* stop_out, void* memory_ready, void* stream, b2s_ctx** ctx_out);
/* Backward.  d_memory_out [B,S,Dm] is overwritten.  d_guided: device scalar = d loss / d guided_loss (NULL: the guided-attention term gets no
 * gradient).  flags = 0 and dmem_done = NULL: the plain call.  flags bit 0: do not compute d_memory (frozen encoder; d_memory_out may be NULL).  bit 1: the caller's next call
 * on this model and stream is b2s_encoder_backward -- the second stream (weight-gradient GEMMs) is then joined at the end of
 * that call instead of this one; ctx (its workspace) must stay alive until that call has returned, and the last decoder
 * stage's hook fires from inside it. */
#define B2S_DEC_BWD_NO_DMEMORY 1
#define B2S_DEC_BWD_DEFER_JOIN 2
/* bit 2: like DEFER_JOIN the second stream is not joined here, but everything still queued (the last stages' weight-gradient groups) is
 * handed to it by this call, ordered behind this call's stream -- required when the next entry point runs on a DIFFERENT stream */
#define B2S_DEC_BWD_FLUSH_TAIL 4
/* dmem_done (hipEvent_t as void*, or NULL) is recorded on `stream` as soon as d_memory_out is complete -- after the FIRST decoder layer's
 * encoder-decoder attention backward, ahead of that layer's self-attention, the prenet backward and their weight gradients -- so that an
 * encoder backward on another stream can start then (train.py has no counterpart: autograd runs one stream). */
int b2s_decoder_backward(b2s_model* m, b2s_ctx* ctx, const float* d_mels, const float* d_stop, const float* d_guided,
                         int flags, float* d_memory_out, void* dmem_done, void* stream);
/* Guided-attention loss of the forward held in ctx (already multiplied by guided_attention_weight):
 *   weight * mean over layers, heads and valid (b, t < T_b, n < N_b) of  A[b,h,t,n] * (1 - exp(-(n/N_b - t/T_b)^2 / (2 sigma^2)))
 * written to out[0]; if add_to != NULL it is also added to add_to[0] (the total loss).  Error if the weight is 0. */
int b2s_decoder_guided_loss(b2s_model* m, b2s_ctx* ctx, float* out, float* add_to, void* stream);
/* Alignments of the last forward held in ctx (attention.py:88): which = 0 decoder self, 1 encoder-decoder;
 * out [B, H, Lk, Lq] fp32. */
int b2s_decoder_alignment(b2s_model* m, b2s_ctx* ctx, int which, int layer, float* out, void* stream);

/* ---- Postnet.forward (tacotron.py:81-90).  out = (add ? add : 0) + postnet(inputs) ----------------
 * train != 0: batch statistics + running-stat update + dropout; else running statistics. */
size_t b2s_postnet_ws_bytes(const b2s_model* m, int B, int T);
int b2s_postnet_forward(b2s_model* m, const float* inputs, const int32_t* lengths, const float* add, int B, int T,
                        int train, uint64_t seed, void* ws, size_t ws_bytes, float* out, void* stream,
                        b2s_ctx** ctx_out);
/* d_inputs_out [B,T,num_mels] = gradient through the conv stack only (caller adds the skip path).  flags = 0: the plain call.
 * flags bit 0: the caller's next call on this model and stream is b2s_decoder_backward -- the second stream (the conv
 * weight-gradient GEMMs run there) is joined by that call (or, with B2S_DEC_BWD_DEFER_JOIN, by the encoder backward after it);
 * ctx must stay alive until the joining call has returned. */
#define B2S_POST_BWD_DEFER_JOIN 1
int b2s_postnet_backward(b2s_model* m, b2s_ctx* ctx, const float* d_out, float* d_inputs_out, int flags, void* stream);

void b2s_ctx_free(b2s_ctx* ctx);

/* Data-parallel hook (train.py:125 DistributedDataParallel): `hook(stage, user)` is called on the host right
 * after the kernels that complete the parameter gradients of a backward stage have been enqueued, so the
 * caller can launch that stage's gradient all-reduce (RCCL) while later stages still compute.  Stages in
 * execution order: 0 postnet; 1 decoder heads + output LayerNorm; 2..1+Ld decoder layers Ld-1..0; 2+Ld prenet
 * + decoder pe_scale; 3+Ld speaker/language nets + encoder output LayerNorm; 4+Ld..3+Ld+Le encoder layers
 * Le-1..0; 4+Ld+Le byte embedding + encoder pe_scale.
 * stream: the stream the hook launches its collective on.  NULL: the stream of the backward call -- the library then makes
 * that stream wait for the second stream's weight-gradient work of the stage before the hook fires.  Another stream: only
 * that stream waits, the backward pass is not held up (the hook must launch its work on it; the final optimizer step has to
 * wait for the collectives as before).  hook = NULL removes the hook. */
int b2s_model_set_stage_hook(b2s_model* m, void (*hook)(int stage, void* user), void* user, void* stream);
/* The engine's second stream (hipStream_t; NULL before b2s_model_bind or when the model runs single-stream): the stream its weight-gradient
 * work runs on -- ONE stream per device and process, shared by every model bound there and never destroyed (HIP assigns hardware queues when a
 * stream is created; a fresh stream per model eventually lands on the caller's queue and serialises the step, profiles/NOTES_r06.md section 8).  A data-parallel caller passes it to b2s_model_set_stage_hook, so that the gradient exchange is launched from the
 * stream that completes the gradients instead of a fifth stream: more than four concurrently ACTIVE HIP streams (main, second, encoder,
 * exchange, RCCL's own) were measured at 12.6 ms per step against 7.9 with four (MI355X, profiles/NOTES_r04.md). */
void* b2s_model_second_stream(b2s_model* m);
/* Side stream (hipStream_t, or NULL to clear): a stream of the caller that is idle while b2s_decoder_backward runs -- HipTrainer passes the
 * stream its encoder forward / backward run on.  The decoder backward then launches the dK / dV kernel of every encoder-decoder attention
 * (transformer/attention.py:72-92 under autograd: the gradient of the memory-side K / V) there: its results feed only the layer's kv weight
 * gradient and the single d(memory) GEMM at the end of the call, so the query-side chain on the call's stream does not wait for it.  The
 * call's stream has joined the side stream's work when b2s_decoder_backward returns (and where it records dmem_done).  Results are
 * identical with and without a side stream (same kernels, same order of every reduction).  No further stream is created. */
int b2s_model_set_side_stream(b2s_model* m, void* stream);
/* Give up a backward pass between its entry points (after a failed call, or when the caller will not make the joining call that
 * B2S_POST_BWD_DEFER_JOIN / B2S_DEC_BWD_DEFER_JOIN promised): queued weight-gradient work, reductions and stage hooks are dropped
 * unlaunched / unfired, `stream` waits for what the second stream is already running.  Call before freeing the contexts.  The gradient
 * buffers are incomplete afterwards.  b2s_zero_grads does the same when it finds such leftovers. */
int b2s_model_backward_abort(b2s_model* m, void* stream);
/* Process-wide tile-shape policy of the large bf16 GEMMs: 0 = per-shape choice (256x96 tiles where that gives whole rounds of one
 * workgroup per CU: fastest with all 256 CUs free), 4 = 256x128 tiles everywhere (192 / 576 instead of 256 / 768 workgroups for the
 * N = 768 / 2304 projections: no second round when a communication library's kernels hold some CUs).  The data-parallel trainer
 * (world size > 1) selects 4; B2S_GEMM256_NB overrides both.  Results do not depend on the policy beyond fp32 summation order. */
int b2s_gemm_set_tile_policy(int policy);

/* ---- compute_loss (tacotron.py:136-158) ------------------------------------------------------------
 * losses_out[7] = loss, bef_loss, aft_loss, mse_loss, l2, stop_loss, sum(lengths); aft_losses_out[B].
 * scratch: >= (4 + B) floats. */
int b2s_loss_forward(b2s_model* m, const float* mel_bef, const float* mel_aft, const float* stop_logits,
                     const float* mel_targets, const int32_t* target_lengths, int B, int T, float* losses_out,
                     float* aft_losses_out, float* scratch, void* stream);
/* Gradients of w[0]*bef_loss + w[1]*aft_loss + w[2]*stop_loss w.r.t. mel_bef / mel_aft / stop_logits;
 * w = device float[3] (NULL -> 1,1,1, i.e. the gradient of `loss` without the L2 term). */
int b2s_loss_backward(b2s_model* m, const float* mel_bef, const float* mel_aft, const float* stop_logits,
                      const float* mel_targets, const int32_t* target_lengths, int B, int T, const float* w,
                      float* d_bef, float* d_aft, float* d_stop, void* stream);
/* grad[p] += reg_weight * (*grad_scale) * p for the L2 member set (tacotron.py:144-146). */
int b2s_l2_backward(b2s_model* m, const float* grad_scale, void* stream);

/* ---- autoregressive decode (synthesize.py:17-72 eval_batch) -----------------------------------------
 * KV-cached single-frame steps; with use_graph the step is captured once in a hipGraph and replayed per
 * frame (step index / stop flags / lengths live in device memory).  Reference semantics are kept: a sample
 * stops when its stop logit > 0, finished samples emit exact zeros, lengths follow the reference's
 * target_lengths (incl. its off-by-one for samples that never stop).  Dropout (train != 0, the reference
 * synthesises with decoder.train()) uses the in-kernel RNG salted with the step index. */
typedef struct b2s_decode_state b2s_decode_state;
size_t b2s_decode_ws_bytes(const b2s_model* m, int B, int S, int max_frames, int keep_self_alignments);
int b2s_decode_begin(b2s_model* m, const float* memory, const int32_t* input_lengths, int B, int S, int max_frames, int train,
                     uint64_t seed, int keep_self_alignments, void* ws, size_t ws_bytes, void* stream, b2s_decode_state** out);
int b2s_decode_run(b2s_model* m, b2s_decode_state* s, int n_steps, int use_graph, void* stream);
/* blocking: frames generated so far and whether every sample has stopped (the only host sync of the loop) */
int b2s_decode_status(b2s_decode_state* s, int* frames_host, int* all_finished_host, void* stream);
/* mels_out [B, n_frames, num_mels], lengths_out [B] int32 (device) */
int b2s_decode_fetch(b2s_model* m, b2s_decode_state* s, int n_frames, float* mels_out, int32_t* lengths_out, void* stream);
/* which = 1: encoder-decoder rows -> [B,H,S,n_frames]; which = 0: self rows -> [B,H,n_frames,n_frames] */
int b2s_decode_alignment(b2s_model* m, b2s_decode_state* s, int which, int layer, int n_frames, float* align_out, void* stream);
void b2s_decode_end(b2s_decode_state* s);

/* ---- optimizer (train.py:130-131,188-189): Adam(lr, eps) with bias correction, over all bound
 * parameters; m/v are caller-owned flat fp32 state laid out like the bound grads.  step is 1-based.
 * l2 > 0 folds the L2 gradient (l2 * p) in; grad_scale multiplies the bound gradient first (1/world). */
int b2s_adam_bind(b2s_model* m, void* const* exp_avg_host, void* const* exp_avg_sq_host, int n);
int b2s_adam_step(b2s_model* m, float lr, int step, float beta1, float beta2, float eps, float l2, float grad_scale,
                  void* stream);
/* Data-parallel runs with a bf16 gradient payload: `wire_bf16` is the exchange's wire buffer -- bf16, element i = gradient element i of the
 * flat fp32 gradient buffer that starts at `grad_base` (the bound gradient tensors are slices of it).  From now on every b2s_adam_step*
 * reads its gradients from the wire buffer, i.e. consumes the all-reduced sum where the collective left it: no unpack pass, no fp32
 * re-read (train.py:125,130-131: DDP's averaged gradient feeding optim.step()).  NULL restores the fp32 gradient buffers. */
int b2s_adam_set_grad_wire(b2s_model* m, const void* wire_bf16, const float* grad_base);
/* One optimizer step applied in pieces: `groups` is a mask of parameter groups (B2S_ADAM_ENCODER | _DECODER | _POSTNET)
 * whose gradients are final.  Every group must be stepped exactly once per `step`.
 * behind_mark = 0: the groups are updated on `stream` after the second stream has been joined.
 * behind_mark = 1: the groups are updated on `stream` by a grid capped at 512 workgroups, behind the mark b2s_model_mark_grads_ready left on
 * the second stream (and nothing later): the trainer marks after the decoder backward, enqueues the encoder backward on its own stream, and
 * only then issues the decoder / postnet update -- which runs beside the encoder backward on the device, while a host-side failure in the
 * encoder backward still finds no part of the step applied. */
#define B2S_ADAM_ENCODER 1
#define B2S_ADAM_DECODER 2
#define B2S_ADAM_POSTNET 4
int b2s_adam_step_groups(b2s_model* m, float lr, int step, float beta1, float beta2, float eps, float l2, float grad_scale,
                         int groups, int behind_mark, void* stream);
int b2s_model_mark_grads_ready(b2s_model* m);
/* Sharded optimizer for reduce-scatter / all-gather data parallelism (train.py:125,130-131,188-189 semantics -- the mean gradient feeds Adam and every
 * rank ends the step with the same parameters -- with the optimizer's 30 bytes per parameter paid on 1/N of the parameters per rank).
 * b2s_adam_shard: lo / hi = the n sorted, disjoint element ranges of the flat gradient buffer starting at grad_base that THIS rank owns (the slices a
 * reduce-scatter of the gradient buckets leaves it); b2s_adam_step then updates exactly those (b2s_adam_step_groups is refused).  n = 0 un-shards.
 * b2s_param_wire: `wire` = a flat fp32 buffer laid out like the gradient buffer.  direction 0 (after b2s_adam_step): the owned parameters are copied
 * into their wire positions; the caller all-gathers the wire; direction 1: every parameter element this rank does not own is overwritten from the
 * wire, together with its compute-dtype shadow / conv GEMM images -- fp32 masters stay REPLICATED (checkpoints, the fp32-read parameters and the
 * parity mode are untouched).  HipTrainer(dp_mode="rs_ag") drives it; the default remains bucketed all-reduce. */
int b2s_adam_shard(b2s_model* m, const float* grad_base, const int64_t* lo, const int64_t* hi, int n);
int b2s_param_wire(b2s_model* m, float* wire, int direction, void* stream);
/* Zero every bound parameter gradient (flags = 0).  Every *_backward entry point ACCUMULATES into the bound gradient
 * buffers (several use atomics), so the host calls this once at the start of each backward pass.
 * flags = B2S_ZERO_GRADS_OVERWRITE_DW: the caller is about to run ONE complete backward pass (postnet, decoder, encoder -- every segment
 * exactly once, as b2s_hip.trainer.HipTrainer does).  bf16 mode: the weight gradients of the encoder / decoder layers are then STORED by
 * their grouped weight-gradient launch instead of accumulated, and only the remaining gradients (biases, LayerNorm / BatchNorm, embeddings,
 * convolutions, prenet, heads) are cleared -- one small kernel instead of a 334 MB memset, and no read-modify-write in the weight-gradient
 * epilogues.  In fp32 mode the flag is ignored. */
#define B2S_ZERO_GRADS_OVERWRITE_DW 1
int b2s_zero_grads(b2s_model* m, void* stream, int flags);
/* Gradient tensors bound as slots of ONE flat buffer with up to `bytes` of alignment padding between them (the Python engine: 256):
 * b2s_zero_grads then clears the padding along with the slots (one memset).  Default 0: separately bound gradient tensors are cleared
 * range by range and nothing between them is touched. */
int b2s_model_set_grad_slot_padding(b2s_model* m, int bytes);

/* ---- op level (used by the standalone modules and the op parity tests) --------------------------- */
/* C[M,N] = A * B^T style GEMM family; see csrc/gemm.h.  dtype operands are fp32 or raw bf16. */
typedef struct b2s_gemm_desc {
    int32_t dtype, trans_a, trans_b, M, N, K, lda, ldb, ldc, c_fp32;
    int32_t batch, batch_inner;
    int64_t a_bs_o, a_bs_i, b_bs_o, b_bs_i, c_bs_o, c_bs_i;
    float alpha;
    int32_t relu, accumulate;
    float drop_p;
    uint64_t seed;
    int32_t conv_cin_a, conv_T, conv_dw_cin; /* conv gather on A (token rows) */
    int32_t rows_per_batch;
} b2s_gemm_desc;
int b2s_gemm(const b2s_gemm_desc* d, const void* A, const void* B, void* C, const float* bias, const float* residual,
             const int32_t* row_len, const int32_t* conv_len, void* stream);
/* The same GEMM with the reduction split over `splitk` workgroup groups (weight-gradient shapes: small M x N, long K).
 * Requires c_fp32 = 1, accumulate = 1 and a linear epilogue.  ws: caller-owned slab workspace of ws_floats >= splitk*M*N
 * floats that must not be shared by launches on different streams (partial tiles are written there and summed into C
 * by a second kernel on the same stream); ws = NULL: partial tiles are added into C with fp32 atomics.  The library holds
 * no slab of its own (csrc/gemm.h: GemmArgs::ws). */
int b2s_gemm_splitk(const b2s_gemm_desc* d, int splitk, const void* A, const void* B, float* C, float* ws, size_t ws_floats,
                    void* stream);
int b2s_layernorm_forward(int dtype, const float* x, const float* gamma, const float* beta, void* y, float* mean,
                          float* rstd, int M, int D, float eps, void* stream);
int b2s_layernorm_backward(int dtype, const void* dy, const float* x, const float* gamma, const float* mean,
                           const float* rstd, float* dx, float* dgamma, float* dbeta, int M, int D, void* stream);
/* MultiheadAttention core (attention.py:72-92) on head-interleaved rows: q [B,Lq,H*dh] (ld ldq) etc.
 * mask_mode bit0 = key-length mask (klen), bit1 = causal; bias = optional dense additive fp32 bias
 * broadcast as bias[b*bias_sb + q*bias_sq + k].  ws: scratch of b2s_attention_ws_bytes().  P_out (optional)
 * receives softmax weights [B,H,Lq,ldp] in compute dtype, ldp = round_up(Lk, 8). */
size_t b2s_attention_ws_bytes(int dtype, int B, int H, int Lq, int Lk);
int b2s_attention_forward(int dtype, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* ctx,
                          int ldc, int B, int H, int Lq, int Lk, int dh, int mask_mode, const int32_t* klen,
                          const float* bias, int64_t bias_sb, int64_t bias_sq, float drop_p, uint64_t seed, void* ws,
                          void* P_out, void* Pd_out, void* stream);
int b2s_attention_backward(int dtype, const void* dctx, int ldc, const void* q, int ldq, const void* k, int ldk,
                           const void* v, int ldv, const void* P, const void* Pd, void* dq, int lddq, void* dk,
                           int lddk, void* dv, int lddv, int B, int H, int Lq, int Lk, int dh, float drop_p,
                           uint64_t seed, void* ws, void* stream);
/* Fused (flash-style) form of the same attention core: logits never reach HBM; lse_out [B,H,Lq] = log-sum-exp
 * of the scaled masked logits (saved for backward / alignments).  Head sizes 32, 64, 96; no dense bias. */
int b2s_flash_attention_forward(int dtype, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* ctx,
                                int ldc, int B, int H, int Lq, int Lk, int dh, int mask_mode, const int32_t* klen,
                                float drop_p, uint64_t seed, float* lse_out, void* stream);
int b2s_flash_attention_backward(int dtype, const void* dctx, const void* ctx, int ldc, const void* q, int ldq,
                                 const void* k, int ldk, const void* v, int ldv, const float* lse, float* dsum_scratch,
                                 void* dq, int lddq, void* dk, int lddk, void* dv, int lddv, int B, int H, int Lq,
                                 int Lk, int dh, int mask_mode, const int32_t* klen, float drop_p, uint64_t seed,
                                 void* stream);
/* align_out [B,H,Lk,Lq] (attention.py:88), recomputed from q, k and lse */
int b2s_flash_attention_align(int dtype, const void* q, int ldq, const void* k, int ldk, const float* lse, int B, int H,
                              int Lq, int Lk, int dh, int mask_mode, const int32_t* klen, float* align_out, void* stream);
int b2s_align_from_probs(int dtype, const void* P, float* align, int B, int H, int Lq, int Lk, void* stream);
/* out = (a + b) + c (fp32, n elements; c = NULL: out = a + b): the three contributions to d(mel_before) -- postnet input gradient, d(mel_after) routed around it and the direct loss
 * term (tacotron.py:126-133 + autograd) -- in one launch */
int b2s_add3(const float* a, const float* b, const float* c, float* out, int64_t n, void* stream);
/* fp32 <-> compute dtype.  With dtype = B2S_DTYPE_BF16 these are also the gradient payload conversions of the data-parallel exchange (b2s_hip/dp.py):
 * fp32 gradients -> bf16 wire buffer and back (half the bytes per all-reduce over xGMI; parameters, Adam moments and the accumulation inside a
 * rank stay fp32). */
int b2s_cast(int dtype, const float* in, void* out, int64_t n, void* stream);      /* fp32 -> compute dtype */
int b2s_cast_back(int dtype, const void* in, float* out, int64_t n, void* stream); /* compute dtype -> fp32 */
/* keep-mask of the dropout RNG for element indices [0,n): out[i] = 1 or 0 (statistical tests) */
int b2s_dropout_mask(float p, uint64_t seed, uint32_t op_id, uint8_t* out, int64_t n, void* stream);
/* keep-mask of the TRAINING kernels' attention-weight dropout for a [rows, Lk] weight matrix (row = (b * H + h) * Lq + q).  A row has a seed drawn with
 * the full hash, seed = lowbias32(row * 0x9E3779B1 + key); the four keys 4 kq .. 4 kq + 3 of the row share one mixing step y = x ^ (x >> 16),
 * x = seed + kq * 0x9E3779B1, and take their 16-bit fields from two multiplies of it: word = y * ((k & 2) ? 0xC2B2AE35 : 0x85EBCA6B), field = the
 * (k & 1)-th half of word; a key is kept when (int16) field >= ((p * 2^32) >> 16) - 32768 (a lane of the attention kernels owns runs of four adjacent
 * keys of one row: four integer operations per four weights) */
int b2s_dropout_mask_attn(float p, uint64_t seed, uint32_t op_id, uint8_t* out, int64_t rows, int Lk, void* stream);
/* Read-only query of the dropout-site table (csrc/drop_sites.h) -- everything a checker needs to regenerate the mask the model path
 * applied at any of the reference's dropout calls (transformer/modules.py:18,55,64,67,120,132,138,141, attention.py:89, tacotron.py:58,62,89),
 * so that the dropout-ON forward / backward can be compared with a CPU restatement element for element.  site: "encoder.embed",
 * "encoder.attn", "encoder.attn_res", "encoder.ffn_hidden", "encoder.ffn_res", "decoder.prenet0", "decoder.prenet1", "decoder.embed",
 * "decoder.self_attn", "decoder.self_res", "decoder.cross_attn", "decoder.cross_res", "decoder.ffn_hidden", "decoder.ffn_res",
 * "postnet.conv"; layer: layer (postnet: conv) index, 0 for the per-segment sites.  decode = 0: the training segments
 * (b2s_{encoder,decoder,postnet}_forward, seeded by their `seed` argument); decode = 1: the autoregressive loop (b2s_decode_begin's seed).
 * The mask is keep(idx) = lowbias32(idx * 0x9E3779B1 + key) >= p * 2^32, key = f(seed, *op_id_out) as in b2s_dropout_mask, with
 *   *kind_out = 0: idx = row * C + column of the activation [rows, C] the site acts on (row = b * L + position; decode loop: row = b);
 *   *kind_out = 1: the softmax weights [B, H, Lq, Lk] -- training segments: the row-seed / key-quad rule of b2s_dropout_mask_attn with rows = (b * H + h) * Lq + q;
 *                  decode loop: idx = (b * H + h) * 4096 + k in the element rule above;
 * in the decode loop the key of frame t is additionally XORed with lowbias32(salt(t)): *salt_out = 1: t * 2246822519 + 3266489917,
 * 2: t + 0x9e3779b9, 3: t * 2654435761 + 77 (0: none). */
int b2s_dropout_site(const char* site, int layer, int decode, uint32_t* op_id_out, int* kind_out, int* salt_out);

/* ---- measurement: per-launch HIP-event timing of the MFMA GEMM kernel on its launch stream (bench.py).
 * variant v = dtype*8 + trans_a*4 + trans_b*2 + conv_gather (16 variants), v = 16: grouped bf16 weight-gradient launches;
 * out[v*3 + {0,1,2}] = flops, milliseconds, launches for v < n_variants. */
void b2s_prof_enable(int on);
int b2s_prof_collect(double* out, int n_variants);

/* ---- fused encoder sublayer kernels, op level (csrc/enc_fused.h).  bf16 compute mode, the default encoder dims (hidden 512, 8 heads of 64,
 * FFN 2048) and S <= 128 rows per utterance; M = B * S token rows, utterance b = rows [b*S, (b+1)*S).  The engine's encoder forward /
 * backward (b2s_encoder_forward / _backward) launches exactly these per sublayer; the entry points exist for the parity tests.
 * Reference: transformer/modules.py:49-69 (TransformerEncoder.forward), transformer/attention.py:53-122, transformer/modules.py:8-20.
 * slabs: [8][M][512] PARTIAL sublayer outputs (fp32, or bf16 when slab_bf16), one per head or per pair of hidden slices {j, j + 8} of 128
 * units; the reduce + LayerNorm kernels sum them in slab order (ns = 8). */
/* MultiheadAttention.forward minus the residual: slabs[h] = (softmax(q_h k_h^T / 8 + key mask) v_h) Wo[:, h*64:(h+1)*64]^T ; also writes the
 * head-interleaved qkv [M,1536], ctx [M,512] (bf16) and the log-sum-exp rows [B*8, S] the backward needs. */
int b2s_encf_attention_forward(const void* hN, const void* Wqkv, const void* Wo, const int32_t* klen, int B, int S, float drop_p, uint64_t seed,
                               uint32_t op_id, void* qkv, void* ctx, float* lse, void* slabs, int slab_bf16, void* stream);
/* its backward w.r.t. hN (partial slabs) and the d[q k v] operand [M,1536] of the weight-gradient GEMM; WoT / WqkvT: transposed bf16 weights */
int b2s_encf_attention_backward(const void* dY, const void* qkv, const void* ctx, const float* lse, const void* WoT, const void* WqkvT,
                                const int32_t* klen, int B, int S, float drop_p, uint64_t seed, uint32_t op_id, void* dqkv, void* slabs, int slab_bf16,
                                void* stream);
/* FFNLayer minus the residual.  backward = 0: slabs[j] = sum over the hidden slices s in {j, j + 8} of dropout(relu(X W1[s]^T)) W2[:, s]^T,
 * f_io <- the hidden activations [M,2048]; backward = 1: X = dY, Wa = W2^T, Wb = W1^T, f_io = the saved activations (ReLU / dropout
 * mask), dz <- d hidden [M,2048], slabs[j] = partial d LN output */
int b2s_encf_ffn_sublayer(int backward, const void* X, const void* Wa, const void* Wb, void* f_io, void* dz, int B, int S, float drop_p, uint64_t seed,
                          uint32_t op_id, void* slabs, int slab_bf16, void* stream);
/* x_out = x_in + dropout(sum_s slabs[s]); h = LayerNorm(x_out) (bf16 [M,512] and / or fp32 with leading dimension ldh32); mean / rstd rows */
int b2s_encf_reduce_layernorm_forward(const float* x_in, const void* slabs, int ns, int slab_bf16, float drop_p, uint64_t seed, uint32_t op_id,
                                      const float* gamma, const float* beta, float* x_out, void* h, float* h32, int ldh32, float* mean, float* rstd,
                                      int M, void* stream);
/* dx += LayerNorm'(sum_s slabs[s]); dgamma / dbeta += (caller zeroes); dy2 (optional) = bf16(dropout(dx)); ws: 768 * 1024 floats of scratch */
int b2s_encf_reduce_layernorm_backward(const void* slabs, int ns, int slab_bf16, const float* x_in, const float* gamma, const float* mean,
                                       const float* rstd, float* dx, float* dgamma, float* dbeta, float* ws, void* dy2, float drop_p, uint64_t seed,
                                       uint32_t op_id, int M, void* stream);
/* dst[C][R] = src[R][C]^T (bf16, R and C multiples of 64): the transposed weight copies of the fused backward kernels */
int b2s_transpose_bf16(const void* src, void* dst, int R, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* B2S_HIP_H */
