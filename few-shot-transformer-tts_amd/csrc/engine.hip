// Model-level orchestration of the Transformer-TTS hot path on MI355X: encoder / decoder / postnet /
// loss forward and hand-derived backward, built from the MFMA GEMM (gemm.hip) and the row kernels
// (rowops.hip).  Reference: transformer/tacotron.py, transformer/modules.py, transformer/attention.py.
//
// Memory plan (HBM): every activation is token-major [B*L, C].  The residual stream, statistics,
// parameter gradients and logits are fp32; GEMM operands ("T") are bf16 in performance mode and fp32
// in parity mode.  One caller-provided workspace per forward call holds the saved activations and the
// scratch of both passes (288 GB of HBM3E: nothing is recomputed, nothing is aliased).
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include "engine.h"
#include "attention.h"
#include "enc_fused.h"
#include "drop_sites.h"

thread_local char g_b2s_err[512] = "";
int b2s_fail(const char* file, int line, const char* fmt, ...) {
    char msg[400];
    va_list ap; va_start(ap, fmt); vsnprintf(msg, sizeof(msg), fmt, ap); va_end(ap);
    const char* base = strrchr(file, '/');
    snprintf(g_b2s_err, sizeof(g_b2s_err), "%s:%d: %s", base ? base + 1 : file, line, msg);
    return 1;
}

int b2s_model::id(const std::string& n) const {
    auto it = index.find(n);
    return it == index.end() ? -1 : it->second;
}

namespace {

inline int rup8(int x) { return (x + 7) & ~7; }
inline hipStream_t S_(void* s) { return (hipStream_t)s; }

// ------------------------------------------------------------------------------------------------ layout
void add_t(b2s_model* m, const std::string& name, std::vector<int64_t> shape, int kind, bool gw) {
    TensorInfo t;
    t.name = name; t.shape = shape; t.kind = kind; t.gemm_weight = gw;
    t.numel = 1; for (auto d : shape) t.numel *= d;
    t.l2 = kind == 1 && name.find("weight") != std::string::npos && name.find("layer_norm") == std::string::npos &&
           name.find("batchnorm") == std::string::npos && name.find("encoder.speaker_embed") == std::string::npos &&
           name.find("encoder.embed") == std::string::npos;                      // tacotron.py:144-146
    m->index[name] = (int)m->tinfo.size();
    m->tinfo.push_back(t);
}
std::string nm(const std::string& p, const char* list, int i, const char* leaf) {
    return p + list + "." + std::to_string(i) + "." + leaf;
}
// state_dict layout of Tacotron(hp): registration order of the reference modules (tacotron.py:8-123,
// modules.py:23-106, attention.py:30-51)
void build_layout(b2s_model* m) {
    const b2s_config& c = m->cfg;
    const int De = c.embed_size, Dh = c.encoder_hidden, Dd = c.decoder_hidden;
    const int Dm = m->Dm;
    auto ln = [&](const std::string& p) { add_t(m, p + ".weight", {0}, 1, false); add_t(m, p + ".bias", {0}, 1, false); };
    auto ln_n = [&](const std::string& p, int n) {
        add_t(m, p + ".weight", {n}, 1, false); add_t(m, p + ".bias", {n}, 1, false);
    };
    (void)ln;
    add_t(m, "encoder.embed.weight", {c.vocab_size, De}, 1, false);
    if (c.multi_speaker) {
        add_t(m, "encoder.speaker_embed.weight", {c.max_num_speaker, c.speaker_embedding_size}, 1, false);
        add_t(m, "encoder.speaker_layer.weight", {c.speaker_embedding_size, c.speaker_embedding_size}, 1, false);
        add_t(m, "encoder.speaker_layer.bias", {c.speaker_embedding_size}, 1, false);
    }
    if (c.multi_lingual) {
        add_t(m, "encoder.language_embed.weight", {c.language_embedding_size, c.max_num_language}, 1, false);
        add_t(m, "encoder.language_layer.weight", {c.language_embedding_size, c.language_embedding_size}, 1, false);
        add_t(m, "encoder.language_layer.bias", {c.language_embedding_size}, 1, false);
    }
    std::string p = "encoder.encoder.";
    add_t(m, p + "pe_scale", {}, 1, false);
    for (int i = 0; i < c.n_encoder_layer; ++i) {
        int n = i == 0 ? De : Dh;
        add_t(m, nm(p, "self_attentions", i, "qkv_transform.weight"), {3 * n, n}, 1, true);
        add_t(m, nm(p, "self_attentions", i, "output_transform.weight"), {n, n}, 1, true);
    }
    for (int i = 0; i < c.n_encoder_layer; ++i) ln_n(p + "attn_layer_norms." + std::to_string(i), i == 0 ? De : Dh);
    for (int i = 0; i < c.n_encoder_layer; ++i) {
        add_t(m, nm(p, "ffn_layers", i, "input_layer.weight"), {4 * Dh, Dh}, 1, true);
        add_t(m, nm(p, "ffn_layers", i, "output_layer.weight"), {Dh, 4 * Dh}, 1, true);
    }
    for (int i = 0; i < c.n_encoder_layer; ++i) ln_n(p + "ffn_layer_norms." + std::to_string(i), Dh);
    ln_n(p + "output_layer_norm", Dh);
    add_t(m, "decoder.prenet.dense0.weight", {c.prenet_hidden, c.num_mels}, 1, true);
    add_t(m, "decoder.prenet.dense0.bias", {c.prenet_hidden}, 1, false);
    add_t(m, "decoder.prenet.dense1.weight", {c.prenet_hidden, c.prenet_hidden}, 1, true);
    add_t(m, "decoder.prenet.dense1.bias", {c.prenet_hidden}, 1, false);
    add_t(m, "decoder.prenet.dense_final.weight", {Dd, c.prenet_hidden}, 1, true);
    p = "decoder.decoder.";
    add_t(m, p + "pe_scale", {}, 1, false);
    for (int i = 0; i < c.n_decoder_layer; ++i) {
        int n = i == 0 ? Dm : Dd;
        add_t(m, nm(p, "self_attentions", i, "qkv_transform.weight"), {3 * n, n}, 1, true);
        add_t(m, nm(p, "self_attentions", i, "output_transform.weight"), {n, n}, 1, true);
    }
    for (int i = 0; i < c.n_decoder_layer; ++i) ln_n(p + "attn_layer_norms." + std::to_string(i), i == 0 ? Dm : Dd);
    for (int i = 0; i < c.n_decoder_layer; ++i) {
        add_t(m, nm(p, "encdec_attentions", i, "q_transform.weight"), {Dd, Dd}, 1, true);
        add_t(m, nm(p, "encdec_attentions", i, "kv_transform.weight"), {2 * Dd, Dd}, 1, true);
        add_t(m, nm(p, "encdec_attentions", i, "output_transform.weight"), {Dd, Dd}, 1, true);
    }
    for (int i = 0; i < c.n_decoder_layer; ++i) ln_n(p + "encdec_layer_norms." + std::to_string(i), i == 0 ? Dm : Dd);
    for (int i = 0; i < c.n_decoder_layer; ++i) {
        add_t(m, nm(p, "ffn_layers", i, "input_layer.weight"), {4 * Dd, Dd}, 1, true);
        add_t(m, nm(p, "ffn_layers", i, "output_layer.weight"), {Dd, 4 * Dd}, 1, true);
    }
    for (int i = 0; i < c.n_decoder_layer; ++i) ln_n(p + "ffn_layer_norms." + std::to_string(i), Dd);
    ln_n(p + "output_layer_norm", Dd);
    add_t(m, "decoder.mel_net.weight", {c.num_mels, Dd}, 1, true);
    add_t(m, "decoder.stop_net.weight", {1, Dd}, 1, false);
    add_t(m, "decoder.stop_net.bias", {1}, 1, false);
    for (int i = 0; i < c.n_postnet_layer; ++i) {
        int cin = i == 0 ? c.num_mels : c.postnet_hidden;
        int cout = i == c.n_postnet_layer - 1 ? c.num_mels : c.postnet_hidden;
        add_t(m, "postnet.conv_layers." + std::to_string(i) + ".weight", {cout, cin, 5}, 1, false);
    }
    for (int i = 0; i < c.n_postnet_layer; ++i) {
        int cout = i == c.n_postnet_layer - 1 ? c.num_mels : c.postnet_hidden;
        std::string q = "postnet.batchnorm_layers." + std::to_string(i) + ".";
        add_t(m, q + "weight", {cout}, 1, false);
        add_t(m, q + "bias", {cout}, 1, false);
        add_t(m, q + "running_mean", {cout}, 0, false);
        add_t(m, q + "running_var", {cout}, 0, false);
        add_t(m, q + "num_batches_tracked", {}, 2, false);
    }
}

// sinusoid table, float64 then cast (common.py:4-29)
void fill_pe(std::vector<float>& out, int length, int channels) {
    const int nts = channels / 2;
    const double inc = std::log(1.0e4 / 1.0) / (double)(nts - 1);
    out.assign((size_t)length * channels, 0.f);
    for (int i = 0; i < nts; ++i) {
        const double inv = 1.0 * std::exp((double)i * -inc);
        for (int pos = 0; pos < length; ++pos) {
            const double st = (double)pos * inv;
            out[(size_t)pos * channels + i] = (float)std::sin(st);
            out[(size_t)pos * channels + nts + i] = (float)std::cos(st);
        }
    }
}
int ensure_pe(b2s_model* m, int len) {
    if (len <= m->pe_len) return 0;
    int n = 2048; while (n < len) n *= 2;
    std::vector<float> h;
    float *pe_e = nullptr, *pe_d = nullptr;
    fill_pe(h, n, m->cfg.embed_size);
    B2S_HIP(hipMalloc(&pe_e, h.size() * 4)); B2S_HIP(hipMemcpy(pe_e, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    fill_pe(h, n, m->cfg.decoder_hidden);
    B2S_HIP(hipMalloc(&pe_d, h.size() * 4)); B2S_HIP(hipMemcpy(pe_d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    // old tables (if any) stay alive until the model is destroyed: earlier launches may still read them
    m->owned.push_back(pe_e); m->owned.push_back(pe_d);
    m->pe_enc = pe_e; m->pe_dec = pe_d; m->pe_len = n;
    return 0;
}

// ------------------------------------------------------------------------------------------------ GEMM helpers
// Y[M,N] = X[M,K] * W[N,K]^T
int linear(const b2s_model* m, hipStream_t st, const void* X, int ldx, const void* W, int M, int N, int K, void* out,
           int out_fp32, int ldo, const GemmEpilogue& e) {
    GemmArgs g;
    g.A.p = X; g.A.ld = ldx; g.A.R = M; g.A.C = K;
    g.B.p = W; g.B.ld = K; g.B.R = N; g.B.C = K;
    g.M = M; g.N = N; g.K = K; g.C = out; g.c_fp32 = out_fp32; g.ldc = ldo; g.epi = e;
    return b2s_gemm_launch(g, m->dtype, false, false, st);
}
// dX[M,Nin] = dY[M,Kout] * W[Kout,Nin]
int linear_dx(const b2s_model* m, hipStream_t st, const void* dY, int lddy, const void* W, int M, int Nin, int Kout,
              void* out, int out_fp32, int ldo, const GemmEpilogue& e) {
    GemmArgs g;
    g.A.p = dY; g.A.ld = lddy; g.A.R = M; g.A.C = Kout;
    g.B.p = W; g.B.ld = Nin; g.B.R = Kout; g.B.C = Nin;
    g.M = M; g.N = Nin; g.K = Kout; g.C = out; g.c_fp32 = out_fp32; g.ldc = ldo; g.epi = e;
    return b2s_gemm_launch(g, m->dtype, false, true, st);
}
// The second stream is ONE stream per device and process, shared by every model bound there and never destroyed.  HIP maps a stream onto one of a few
// hardware queues (4 by default) when it is created -- the least-loaded one at that moment -- and two streams of one queue do not run beside each
// other: with a fresh second stream per model (and a fresh encoder stream per trainer) the third model of a process landed on the main stream's
// queue and ran the step in 13.3 instead of 6.5 ms (tools/leg_order_lab.py; bench.py --gpus N builds one trainer per leg).  Models of one process
// do not run concurrently in any flow of this package; if they did, the shared stream would order their weight-gradient work, not break it.
int aux_stream_of(hipStream_t* out) {
    static std::mutex mu;
    static std::map<int, hipStream_t> per_device;
    int dev = 0;
    B2S_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    auto it = per_device.find(dev);
    if (it == per_device.end()) {
        hipStream_t s = nullptr;
        B2S_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        it = per_device.emplace(dev, s).first;
    }
    *out = it->second;
    return 0;
}
// ---- aux-stream plumbing (see b2s_model::aux)
int guard_write(const b2s_model* m, const void* buf, hipStream_t st) {       // main stream is about to overwrite `buf`
    auto it = m->aux_readers.find(buf);
    if (it != m->aux_readers.end()) { B2S_HIP(hipStreamWaitEvent(st, it->second, 0)); m->aux_readers.erase(it); }
    return 0;
}
// a launch whose result is only a parameter gradient: with deferred weight gradients it joins the next hand-over to the second stream
// (its operands must stay alive until then: they are operands of the group or live in the context)
int grad_job(const b2s_model* m, hipStream_t st, std::function<int(hipStream_t)> job) {
    if (m->dw_group) { m->aux_jobs.push_back(std::move(job)); return 0; }
    return job(st);
}
int grad_colsum(const b2s_model* m, hipStream_t st, int dtype, const void* X, int x_fp32, int ldx, const float* wgt, float* out, int accumulate,
                int M, int C) {
    return grad_job(m, st, [=](hipStream_t s) { return ro_colsum(dtype, X, x_fp32, ldx, wgt, out, accumulate, M, C, s); });
}
int flush_colsums(const b2s_model* m, hipStream_t st) {
    for (const auto& j : m->aux_jobs) B2S_TRY(j(st));
    m->aux_jobs.clear();
    return 0;
}
int join_aux(const b2s_model* m, hipStream_t st) {                           // main stream waits for every queued dW GEMM
    if (m->aux && m->aux_dirty) {
        hipEvent_t e = m->next_event();
        B2S_HIP(hipEventRecord(e, m->aux));
        B2S_HIP(hipStreamWaitEvent(st, e, 0));
        m->aux_dirty = false;
        m->aux_readers.clear();
    }
    return 0;
}
// weight-gradient GEMMs reduce over all B*L tokens into a small [out,in] matrix: split K over workgroups (fp32 slabs +
// a reduce kernel).  bf16: the 256-row tile kernel runs one workgroup per CU, and these GEMMs share the chip with the
// main stream's kernels (second stream) -- the split is chosen for B2S_DW_BLOCKS workgroups (default: measured best),
// not for all 256 CUs, so that main-stream kernels always find free CUs.
int pick_splitk(int Mo, int No, int K, int dtype) {
    const int bk = dtype ? 64 : 16;
    const int nk = cdiv(K, bk);
    if (dtype) {
        constexpr int target = 256;
        const long tiles = (long)cdiv(Mo, 256) * cdiv(No, 128);
        int s = (int)std::min<long>(std::max<long>(1, target / tiles), std::max(1, nk / 4));
        return std::max(1, std::min(s, 32));
    }
    const long tiles = (long)cdiv(Mo, 128) * cdiv(No, 128);
    int s = (int)std::min<long>(std::max<long>(1, 512 / tiles), std::max(1, nk / 4));
    return std::max(1, std::min(s, 32));
}
// dW[Nout,Kin] = dY[M,Nout]^T * X[M,Kin]
int linear_dw(const b2s_model* m, hipStream_t st, const void* dY, int lddy, const void* X, int ldx, int M, int Nout,
              int Kin, float* dW) {
    GemmArgs g;
    g.A.p = dY; g.A.ld = lddy; g.A.R = M; g.A.C = Nout;
    g.B.p = X; g.B.ld = ldx; g.B.R = M; g.B.C = Kin;
    g.M = Nout; g.N = Kin; g.K = M; g.C = dW; g.c_fp32 = 1; g.ldc = Kin;
    // parameter gradients accumulate (the host zeroes them once per backward pass) -- except, in an overwrite pass (engine.h: dw_ow), the
    // layer weights that one grouped launch writes exactly once: those are stored
    g.epi.accumulate = (m->dw_overwrite_pass && m->dw_group && m->dw_ow_ptrs.count(dW)) ? 0 : 1;
    if (m->dw_group) { m->dw_pending.push_back(g); return 0; }      // launched by end_stage() as part of the stage's group
    g.splitk = pick_splitk(Nout, Kin, M, m->dtype);
    m->set_ws(g, m->aux ? m->aux : st);
    if (!m->aux) return b2s_gemm_launch(g, m->dtype, true, true, st);
    hipEvent_t ready = m->next_event();
    B2S_HIP(hipEventRecord(ready, st));                    // dY (and X) are complete at this point of the main stream
    B2S_HIP(hipStreamWaitEvent(m->aux, ready, 0));
    B2S_TRY(b2s_gemm_launch(g, m->dtype, true, true, m->aux));
    hipEvent_t done = m->next_event();
    B2S_HIP(hipEventRecord(done, m->aux));
    m->aux_readers[dY] = done;
    m->aux_dirty = true;
    return 0;
}

// launch the deferred weight-gradient problems of the current stage as grouped GEMMs on the aux stream
int flush_dw(const b2s_model* m, hipStream_t st) {
    if (m->dw_pending.empty()) return flush_colsums(m, st);
    std::vector<GemmArgs>& q = m->dw_pending;
    std::stable_sort(q.begin(), q.end(), [](const GemmArgs& a, const GemmArgs& b) { return a.K > b.K; });    // long tiles first
    constexpr bool serial = false;       // experiment: groups on the main stream
    if (serial) {
        for (size_t i = 0; i < q.size(); i += B2S_MAX_GROUP)
            B2S_TRY(b2s_gemm_grouped_launch(q.data() + i, (int)std::min<size_t>(B2S_MAX_GROUP, q.size() - i), st));
        q.clear();
        return flush_colsums(m, st);
    }
    hipEvent_t ready = m->next_event();
    B2S_HIP(hipEventRecord(ready, st));
    B2S_HIP(hipStreamWaitEvent(m->aux, ready, 0));
    if (m->side_ev) B2S_HIP(hipStreamWaitEvent(m->aux, m->side_ev, 0));      // (the kv weight gradients read dK / dV written on the side stream)
    // the stage's LayerNorm parameter-gradient reductions ride along: nothing on the main stream needs them before the join
    if (m->ln_jobs.n > 0) { B2S_TRY(ro_ln_param_reduce_batch(m->ln_jobs, m->aux)); m->ln_jobs.n = 0; }
    B2S_TRY(flush_colsums(m, m->aux));
    // a stage with only a few output tiles (prenet: 9, mel / stop heads: 6) would walk the whole token dimension inside
    // each of them (127 K steps, > 100 us on a handful of CUs, and the drain at the end of the entry point waits for it):
    // those go through the split-K launch instead
    long tiles = 0;
    for (const GemmArgs& g : q) tiles += b2s_gemm_glds256_tiles(g);
    if (tiles < 32) {
        for (GemmArgs g : q) {
            B2S_CHECK(g.epi.accumulate, "internal: an overwriting weight gradient reached the split-K path");
            g.splitk = pick_splitk(g.M, g.N, g.K, m->dtype);
            m->set_ws(g, m->aux);
            B2S_TRY(b2s_gemm_launch(g, m->dtype, true, true, m->aux));
        }
    } else {
        // Launches are packed per depth class (q is sorted by K; a class = K within 2x: a decoder layer is 252 tiles over all 8148 tokens + 36
        // tiles over the 1596 memory rows -- launched together the 288 tiles need a second round of deep tiles on 32 CUs, 172 us; apart
        // 129 + 20 us) and, inside a class, first-fit-decreasing by tile count into launches of at most B2S_MAX_GROUP problems and at most
        // ONE ROUND of tiles (256, one workgroup per CU): two stages handed over together used to go out as 8 + 6 problems = 396 + 108 tiles
        // = three rounds of 8148-deep tiles (278 + 124 us) where 252 + 252 is two (2 x 125 us), and heads + layer as 258 tiles (265 us)
        // where 252 + a 6-tile rest is one.
        // tail policy (engine.h: dw_hold_from): at most dw_tail_cap tiles per launch instead -- the second stream then never holds more than
        // that many CUs at a time and the encoder chain on the caller's other stream keeps finding free ones
        // ... for a hand-over that nothing else overlaps (dw_flush_exposed: the drain at the end of a backward entry point -- the fine-tune
        // step's last groups, 7.34 -> 7.24 ms).  Beside the main stream's GEMMs the same packing LOSES (7.43 -> 7.55 ms per step): a
        // 252-tile launch takes every CU for a whole round and the main stream's one-round GEMMs stall behind it, while the CU time
        // (tiles x time per tile) is the same either way -- there the launches stay contiguous runs of up to B2S_MAX_GROUP problems.
        constexpr int round_env = 256;       // (0: never)
        const int round_tiles = m->dw_flush_exposed ? round_env : 0;
        const bool capped = m->dw_flush_capped && m->dw_tail_cap > 0;
#ifdef B2S_LAB
        static const long lab_cap = getenv("B2S_LAB_DW_CAP") ? atol(getenv("B2S_LAB_DW_CAP")) : 0;      // (lab: every hand-over in launches of at most this many tiles)
#else
        constexpr long lab_cap = 0;
#endif
        const long cap = capped ? m->dw_tail_cap : (round_tiles > 0 ? round_tiles : (lab_cap > 0 ? lab_cap : (1L << 30)));
        size_t i = 0;
        while (i < q.size()) {
            size_t ce = i + 1;
            while (ce < q.size() && q[ce].K * 2 > q[i].K) ++ce;
            if (round_tiles > 0)
                std::stable_sort(q.begin() + i, q.begin() + ce, [](const GemmArgs& a, const GemmArgs& b) { return b2s_gemm_glds256_tiles(a) > b2s_gemm_glds256_tiles(b); });
            std::vector<char> used(ce - i, 0);
            for (;;) {
                GemmArgs grp[B2S_MAX_GROUP];
                int n = 0; long t = 0; bool acc_all = true;
                for (size_t k = i; k < ce && n < B2S_MAX_GROUP; ++k) {
                    const long tk = b2s_gemm_glds256_tiles(q[k]);
                    if (used[k - i] || (n > 0 && t + tk > cap)) { if (round_tiles <= 0 && !used[k - i]) break; continue; }     // (old packing: contiguous runs only)
                    grp[n++] = q[k]; t += tk; used[k - i] = 1; acc_all = acc_all && q[k].epi.accumulate;
                }
                if (!n) break;
                // a launch of a few tiles (the rest of a class: the prenet's 256 x 80 dense0 gradient alone was ONE workgroup walking all 8148
                // tokens, 99 us at the end of the fine-tune step) splits K over workgroups instead
                if (t < 32 && acc_all) {
                    for (int k = 0; k < n; ++k) {
                        GemmArgs g = grp[k];
                        g.splitk = pick_splitk(g.M, g.N, g.K, m->dtype);
                        m->set_ws(g, m->aux);
                        B2S_TRY(b2s_gemm_launch(g, m->dtype, true, true, m->aux));
                    }
                } else
                    B2S_TRY(b2s_gemm_grouped_launch(grp, n, m->aux));
            }
            i = ce;
        }
    }
    hipEvent_t done = m->next_event();
    B2S_HIP(hipEventRecord(done, m->aux));
    for (const GemmArgs& g : q) m->aux_readers[g.A.p] = done;
    m->aux_dirty = true;
    m->pending_ev = done;
    q.clear();
    return 0;
}
// end of backward stage `stage`: its parameter gradients are complete once the aux stream has drained.  drain: last stage
// of this C entry point -- the main stream rejoins the aux stream and every outstanding hook fires.
int flush_ln_jobs(const b2s_model* m, hipStream_t st);
// Ordering for the stage hook.  The hook launches a collective on gradients that the second stream completes: the stream the hook
// works on must wait for that.  With a dedicated hook stream the backward's own stream never waits for the second stream here (it
// used to, at every stage -- 0.45 ms per step in data-parallel runs).
// Lab builds only (csrc/build.sh --lab, -DB2S_LAB -> tools/bin/libb2s_hip_lab.so): B2S_LAB_NO_HOOK_ORDER drops these waits -- to show that
// tests/test_gpu_dp_race.py fails without them.  The product library does not contain the switch (tests/test_host_logic.py greps for it).
#ifdef B2S_LAB
static const bool g_lab_no_hook_order = getenv("B2S_LAB_NO_HOOK_ORDER") != nullptr;
#else
constexpr bool g_lab_no_hook_order = false;
#endif
int hook_after_event(const b2s_model* m, hipStream_t st, hipEvent_t ev) {      // ev: second-stream event after the stage's last gradient work
    if (g_lab_no_hook_order) return 0;
    if (ev) B2S_HIP(hipStreamWaitEvent(m->hook_stream ? m->hook_stream : st, ev, 0));
    return 0;
}
int hook_after_stream(const b2s_model* m, hipStream_t st) {                    // the stage's gradients are complete at this point of `st`
    if (g_lab_no_hook_order) return 0;
    if (m->stage_hook && m->hook_stream && m->hook_stream != st) {
        hipEvent_t e = m->next_event();
        B2S_HIP(hipEventRecord(e, st));
        B2S_HIP(hipStreamWaitEvent(m->hook_stream, e, 0));
    }
    return 0;
}
// fire the hooks of the stages in `v` (in order) and empty it
void fire_stages(const b2s_model* m, std::vector<int>& v) {
    for (int s : v) m->stage_done(s);
    v.clear();
}
int end_stage(const b2s_model* m, hipStream_t st, int stage, bool drain, bool force_flush = false) {
    constexpr bool serial = false;
    if (!m->dw_group) {
        B2S_TRY(flush_ln_jobs(m, st));                     // the stage's LayerNorm parameter gradients
        B2S_TRY(join_aux(m, st));
        B2S_TRY(hook_after_stream(m, st));
        fire_stages(m, m->pending_stages);                 // (a deferred postnet stage)
        m->pending_ev = nullptr;
        m->stage_done(stage);
        return 0;
    }
    // The weight-gradient groups -- and with them the LayerNorm parameter reductions and the gradient-only launches -- of TWO stages are
    // handed to the second stream together: every hand-over costs the main stream one event record (~6.5 us of idle queue), and a
    // stage without weight-gradient GEMMs (an output LayerNorm) has nothing to hand over by itself.  The hooks of the stages a
    // hand-over covers fire together at the next one, once the hook's stream has been ordered behind the second stream's event
    // (two decoder-layer stages are one 32 MB bucket of the gradient exchange anyway).
    constexpr int per_flush = 2;
    const bool held = m->dw_hold_from >= 0 && stage >= m->dw_hold_from;          // (tail policy, engine.h)
    if (!serial && !drain && !force_flush && (held || m->dw_pending.empty() || ++m->dw_stages_pending < per_flush)) {
        m->unflushed_stages.push_back(stage);
        return 0;
    }
    m->dw_stages_pending = 0;
    if (serial || m->dw_pending.empty()) B2S_TRY(flush_ln_jobs(m, st));
    hipEvent_t prev_ev = m->pending_ev;
    m->pending_ev = nullptr;
    m->dw_flush_exposed = drain && !m->dw_flush_capped;      // (flush_dw: one round of tiles per launch when nothing overlaps the hand-over)
    const int rc_flush = flush_dw(m, st);                 // sets pending_ev when it launched something
    m->dw_flush_exposed = false;
    B2S_TRY(rc_flush);
    if (m->stage_hook && !m->pending_stages.empty()) {
        if (prev_ev) B2S_TRY(hook_after_event(m, st, prev_ev));        // (the event implies the main stream's part of those stages: the group waited for it)
        else B2S_TRY(hook_after_stream(m, st));                        // stages without second-stream work
    }
    fire_stages(m, m->pending_stages);
    m->pending_stages = m->unflushed_stages;
    m->unflushed_stages.clear();
    m->pending_stages.push_back(stage);
    if (drain) {
        B2S_TRY(join_aux(m, st));
        B2S_TRY(hook_after_stream(m, st));
        fire_stages(m, m->pending_stages);
        m->pending_ev = nullptr;
    }
    return 0;
}

struct AttnScratch { float* S; float* dP; void* dS; };
struct GuidedArgs { float* rows = nullptr; const int* qlen = nullptr; const float* scale = nullptr; float inv2s2 = 0.f; };
// fused attention (attention.hip) unless B2S_ATTN_V1 is set (A/B switch: materialised logits through the GEMM)
bool use_flash(int dh) {
    constexpr bool v1 = false;
    return !v1 && b2s_flash_supported(dh);
}
AttnArgs flash_args(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int B, int H, int Lq, int Lk, int dh,
                    int mask_mode, const int* klen, DropCfg drop, float* lse) {
    AttnArgs a;
    a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk;
    a.scale = 1.f / sqrtf((float)dh); a.mask_mode = mask_mode; a.klen = klen; a.drop = drop; a.lse = lse;
    return a;
}

// softmax(scale * Q K^T + mask) V on head-interleaved rows (attention.py:72-92)
int attn_core_fwd(int dtype, hipStream_t st, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv,
                  void* ctx, int ldc, int B, int H, int Lq, int Lk, int dh, int mask_mode, const int* klen,
                  const float* bias, long bias_sb, long bias_sq, DropCfg drop, float* S, void* P, void* Pd, float* lse = nullptr,
                  const GuidedArgs* ga = nullptr, const int* qskip = nullptr, const int* qoff = nullptr, const int* koff = nullptr) {
    if (lse && !bias && use_flash(dh)) {
        AttnArgs a = flash_args(q, ldq, k, ldk, v, ldv, B, H, Lq, Lk, dh, mask_mode, klen, drop, lse);
        a.out = ctx; a.ldo = ldc; a.qskip = qskip; a.qoff = qoff; a.koff = koff;
        if (ga) { a.ga_rows = ga->rows; a.qlen = ga->qlen; a.ga_inv2s2 = ga->inv2s2; }
        return b2s_flash_fwd(dtype, a, dh, st);
    }
    B2S_CHECK(!ga, "the guided-attention term needs the fused attention kernels (head size 32 / 64 / 96)");
    B2S_CHECK(!qoff && !koff, "internal: ragged rows need the fused attention kernels");
    const int ldp = rup8(Lk);
    GemmArgs g;
    g.A.p = q; g.A.ld = ldq; g.A.R = Lq; g.A.C = dh; g.A.bs_o = (long)Lq * ldq; g.A.bs_i = dh;
    g.B.p = k; g.B.ld = ldk; g.B.R = Lk; g.B.C = dh; g.B.bs_o = (long)Lk * ldk; g.B.bs_i = dh;
    g.M = Lq; g.N = Lk; g.K = dh; g.batch = B * H; g.batch_inner = H;
    g.C = S; g.c_fp32 = 1; g.ldc = ldp; g.cs_o = (long)H * Lq * ldp; g.cs_i = (long)Lq * ldp;
    B2S_TRY(b2s_gemm_launch(g, dtype, false, false, st));
    const float scale = 1.f / sqrtf((float)dh);
    B2S_TRY(ro_softmax_fwd(dtype, S, P, drop.thresh ? Pd : nullptr, B, H, Lq, Lk, ldp, scale, mask_mode, klen, bias,
                           bias_sb, bias_sq, drop, st));
    GemmArgs h;
    h.A.p = drop.thresh ? Pd : P; h.A.ld = ldp; h.A.R = Lq; h.A.C = Lk; h.A.bs_o = (long)H * Lq * ldp; h.A.bs_i = (long)Lq * ldp;
    h.B.p = v; h.B.ld = ldv; h.B.R = Lk; h.B.C = dh; h.B.bs_o = (long)Lk * ldv; h.B.bs_i = dh;
    h.M = Lq; h.N = dh; h.K = Lk; h.batch = B * H; h.batch_inner = H;
    h.C = ctx; h.c_fp32 = 0; h.ldc = ldc; h.cs_o = (long)Lq * ldc; h.cs_i = dh;
    return b2s_gemm_launch(h, dtype, false, true, st);
}
int attn_core_bwd(int dtype, hipStream_t st, const void* dctx, int ldc, const void* q, int ldq, const void* k, int ldk,
                  const void* v, int ldv, const void* P, const void* Pd, void* dq, int lddq, void* dk, int lddk,
                  void* dv, int lddv, int B, int H, int Lq, int Lk, int dh, DropCfg drop, float* dP, void* dS,
                  float* lse = nullptr, const void* O = nullptr, int mask_mode = 0, const int* klen = nullptr,
                  const GuidedArgs* ga = nullptr, const int* qskip = nullptr, const int* qoff = nullptr, const int* koff = nullptr,
                  hipStream_t st_dkv = nullptr, hipEvent_t ev_dq = nullptr) {
    if (lse && use_flash(dh)) {
        AttnArgs a = flash_args(q, ldq, k, ldk, v, ldv, B, H, Lq, Lk, dh, mask_mode, klen, drop, lse);
        a.qskip = qskip; a.qoff = qoff; a.koff = koff;
        a.dout = dctx; a.ldo = ldc; a.dsum = dP; a.dq = dq; a.dk = dk; a.dv = dv; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
        if (ga) { a.ga_rows = ga->rows; a.qlen = ga->qlen; a.ga_scale = ga->scale; a.ga_inv2s2 = ga->inv2s2; }
        return b2s_flash_bwd(dtype, a, dh, O, st, st_dkv, ev_dq);
    }
    B2S_CHECK(!st_dkv, "internal: the side stream needs the fused attention kernels");
    B2S_CHECK(!ga, "the guided-attention term needs the fused attention kernels");
    B2S_CHECK(!qoff && !koff, "internal: ragged rows need the fused attention kernels");
    const int ldp = rup8(Lk);
    const long ps_o = (long)H * Lq * ldp, ps_i = (long)Lq * ldp;
    const void* Pdrop = drop.thresh ? Pd : P;
    {   // dPraw = dctx * V^T
        GemmArgs g;
        g.A.p = dctx; g.A.ld = ldc; g.A.R = Lq; g.A.C = dh; g.A.bs_o = (long)Lq * ldc; g.A.bs_i = dh;
        g.B.p = v; g.B.ld = ldv; g.B.R = Lk; g.B.C = dh; g.B.bs_o = (long)Lk * ldv; g.B.bs_i = dh;
        g.M = Lq; g.N = Lk; g.K = dh; g.batch = B * H; g.batch_inner = H;
        g.C = dP; g.c_fp32 = 1; g.ldc = ldp; g.cs_o = ps_o; g.cs_i = ps_i;
        B2S_TRY(b2s_gemm_launch(g, dtype, false, false, st));
    }
    {   // dV = Pdrop^T * dctx
        GemmArgs g;
        g.A.p = Pdrop; g.A.ld = ldp; g.A.R = Lq; g.A.C = Lk; g.A.bs_o = ps_o; g.A.bs_i = ps_i;
        g.B.p = dctx; g.B.ld = ldc; g.B.R = Lq; g.B.C = dh; g.B.bs_o = (long)Lq * ldc; g.B.bs_i = dh;
        g.M = Lk; g.N = dh; g.K = Lq; g.batch = B * H; g.batch_inner = H;
        g.C = dv; g.c_fp32 = 0; g.ldc = lddv; g.cs_o = (long)Lk * lddv; g.cs_i = dh;
        B2S_TRY(b2s_gemm_launch(g, dtype, true, true, st));
    }
    const float scale = 1.f / sqrtf((float)dh);
    B2S_TRY(ro_softmax_bwd(dtype, P, dP, dS, B, H, Lq, Lk, ldp, scale, drop, st));
    {   // dQ = dS * K
        GemmArgs g;
        g.A.p = dS; g.A.ld = ldp; g.A.R = Lq; g.A.C = Lk; g.A.bs_o = ps_o; g.A.bs_i = ps_i;
        g.B.p = k; g.B.ld = ldk; g.B.R = Lk; g.B.C = dh; g.B.bs_o = (long)Lk * ldk; g.B.bs_i = dh;
        g.M = Lq; g.N = dh; g.K = Lk; g.batch = B * H; g.batch_inner = H;
        g.C = dq; g.c_fp32 = 0; g.ldc = lddq; g.cs_o = (long)Lq * lddq; g.cs_i = dh;
        B2S_TRY(b2s_gemm_launch(g, dtype, false, true, st));
    }
    {   // dK = dS^T * Q
        GemmArgs g;
        g.A.p = dS; g.A.ld = ldp; g.A.R = Lq; g.A.C = Lk; g.A.bs_o = ps_o; g.A.bs_i = ps_i;
        g.B.p = q; g.B.ld = ldq; g.B.R = Lq; g.B.C = dh; g.B.bs_o = (long)Lq * ldq; g.B.bs_i = dh;
        g.M = Lk; g.N = dh; g.K = Lq; g.batch = B * H; g.batch_inner = H;
        g.C = dk; g.c_fp32 = 0; g.ldc = lddk; g.cs_o = (long)Lk * lddk; g.cs_i = dh;
        B2S_TRY(b2s_gemm_launch(g, dtype, true, true, st));
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ planning
struct Scratch {
    // operands of deferred weight-gradient GEMMs stay live until the stage's group has run: every use takes the next
    // buffer of a ring with one buffer per use and layer (ring size 1 = the plain single scratch buffer when nothing is deferred)
    std::vector<void*> r_dyT, r_dz, r_dqkv, r_dkv;
    int i_dyT = 0, i_dz = 0, i_dqkv = 0, i_dkv = 0;
    bool dy_ready = false;       // dyT already holds bf16(dropout(dx)) for the next sublayer (written by the LayerNorm backward)
    static void* rot(const std::vector<void*>& r, int& i) { void* p = r[(size_t)i % r.size()]; ++i; return p; }
    float *S = nullptr, *dP = nullptr; void* dS = nullptr;
    void *dyT = nullptr, *dz = nullptr, *dqkv = nullptr, *dctx = nullptr, *dh = nullptr, *dkv = nullptr;
    float *dx = nullptr, *a3 = nullptr, *dmem = nullptr, *dstop_m = nullptr, *lnws = nullptr;
    // one partial buffer per LayerNorm whose reduction is pending; with deferred weight gradients the reduction runs on the second
    // stream at the end of the stage, so every LayerNorm of the pass has its own (no reuse, no hazard)
    std::vector<float*> r_lnws;
    int i_lnws = 0;
    void *dmelT = nullptr, *doutT = nullptr, *da3 = nullptr, *dz1 = nullptr, *dz2 = nullptr;
    void* dkvcat = nullptr;          // [Mk][L*2D]: dK / dV of every decoder layer (models with a kv_cat weight slab)
    void* dctx_x = nullptr; float* dsum_x = nullptr;     // encoder-decoder attention backward with its dK / dV kernel on the side stream: own d ctx / row-sum buffers (the self-attention that follows on the main stream rewrites dctx / dP at once)
    void* slabs = nullptr;           // fused encoder: [16][M][512] partial sublayer outputs (enc_fused.h)
};

// the fused encoder sublayer kernels serve this call?  (a pure function of the model and S: forward and backward decide alike)
bool enc_use_fused(const b2s_model* m, int S) {
    return m->enc_fused && b2s_encf_supported(m->cfg.encoder_hidden, m->cfg.n_attention_head, 4 * m->cfg.encoder_hidden, S);
}
// Utterances longer than the fused kernels' 128-row tile (multi-byte scripts: BASELINE configs[2] has S = 256): the attention sublayer needs every
// key of its utterance and runs kernel by kernel, but the FFN sublayer and the slab-sum / LayerNorm kernels are ROW-wise -- any cut of the M = B * S
// token rows into B' chunks of S' <= 128 rows is a valid "batch" for them.  S' = the largest divisor of M in [64, 128] (0: none -> unfused FFN too).
int enc_ffn_cut(const b2s_model* m, int B, int S) {
    if (!m->enc_fused || S <= encf::MAXS) return 0;
    const long M = (long)B * S;
    for (int sp = encf::MAXS; sp >= 64; --sp)
        if (M % sp == 0) return sp;
    return 0;
}
// 0: kernel by kernel; 1: both sublayers fused; 2: FFN sublayers fused (rows re-cut), attention sublayers kernel by kernel
int enc_mode(const b2s_model* m, int B, int S) { return enc_use_fused(m, S) ? 1 : (enc_ffn_cut(m, B, S) ? 2 : 0); }
void plan_attn(Arena& a, AttnSave& s, int esz, long M, int D, int B, int H, int Lq, int Lk, bool cross, long Mk,
               bool dropout) {
    s.Lq = Lq; s.Lk = Lk; s.ldp = rup8(Lk);
    s.mean = a.f32(M); s.rstd = a.f32(M);
    s.h = a.T(M * D, esz);
    s.qkv = a.T(M * (cross ? D : 3 * D), esz);
    if (cross) { s.kv = a.T(Mk * 2 * D, esz); s.ldkv = 2 * D; }        // (re-pointed into the ctx-wide K/V buffer by plan_decoder when there is one)
    const long pn = (long)B * H * Lq * s.ldp;
    if (use_flash(D / H)) {
        s.lse = a.f32((long)B * H * Lq);
        s.P = s.Pd = nullptr;
    } else {
        s.P = a.T(pn, esz);
        s.Pd = dropout ? a.T(pn, esz) : s.P;
    }
    s.ctx = a.T(M * D, esz);
}
void plan_ffn(Arena& a, FfnSave& s, int esz, long M, int D) {
    s.mean = a.f32(M); s.rstd = a.f32(M);
    s.h = a.T(M * D, esz);
    s.f = a.T(M * 4 * D, esz);
}

void plan_encoder(const b2s_model* m, b2s_ctx& c, Arena& a, Scratch& sc, std::vector<float*>& xs) {
    const b2s_config& cf = m->cfg;
    const int B = c.B, S = c.S, D = cf.encoder_hidden, H = cf.n_attention_head, esz = m->esz;
    const long M = (long)B * S;
    const bool dr = c.train && cf.transformer_dropout_rate > 0.f;
    c.self_attn.assign(cf.n_encoder_layer, AttnSave());
    c.ffn.assign(cf.n_encoder_layer, FfnSave());
    xs.clear();
    xs.push_back(a.f32(M * D));
    for (int l = 0; l < cf.n_encoder_layer; ++l) {
        c.self_attn[l].x_in = xs.back();
        plan_attn(a, c.self_attn[l], esz, M, D, B, H, S, S, false, 0, dr);
        xs.push_back(a.f32(M * D));
        c.ffn[l].x_in = xs.back();
        plan_ffn(a, c.ffn[l], esz, M, D);
        xs.push_back(a.f32(M * D));
    }
    c.x_final = xs.back();
    c.mean_f = a.f32(M); c.rstd_f = a.f32(M);
    c.memT = a.T(M * m->Dm, esz);
    if (cf.multi_speaker) { c.spk_e = a.f32((long)B * cf.speaker_embedding_size); c.spk_h = a.f32((long)B * cf.speaker_embedding_size); c.spk_dh = a.f32((long)2 * B * cf.speaker_embedding_size); }
    if (cf.multi_lingual) { c.lang_e = a.f32((long)B * cf.language_embedding_size); c.lang_h = a.f32((long)B * cf.language_embedding_size); c.lang_dh = a.f32((long)2 * B * cf.language_embedding_size); }
    // scratch (forward + backward)
    const long pn = (long)B * H * S * rup8(S);
    sc.S = a.f32(pn); sc.dP = a.f32(pn); sc.dS = a.T(pn, esz);
    // Operands of deferred weight-gradient GEMMs must survive until their stage's group has run on the second stream.  One buffer
    // per use and layer (HBM is plentiful): a ring of two stages made the main stream wait on a second-stream event at every reuse
    // -- 44 event waits of 6-13 us per step although the events had long fired.
    const int rings = m->dw_group ? cf.n_encoder_layer : 1;
    for (int i = 0; i < (rings == 1 ? 1 : 2 * rings); ++i) sc.r_dyT.push_back(a.T(M * D, esz));    // 2 uses per layer
    for (int i = 0; i < rings; ++i) { sc.r_dz.push_back(a.T(M * 4 * D, esz)); sc.r_dqkv.push_back(a.T(M * 3 * D, esz)); }
    sc.dyT = sc.r_dyT[0]; sc.dz = sc.r_dz[0]; sc.dqkv = sc.r_dqkv[0];
    sc.dctx = a.T(M * D, esz); sc.dh = a.T(M * D, esz); sc.dx = a.f32(M * D);
    sc.lnws = a.f32((long)RO_LN_WS_ROWS * 2 * D);
    for (int i = 0; i < (m->dw_group ? 2 * cf.n_encoder_layer + 2 : RO_LN_BATCH); ++i) sc.r_lnws.push_back(a.f32((long)RO_LN_WS_ROWS * 2 * D));
    if (enc_mode(m, B, S)) sc.slabs = a.take((size_t)encf::NSF * M * D * (m->enc_slab_bf16 ? 2 : 4));
}

void plan_decoder(const b2s_model* m, b2s_ctx& c, Arena& a, Scratch& sc, std::vector<float*>& xs) {
    const b2s_config& cf = m->cfg;
    const int B = c.B, S = c.S, T = c.T, D = cf.decoder_hidden, H = cf.n_attention_head, esz = m->esz;
    const long M = (long)B * T, Mk = (long)B * S;
    const bool dr = c.train && cf.transformer_dropout_rate > 0.f;
    c.self_attn.assign(cf.n_decoder_layer, AttnSave());
    c.cross_attn.assign(cf.n_decoder_layer, AttnSave());
    c.ffn.assign(cf.n_decoder_layer, FfnSave());
    c.memT = a.T(Mk * D, esz);
    c.tgtT = a.T(M * cf.num_mels, esz);
    c.a1 = a.T(M * cf.prenet_hidden, esz);
    c.a2 = a.T(M * cf.prenet_hidden, esz);
    xs.clear();
    xs.push_back(a.f32(M * D));
    for (int l = 0; l < cf.n_decoder_layer; ++l) {
        c.self_attn[l].x_in = xs.back();
        plan_attn(a, c.self_attn[l], esz, M, D, B, H, T, T, false, 0, dr);
        xs.push_back(a.f32(M * D));
        c.cross_attn[l].x_in = xs.back();
        plan_attn(a, c.cross_attn[l], esz, M, D, B, H, T, S, true, Mk, dr);
        xs.push_back(a.f32(M * D));
        c.ffn[l].x_in = xs.back();
        plan_ffn(a, c.ffn[l], esz, M, D);
        xs.push_back(a.f32(M * D));
    }
    if (m->kv_cat) {
        const int L = cf.n_decoder_layer;
        c.kvcat = a.T(Mk * L * 2 * D, esz);
        for (int l = 0; l < L; ++l) { c.cross_attn[l].kv = (char*)c.kvcat + (size_t)l * 2 * D * esz; c.cross_attn[l].ldkv = L * 2 * D; }
    }
    c.x_final = xs.back();
    c.mean_f = a.f32(M); c.rstd_f = a.f32(M);
    c.outT = a.T(M * D, esz);
    c.rowoff = (int*)a.f32(B + 2); c.melc = a.f32(M * cf.num_mels); c.stopc = a.f32(M);
    if (cf.guided_attention_weight > 0.f) { c.ga_rows = a.f32((long)cf.n_decoder_layer * B * H * T); c.ga_small = a.f32(4); }
    const long pn = (long)B * H * T * rup8(std::max(T, S));
    sc.S = a.f32(pn); sc.dP = a.f32(pn); sc.dS = a.T(pn, esz);
    sc.a3 = a.f32(M * D);
    const int rings = m->dw_group ? cf.n_decoder_layer : 1;                                        // (one buffer per use and layer: see plan_encoder)
    for (int i = 0; i < (rings == 1 ? 1 : 3 * rings); ++i) sc.r_dyT.push_back(a.T(M * D, esz));    // 3 uses per layer
    for (int i = 0; i < (rings == 1 ? 1 : 2 * rings); ++i) sc.r_dqkv.push_back(a.T(M * 3 * D, esz));       // 2 uses per layer
    for (int i = 0; i < rings; ++i) { sc.r_dz.push_back(a.T(M * 4 * D, esz)); sc.r_dkv.push_back(a.T(Mk * 2 * D, esz)); }
    sc.dyT = sc.r_dyT[0]; sc.dz = sc.r_dz[0]; sc.dqkv = sc.r_dqkv[0]; sc.dkv = sc.r_dkv[0];
    sc.dctx = a.T(M * D, esz); sc.dh = a.T(M * D, esz); sc.dx = a.f32(M * D);
    sc.dmelT = a.T(M * cf.num_mels, esz); sc.doutT = a.T(M * D, esz); sc.da3 = a.T(M * D, esz);
    sc.dz1 = a.T(M * cf.prenet_hidden, esz); sc.dz2 = a.T(M * cf.prenet_hidden, esz);
    sc.dstop_m = a.f32(M);
    if (m->kv_cat) sc.dkvcat = a.T(Mk * cf.n_decoder_layer * 2 * D, esz);
    sc.lnws = a.f32((long)RO_LN_WS_ROWS * 2 * D);
    for (int i = 0; i < (m->dw_group ? 3 * cf.n_decoder_layer + 2 : RO_LN_BATCH); ++i) sc.r_lnws.push_back(a.f32((long)RO_LN_WS_ROWS * 2 * D));
    if (m->kv_cat) { sc.dctx_x = a.T(M * D, esz); sc.dsum_x = a.f32((long)B * H * T); }
}

struct PostScratch { std::vector<void*> du, dy; float* stat; int stat_stride; void* col = nullptr; };       // stat: [n layers][2 maxc] column sums (zeroed once per forward)
void plan_postnet(const b2s_model* m, b2s_ctx& c, Arena& a, PostScratch& ps) {
    const b2s_config& cf = m->cfg;
    const long M = (long)c.B * c.T;
    const int n = cf.n_postnet_layer, esz = m->esz;
    c.u.assign(n, nullptr); c.y.assign(n, nullptr); c.bn_mean.assign(n, nullptr); c.bn_rstd.assign(n, nullptr);
    ps.du.assign(n + 1, nullptr);
    for (int i = 0; i < n; ++i) {
        int cin = i == 0 ? cf.num_mels : cf.postnet_hidden;
        int cout = i == n - 1 ? cf.num_mels : cf.postnet_hidden;
        c.u[i] = a.T(M * cin, esz);
        c.y[i] = a.f32(M * cout);
        c.bn_mean[i] = a.f32(cout); c.bn_rstd[i] = a.f32(cout);
        ps.du[i] = a.T(M * cin, esz);                 // gradient w.r.t. u[i]
        // dy (gradient w.r.t. conv output) reuses du slot i+1 sized below
    }
    int maxc = std::max(cf.num_mels, cf.postnet_hidden);
    ps.du[n] = a.T(M * maxc, esz);
    // dy (gradient w.r.t. a conv output, T): one buffer per layer -- the weight-gradient GEMMs that read them run on the second stream
    // after the whole backward pass of the postnet
    ps.dy.assign(n, nullptr);
    for (int i = 0; i < n; ++i) ps.dy[i] = a.T(M * (i == n - 1 ? cf.num_mels : cf.postnet_hidden), esz);
    ps.stat_stride = 2 * maxc;
    ps.stat = a.f32((long)n * ps.stat_stride);
    // bf16: the gathered [tokens, 5 taps x channels] operand of one conv weight gradient, written out (41.7 MB at the default sizes) so
    // that the weight-gradient GEMM reads a plain matrix: with the gather inside the kernel its MFMA waves issue the loads themselves and
    // compute two divisions per 16-byte chunk (60-113 us per layer against 12 + ~40 with the copy); one buffer, reused layer after layer
    // on the second stream
    if (m->dtype == 1) ps.col = a.T(M * 5 * maxc, esz);
}

int check_bound(const b2s_model* m) {
    B2S_CHECK(m && m->bound, "model parameters are not bound (call b2s_model_bind first)");
    return 0;
}

}  // namespace

// ================================================================================================ C ABI: model
extern "C" const char* b2s_last_error(void) { return g_b2s_err; }
extern "C" int b2s_version(void) { return 100; }

extern "C" int b2s_model_create(const b2s_config* cfg, b2s_model** out) {
    B2S_CHECK(cfg && out, "null argument");
    const b2s_config& c = *cfg;
    B2S_CHECK(c.compute_dtype == 0 || c.compute_dtype == 1, "compute_dtype must be 0 (fp32) or 1 (bf16)");
    B2S_CHECK(c.embed_size == c.encoder_hidden, "embed_size (%d) must equal encoder_hidden (%d)", c.embed_size, c.encoder_hidden);
    int Dm = c.encoder_hidden + (c.multi_speaker ? c.speaker_embedding_size : 0) + (c.multi_lingual ? c.language_embedding_size : 0);
    B2S_CHECK(Dm == c.decoder_hidden, "decoder_hidden (%d) must equal encoder_hidden + speaker + language widths (%d) "
              "(transformer/modules.py:86-96 builds decoder layer 0 at the memory width)", c.decoder_hidden, Dm);
    B2S_CHECK(c.encoder_hidden % c.n_attention_head == 0 && c.decoder_hidden % c.n_attention_head == 0,
              "hidden sizes must be divisible by n_attention_head");          // attention.py:40-41
    B2S_CHECK((c.encoder_hidden / c.n_attention_head) % 8 == 0 && (c.decoder_hidden / c.n_attention_head) % 8 == 0,
              "head size must be a multiple of 8");
    B2S_CHECK(c.num_mels % 8 == 0 && c.prenet_hidden % 8 == 0 && c.postnet_hidden % 8 == 0 && c.encoder_hidden % 8 == 0,
              "num_mels / prenet_hidden / postnet_hidden / hidden sizes must be multiples of 8");
    B2S_CHECK(c.encoder_hidden <= 1024 && c.decoder_hidden <= 1024, "hidden sizes above 1024 are not supported");
    B2S_CHECK(c.guided_attention_weight >= 0.f && (c.guided_attention_weight == 0.f || c.guided_attention_sigma > 0.f),
              "guided_attention_weight must be >= 0 and guided_attention_sigma > 0");
    b2s_model* m = new b2s_model();
    m->cfg = c; m->dtype = c.compute_dtype; m->esz = c.compute_dtype ? 2 : 4; m->Dm = Dm;
    {
        static const bool no_fused = getenv("B2S_ENC_FUSED") && atoi(getenv("B2S_ENC_FUSED")) == 0;            // A/B switch: the unfused encoder
        // partial sublayer outputs in bf16 (default): the 8 slabs of a sublayer are written and re-read once each -- 26 MB instead of 52 MB per
        // sublayer and direction; the sum and the residual stream stay fp32 (same rounding point as every other bf16 operand of the step).
        // B2S_ENC_SLAB_BF16=0: fp32 slabs.  Measured (profiles/NOTES_r04.md): 7.76 -> 7.70 ms per step
        constexpr int slab_bf16 = 1;
        m->enc_fused = m->dtype == 1 && !no_fused && c.n_encoder_layer > 0 && c.n_encoder_layer * 4 <= 24 &&
                       b2s_encf_supported(c.encoder_hidden, c.n_attention_head, 4 * c.encoder_hidden, 1);
        m->enc_slab_bf16 = slab_bf16;
        static const bool no_dx16 = getenv("B2S_DX_BF16") && atoi(getenv("B2S_DX_BF16")) == 0;
        auto fast_ln = [](int d) { return d == 512 || d == 768; };
        m->dx_bf16 = m->dtype == 1 && !no_dx16 && true && fast_ln(c.encoder_hidden) && fast_ln(c.decoder_hidden);
    }
    build_layout(m);
    const size_t n = m->tinfo.size();
    m->data.assign(n, nullptr); m->grad.assign(n, nullptr); m->shadow.assign(n, nullptr);
    m->exp_avg.assign(n, nullptr); m->exp_avg_sq.assign(n, nullptr);
    m->conv_wf.assign(c.n_postnet_layer, nullptr); m->conv_wb.assign(c.n_postnet_layer, nullptr);
    *out = m;
    return 0;
}
extern "C" void b2s_model_destroy(b2s_model* m) {
    if (!m) return;
    (void)hipDeviceSynchronize();                 // nothing of this model is in flight any more
    // (m->aux is the process-wide second stream of its device: aux_stream_of() -- never destroyed)
    for (hipEvent_t e : m->ev_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : m->side_evs) (void)hipEventDestroy(e);
    if (m->enc_wT_ev) (void)hipEventDestroy(m->enc_wT_ev);
    for (void* p : m->owned) (void)hipFree(p);
    delete m;
}
extern "C" int b2s_model_num_tensors(const b2s_model* m) { return m ? (int)m->tinfo.size() : -1; }
extern "C" int b2s_model_tensor_info(const b2s_model* m, int i, char* name, int name_cap, int64_t* shape, int* ndim,
                                     int* kind) {
    B2S_CHECK(m && i >= 0 && i < (int)m->tinfo.size(), "tensor index out of range");
    const TensorInfo& t = m->tinfo[i];
    if (name && name_cap > 0) { strncpy(name, t.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (ndim) *ndim = (int)t.shape.size();
    if (shape) for (size_t k = 0; k < t.shape.size(); ++k) shape[k] = t.shape[k];
    if (kind) *kind = t.kind;
    return 0;
}

namespace {
int build_chunks(b2s_model* m, bool l2only, bool with_state, MtChunk** out, int* nout) {
    std::vector<MtChunk> h;
    const int CH = 16384;
    int grp_first[3] = {-1, -1, -1}, last_grp = 0;
    bool grouped = true;                      // parameters come in encoder, decoder, postnet order (state_dict order)
    for (size_t i = 0; i < m->tinfo.size(); ++i) {
        const TensorInfo& t = m->tinfo[i];
        if (t.kind != 1 || (l2only && !t.l2)) continue;
        const int gi = t.name.compare(0, 8, "encoder.") == 0 ? 0 : t.name.compare(0, 8, "postnet.") == 0 ? 2 : 1;
        if (gi < last_grp) grouped = false;
        last_grp = gi;
        if (grp_first[gi] < 0) grp_first[gi] = (int)h.size();
        if (!m->grad[i] && with_state) continue;
        if (with_state && m->cfg.freeze_encoder && t.name.compare(0, 8, "encoder.") == 0) continue;   // frozen: never updated
        for (long o = 0; o < t.numel; o += CH) {
            MtChunk c;
            c.a = (float*)m->data[i] + o; c.b = m->grad[i] ? (float*)m->grad[i] + o : nullptr;
            c.c = with_state ? (float*)m->exp_avg[i] + o : nullptr;
            c.d = with_state ? (float*)m->exp_avg_sq[i] + o : nullptr;
            c.n = (int)std::min<long>(CH, t.numel - o); c.pad = t.l2 ? 1 : 0;
            c.s = (with_state && m->dtype == 1 && t.gemm_weight && m->shadow[i] && m->shadow[i] != m->data[i]) ? (bf16_t*)m->shadow[i] + o : nullptr;
            c.s2 = nullptr; c.cin = 0; c.cout = 0; c.off = o;
            if (with_state && m->dtype == 1 && t.name.compare(0, 20, "postnet.conv_layers.") == 0 && t.shape.size() == 3) {
                const int l = atoi(t.name.c_str() + 20);
                if (l >= 0 && l < m->cfg.n_postnet_layer && m->conv_wf[l] && m->conv_wb[l]) {
                    c.s = (bf16_t*)m->conv_wf[l]; c.s2 = (bf16_t*)m->conv_wb[l]; c.cout = (int)t.shape[0]; c.cin = (int)t.shape[1];
                }
            }
            h.push_back(c);
        }
    }
    MtChunk* d = nullptr;
    if (!h.empty()) {
        B2S_HIP(hipMalloc(&d, h.size() * sizeof(MtChunk)));
        B2S_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(MtChunk), hipMemcpyHostToDevice));
        m->owned.push_back(d);
    }
    if (*out) {                                 // re-bind: the table being replaced (hipFree waits for kernels still reading it)
        auto it = std::find(m->owned.begin(), m->owned.end(), (void*)*out);
        if (it != m->owned.end()) m->owned.erase(it);
        (void)hipFree(*out);
    }
    *out = d; *nout = (int)h.size();
    if (with_state) {
        // chunk ranges per group; a table that is not in group order is treated as one (decoder) group
        const int n = (int)h.size();
        if (!grouped) { m->adam_grp[0] = 0; m->adam_grp[1] = 0; m->adam_grp[2] = n; m->adam_grp[3] = n; }
        else {
            m->adam_grp[3] = n;
            m->adam_grp[2] = grp_first[2] >= 0 ? grp_first[2] : n;
            m->adam_grp[1] = grp_first[1] >= 0 ? grp_first[1] : m->adam_grp[2];
            m->adam_grp[0] = 0;
        }
    }
    return 0;
}
// overwrite mode (engine.h: dw_ow): which weight gradients a grouped launch writes exactly once per backward pass, and the chunk table
// of everything else (cleared by ro_mt_zero)
int build_zero_table(b2s_model* m) {
    const size_t n = m->tinfo.size();
    m->dw_ow.assign(n, 0);
    m->dw_ow_ptrs.clear();
    constexpr bool off = false;
    if (m->dtype == 1 && m->dw_group && !off) {
        auto tiles = [&](const TensorInfo& t) { return (long)cdiv(t.shape[0], 256) * cdiv(t.shape[1], 128); };
        auto layer_key = [](const std::string& nme, std::string& key) {          // "encoder.encoder.<list>.<i>.<leaf>" -> stack + layer index
            for (const char* lst : {".self_attentions.", ".encdec_attentions.", ".ffn_layers."}) {
                const size_t p = nme.find(lst);
                if (p == std::string::npos) continue;
                const size_t q = p + strlen(lst), e = nme.find('.', q);
                key = nme.substr(0, nme.find('.')) + ":" + nme.substr(q, e - q);
                return true;
            }
            return false;
        };
        std::map<std::string, long> stage_tiles;
        std::string key;
        for (size_t i = 0; i < n; ++i)
            if (m->tinfo[i].gemm_weight && m->tinfo[i].shape.size() == 2 && m->grad[i] && layer_key(m->tinfo[i].name, key)) stage_tiles[key] += tiles(m->tinfo[i]);
        for (size_t i = 0; i < n; ++i) {
            const TensorInfo& t = m->tinfo[i];
            if (t.gemm_weight && t.shape.size() == 2 && m->grad[i] && layer_key(t.name, key) && stage_tiles[key] >= 32 &&
                !(m->cfg.freeze_encoder && t.name.compare(0, 8, "encoder.") == 0)) {
                m->dw_ow[i] = 1; m->dw_ow_ptrs.insert((const float*)m->grad[i]);
            }
        }
    }
    std::vector<MtChunk> h;
    const int CH = 16384;
    for (size_t i = 0; i < n; ++i) {
        const TensorInfo& t = m->tinfo[i];
        if (t.kind != 1 || !m->grad[i] || m->dw_ow[i]) continue;
        for (long o = 0; o < t.numel; o += CH) {
            MtChunk c = {};
            c.a = (float*)m->grad[i] + o; c.n = (int)std::min<long>(CH, t.numel - o);
            h.push_back(c);
        }
    }
    if (m->zero_chunks) {
        auto it = std::find(m->owned.begin(), m->owned.end(), (void*)m->zero_chunks);
        if (it != m->owned.end()) m->owned.erase(it);
        (void)hipFree(m->zero_chunks);
        m->zero_chunks = nullptr;
    }
    m->n_zero_chunks = (int)h.size();
    if (!h.empty()) {
        B2S_HIP(hipMalloc(&m->zero_chunks, h.size() * sizeof(MtChunk)));
        B2S_HIP(hipMemcpy(m->zero_chunks, h.data(), h.size() * sizeof(MtChunk), hipMemcpyHostToDevice));
        m->owned.push_back(m->zero_chunks);
    }
    return 0;
}
// one Adam / scatter chunk of tensor i: elements [o, o + cnt) of the tensor (the fields b2s_adam_bind's table carries)
MtChunk state_chunk(const b2s_model* m, size_t i, long o, long cnt) {
    const TensorInfo& t = m->tinfo[i];
    MtChunk c;
    c.a = (float*)m->data[i] + o; c.b = (float*)m->grad[i] + o; c.c = (float*)m->exp_avg[i] + o; c.d = (float*)m->exp_avg_sq[i] + o;
    c.n = (int)cnt; c.pad = t.l2 ? 1 : 0;
    c.s = (m->dtype == 1 && t.gemm_weight && m->shadow[i] && m->shadow[i] != m->data[i]) ? (bf16_t*)m->shadow[i] + o : nullptr;
    c.s2 = nullptr; c.cin = 0; c.cout = 0; c.off = o;
    if (m->dtype == 1 && t.name.compare(0, 20, "postnet.conv_layers.") == 0 && t.shape.size() == 3) {
        const int l = atoi(t.name.c_str() + 20);
        if (l >= 0 && l < m->cfg.n_postnet_layer && m->conv_wf[l] && m->conv_wb[l]) {
            c.s = (bf16_t*)m->conv_wf[l]; c.s2 = (bf16_t*)m->conv_wb[l]; c.cout = (int)t.shape[0]; c.cin = (int)t.shape[1];
        }
    }
    return c;
}
void free_table(b2s_model* m, MtChunk*& p) {
    if (!p) return;
    auto it = std::find(m->owned.begin(), m->owned.end(), (void*)p);
    if (it != m->owned.end()) m->owned.erase(it);
    (void)hipFree(p);
    p = nullptr;
}
// chunk tables of the owned / not-owned parts of every updated parameter (m->shard_ranges: sorted, disjoint element ranges of the flat gradient buffer)
int build_shard_tables(b2s_model* m) {
    free_table(m, m->shard_chunks); free_table(m, m->other_chunks);
    m->n_shard_chunks = m->n_other_chunks = 0;
    if (m->shard_ranges.empty()) return 0;
    const int CH = 16384;
    std::vector<MtChunk> own, oth;
    auto emit = [&](std::vector<MtChunk>& v, size_t i, long o, long e) { for (long x = o; x < e; x += CH) v.push_back(state_chunk(m, i, x, std::min<long>(CH, e - x))); };
    for (size_t i = 0; i < m->tinfo.size(); ++i) {
        const TensorInfo& t = m->tinfo[i];
        if (t.kind != 1 || !m->grad[i]) continue;
        if (m->cfg.freeze_encoder && t.name.compare(0, 8, "encoder.") == 0) continue;
        B2S_CHECK(m->exp_avg[i] && m->exp_avg_sq[i], "b2s_adam_shard: Adam state of %s is not bound", t.name.c_str());
        const long g0 = (const float*)m->grad[i] - m->shard_gbase, g1 = g0 + t.numel;       // the tensor's range in the flat buffer
        long pos = g0;
        for (const auto& r : m->shard_ranges) {
            const long lo = std::max(r.first, g0), hi = std::min(r.second, g1);
            if (lo >= hi) continue;
            if (lo > pos) emit(oth, i, pos - g0, lo - g0);
            emit(own, i, lo - g0, hi - g0);
            pos = hi;
        }
        if (pos < g1) emit(oth, i, pos - g0, g1 - g0);
    }
    auto upload = [&](const std::vector<MtChunk>& h, MtChunk** d, int* n) -> int {
        *n = (int)h.size();
        if (h.empty()) return 0;
        B2S_HIP(hipMalloc(d, h.size() * sizeof(MtChunk)));
        B2S_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(MtChunk), hipMemcpyHostToDevice));
        m->owned.push_back(*d);
        return 0;
    };
    B2S_TRY(upload(own, &m->shard_chunks, &m->n_shard_chunks));
    B2S_TRY(upload(oth, &m->other_chunks, &m->n_other_chunks));
    return 0;
}
int rebuild_adam_chunks(b2s_model* m) {
    const int before = m->n_adam_chunks;
    B2S_TRY(build_chunks(m, false, true, &m->adam_chunks, &m->n_adam_chunks));
    m->l2_fresh = false;
    if (m->n_adam_chunks != before || !m->l2_part) {
        if (m->l2_part) {
            auto it = std::find(m->owned.begin(), m->owned.end(), (void*)m->l2_part);
            if (it != m->owned.end()) m->owned.erase(it);
            (void)hipFree(m->l2_part);
            m->l2_part = nullptr;
        }
        if (m->n_adam_chunks) { B2S_HIP(hipMalloc(&m->l2_part, (size_t)m->n_adam_chunks * sizeof(float))); m->owned.push_back(m->l2_part); }
    }
    if (!m->shard_ranges.empty()) B2S_TRY(build_shard_tables(m));          // (a re-bind replaced parameter pointers: the shard tables follow)
    return 0;
}
// postnet conv weights -> the two GEMM images [Cout][5*Cin] / [Cin][5*Cout] (flipped) in the compute dtype
int relayout_convs(b2s_model* m, hipStream_t st) {
    for (int l = 0; l < m->cfg.n_postnet_layer; ++l) {
        const TensorInfo& t = m->tinfo[m->id("postnet.conv_layers." + std::to_string(l) + ".weight")];
        B2S_TRY(ro_conv_w_relayout(m->dtype, m->P(t.name), m->conv_wf[l], m->conv_wb[l], (int)t.shape[0], (int)t.shape[1], st));
    }
    return 0;
}
}  // namespace

extern "C" int b2s_model_bind(b2s_model* m, void* const* data_host, void* const* grad_host, int n) {
    B2S_CHECK(m && data_host, "null argument");
    B2S_CHECK(n == (int)m->tinfo.size(), "expected %d tensors, got %d", (int)m->tinfo.size(), n);
    for (int i = 0; i < n; ++i) {
        B2S_CHECK(data_host[i], "tensor %s has a null data pointer", m->tinfo[i].name.c_str());
        m->data[i] = data_host[i];
        m->grad[i] = grad_host ? grad_host[i] : nullptr;
    }
    // compute-dtype shadows
    if (m->dtype == 1 && !m->kv_cat && m->cfg.n_decoder_layer > 0 && true) {
        const int L = m->cfg.n_decoder_layer, D = m->cfg.decoder_hidden;
        B2S_HIP(hipMalloc(&m->kv_cat, (size_t)L * 2 * D * D * 2));
        m->owned.push_back(m->kv_cat);
        for (int l = 0; l < L; ++l)
            m->shadow[m->id("decoder.decoder.encdec_attentions." + std::to_string(l) + ".kv_transform.weight")] =
                (char*)m->kv_cat + (size_t)l * 2 * D * D * 2;
    }
    for (int i = 0; i < n; ++i) {
        const TensorInfo& t = m->tinfo[i];
        if (!t.gemm_weight) continue;
        if (m->dtype == 0) { m->shadow[i] = m->data[i]; continue; }
        if (!m->shadow[i] || m->shadow[i] == m->data[i]) {
            void* p = nullptr;
            B2S_HIP(hipMalloc(&p, (size_t)t.numel * 2));
            m->owned.push_back(p); m->shadow[i] = p;
        }
    }
    for (int l = 0; l < m->cfg.n_postnet_layer; ++l) {
        if (m->conv_wf[l]) continue;
        long ne = m->numel("postnet.conv_layers." + std::to_string(l) + ".weight");
        void *a = nullptr, *b = nullptr;
        B2S_HIP(hipMalloc(&a, (size_t)ne * m->esz)); B2S_HIP(hipMalloc(&b, (size_t)ne * m->esz));
        m->owned.push_back(a); m->owned.push_back(b);
        m->conv_wf[l] = a; m->conv_wb[l] = b;
    }
    if (!m->small) { B2S_HIP(hipMalloc(&m->small, 64 * sizeof(float))); m->owned.push_back(m->small); }
    {
        const b2s_config& c = m->cfg;
        if (m->enc_fused && m->enc_wT.empty()) {
            const size_t Dh = c.encoder_hidden, sz[4] = {3 * Dh * Dh, Dh * Dh, 4 * Dh * Dh, 4 * Dh * Dh};
            for (int l = 0; l < c.n_encoder_layer; ++l)
                for (int k = 0; k < 4; ++k) {
                    void* p = nullptr;
                    B2S_HIP(hipMalloc(&p, sz[k] * 2));
                    m->owned.push_back(p); m->enc_wT.push_back(p);
                }
            B2S_HIP(hipEventCreateWithFlags(&m->enc_wT_ev, hipEventDisableTiming));
        }
    }
    if (m->dtype == 1 && !m->sk_ws[0]) {
        // largest split-K user: a postnet conv weight gradient, splitk * Cout * 5 Cin floats (6 x 512 x 2560 = 31 MB at the
        // default sizes); sized generously, HBM is 288 GB
        m->sk_ws_floats = (size_t)24 << 20;
        for (int i = 0; i < 2; ++i)
            if (hipMalloc(&m->sk_ws[i], m->sk_ws_floats * sizeof(float)) == hipSuccess) m->owned.push_back(m->sk_ws[i]);
            else { m->sk_ws[i] = nullptr; (void)hipGetLastError(); }      // (no slab: that stream's split-K launches use atomics)
    }
    if (!m->aux && true) {
        // (a lowest-priority second stream was measured: no change -- a weight-gradient workgroup holds its CU for ~110 us once it
        // has started, whatever the queue priorities say)
        B2S_TRY(aux_stream_of(&m->aux));
        m->ev_pool.resize(256);
        for (auto& ev : m->ev_pool) B2S_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    // measured (MI355X, LJ-shaped step): one grouped launch per backward stage (7 problems, ~290 tiles, full-depth K, no
    // split-K slabs and no reduce kernels) against one split-K launch + slab reduce per weight gradient: 10.42 vs 10.73 ms
    // per step with the producer-wave GEMM (it was 0.1 ms the other way round with the older kernel, whose long tiles
    // left the main stream's kernels fewer idle CUs).  B2S_DW_GROUP=0 restores the per-GEMM launches.
    m->dw_group = m->dtype == 1 && m->aux && !(getenv("B2S_DW_GROUP") && atoi(getenv("B2S_DW_GROUP")) == 0);
    B2S_TRY(ensure_pe(m, 2048));
    m->l2_fresh = false;
    B2S_TRY(build_chunks(m, true, false, &m->l2_chunks, &m->n_l2_chunks));
    if (m->dtype == 1) {                                   // cast table of the bf16 shadows
        std::vector<MtChunk> h;
        const int CH = 16384;
        for (size_t i = 0; i < m->tinfo.size(); ++i) {
            const TensorInfo& t = m->tinfo[i];
            if (!t.gemm_weight || !m->shadow[i] || m->shadow[i] == m->data[i]) continue;
            for (long o = 0; o < t.numel; o += CH) {
                MtChunk c;
                c.a = (float*)m->data[i] + o; c.b = nullptr; c.c = nullptr; c.d = nullptr; c.s = (bf16_t*)m->shadow[i] + o;
                c.n = (int)std::min<long>(CH, t.numel - o); c.pad = 0; c.s2 = nullptr; c.cin = 0; c.cout = 0; c.off = o;
                h.push_back(c);
            }
        }
        if (m->cast_chunks) {
            auto it = std::find(m->owned.begin(), m->owned.end(), (void*)m->cast_chunks);
            if (it != m->owned.end()) m->owned.erase(it);
            (void)hipFree(m->cast_chunks);
            m->cast_chunks = nullptr;
        }
        m->n_cast_chunks = (int)h.size();
        if (!h.empty()) {
            B2S_HIP(hipMalloc(&m->cast_chunks, h.size() * sizeof(MtChunk)));
            B2S_HIP(hipMemcpy(m->cast_chunks, h.data(), h.size() * sizeof(MtChunk), hipMemcpyHostToDevice));
            m->owned.push_back(m->cast_chunks);
        }
    }
    B2S_TRY(build_zero_table(m));
    m->bound = true;
    // A fused optimizer bound earlier (b2s_adam_bind) holds the OLD parameter / gradient pointers in its chunk table: rebuild
    // it against the new ones (the moment buffers belong to the caller and are still the ones it registered), so the step
    // never writes through stale pointers after .to() / load_state_dict(assign=True) replaced a parameter.
    if (m->adam_chunks) B2S_TRY(rebuild_adam_chunks(m));
    return 0;
}

extern "C" int b2s_model_sync_weights(b2s_model* m, void* stream, int shadows_fresh) {
    B2S_TRY(check_bound(m));
    hipStream_t st = S_(stream);
    if (!shadows_fresh) m->l2_fresh = false;
    if (m->dtype && !shadows_fresh) B2S_TRY(ro_mt_cast(m->cast_chunks, m->n_cast_chunks, st));      // every GEMM weight's bf16 shadow, one launch
    // (bf16 mode after a fused optimizer step: the Adam kernel has written both conv weight images itself)
    // (after a fused optimizer step the conv images are current: bf16 -- the Adam kernel writes them itself; fp32 -- the step
    // re-lays them out behind the update, b2s_adam_step_ex)
    if (!(shadows_fresh && m->adam_chunks)) B2S_TRY(relayout_convs(m, st));
    return 0;
}

extern "C" void b2s_ctx_free(b2s_ctx* ctx) { delete ctx; }
// the dropout-site table (drop_sites.h), read-only: what a checker needs to regenerate the mask of any dropout site of the model path
extern "C" int b2s_dropout_site(const char* site, int layer, int decode, uint32_t* op_id_out, int* kind_out, int* salt_out) {
    B2S_CHECK(site && op_id_out, "null argument");
    B2S_CHECK(layer >= 0 && layer < 32, "layer %d out of range (op ids hold 32 layers per segment)", layer);
    B2S_CHECK(!decode || layer < 10, "layer %d out of range (the decode loop's op ids hold 10 layers per site)", layer);
    for (int i = 0; i < DS_COUNT; ++i) {
        if (strcmp(site, g_drop_sites[i].name)) continue;
        B2S_CHECK(!decode || g_drop_sites[i].decode_base > 0, "dropout site %s does not exist in the decode loop", site);
        *op_id_out = decode ? drop_op_decode((DropSite)i, layer) : drop_op((DropSite)i, layer);
        if (kind_out) *kind_out = g_drop_sites[i].kind;
        if (salt_out) *salt_out = decode ? g_drop_sites[i].salt : B2S_SALT_NONE;
        return 0;
    }
    return b2s_fail(__FILE__, __LINE__, "unknown dropout site %s", site);
}
extern "C" void* b2s_model_second_stream(b2s_model* m) { return (m && m->bound) ? (void*)m->aux : nullptr; }
extern "C" int b2s_model_set_side_stream(b2s_model* m, void* stream) {
    B2S_CHECK(m, "null model");
    B2S_CHECK(!stream || S_(stream) != m->aux, "b2s_model_set_side_stream: the library's own second stream cannot be the side stream");
    m->side = S_(stream);
    return 0;
}
extern "C" int b2s_model_set_stage_hook(b2s_model* m, void (*hook)(int, void*), void* user, void* stream) {
    B2S_CHECK(m, "null model");
    m->stage_hook = hook; m->stage_user = user; m->hook_stream = hook ? (hipStream_t)stream : nullptr;
    return 0;
}

// ================================================================================================ encoder
extern "C" size_t b2s_encoder_ws_bytes(const b2s_model* m, int B, int S) {
    if (!m || B <= 0 || S <= 0) return 0;
    b2s_ctx c; c.B = B; c.S = S; c.train = 1;
    Arena a; Scratch sc; std::vector<float*> xs;
    plan_encoder(m, c, a, sc, xs);
    return a.off + 4096;
}

struct EncPlan { Scratch sc; std::vector<float*> xs; };

extern "C" int b2s_encoder_forward(b2s_model* m, const int64_t* inputs, const int32_t* input_lengths,
                                   const int64_t* spk_ids, const float* language_vecs, int B, int S, int train,
                                   uint64_t seed, void* ws, size_t ws_bytes, float* memory_out, void* stream,
                                   b2s_ctx** ctx_out) {
    B2S_TRY(check_bound(m));
    const b2s_config& cf = m->cfg;
    B2S_CHECK(inputs && input_lengths && memory_out && ws, "null argument");
    B2S_CHECK(B > 0 && S > 0, "bad shape B=%d S=%d", B, S);
    B2S_CHECK(!cf.multi_speaker || spk_ids, "input_spk_ids is required (tacotron.py:126-127)");
    B2S_CHECK(!cf.multi_lingual || language_vecs, "input_language_vecs is required (tacotron.py:126-127)");
    B2S_TRY(ensure_pe(m, S));
    hipStream_t st = S_(stream);
    b2s_ctx* c = new b2s_ctx();
    c->kind = 1; c->B = B; c->S = S; c->train = train; c->seed = seed;
    c->ids = inputs; c->in_len = input_lengths; c->spk_ids = spk_ids; c->lang_vecs = language_vecs;
    c->ws = (char*)ws; c->ws_bytes = ws_bytes;
    Arena a; a.base = (char*)ws; a.cap = ws_bytes;
    Scratch sc; std::vector<float*> xs;
    plan_encoder(m, *c, a, sc, xs);
    if (a.overflow || a.off > ws_bytes) { delete c; return b2s_fail(__FILE__, __LINE__, "encoder workspace too small: need %zu bytes, got %zu", a.off, ws_bytes); }
    const int D = cf.encoder_hidden, H = cf.n_attention_head, dh = D / H, dt = m->dtype;
    const long M = (long)B * S;
    const float pt = train ? cf.transformer_dropout_rate : 0.f;
    const std::string p = "encoder.encoder.";
    int rc = 0;
    // speaker / language embedding nets (tacotron.py:36-43): they depend on the ids only and fill columns [D, Dm) of the memory rows, which
    // the encoder stack never touches -- with a second stream they run beside the stack instead of 33 us behind it
    auto embed_nets = [&](hipStream_t s2) -> int {
        const int D = cf.encoder_hidden, Dm = m->Dm;
        int col = D;
        if (cf.multi_speaker) {
            B2S_TRY(ro_spk_embed_fwd((const long*)spk_ids, m->P("encoder.speaker_embed.weight"), m->P("encoder.speaker_layer.weight"),
                                     m->P("encoder.speaker_layer.bias"), c->spk_e, c->spk_h, memory_out, c->memT, dt, Dm, col, B, S,
                                     cf.speaker_embedding_size, s2));
            col += cf.speaker_embedding_size;
        }
        if (cf.multi_lingual)
            B2S_TRY(ro_lang_embed_fwd(language_vecs, cf.max_num_language, m->P("encoder.language_embed.weight"),
                                      m->P("encoder.language_layer.weight"), m->P("encoder.language_layer.bias"), c->lang_e,
                                      c->lang_h, memory_out, c->memT, dt, Dm, col, B, S, cf.language_embedding_size, s2));
        return 0;
    };
    hipEvent_t side_done = nullptr;
    const int emode = cf.n_encoder_layer > 0 ? enc_mode(m, B, S) : 0;
    const bool fused = emode != 0;
    const int fB = emode == 2 ? (int)(M / enc_ffn_cut(m, B, S)) : B, fS = emode == 2 ? enc_ffn_cut(m, B, S) : S;      // the FFN kernels' row cut
    // the fused backward streams transposed copies of this step's encoder weights: written now, beside the forward pass (second stream)
    const bool want_wT = fused && ctx_out && !cf.freeze_encoder;
    auto transposes = [&](hipStream_t s2) -> int {
        EncfTransposeJob jobs[24];
        int n = 0;
        const int Dh = cf.encoder_hidden;
        for (int l = 0; l < cf.n_encoder_layer; ++l) {
            const char* leaf[4] = {"self_attentions", "self_attentions", "ffn_layers", "ffn_layers"};
            const char* w[4] = {"qkv_transform.weight", "output_transform.weight", "input_layer.weight", "output_layer.weight"};
            const int R[4] = {3 * Dh, Dh, 4 * Dh, Dh}, C[4] = {Dh, Dh, Dh, 4 * Dh};
            for (int k = 0; k < 4; ++k)
                jobs[n++] = EncfTransposeJob{(const bf16_t*)m->W(nm(p, leaf[k], l, w[k])), (bf16_t*)m->enc_wT[(size_t)l * 4 + k], R[k], C[k]};
        }
        return b2s_encf_transpose(jobs, n, s2);
    };
    auto run = [&]() -> int {
        const bool side_nets = cf.multi_speaker || cf.multi_lingual;
        if (m->aux && (side_nets || want_wT)) {
            hipEvent_t ready = m->next_event();
            B2S_HIP(hipEventRecord(ready, st));                 // (orders the second stream behind whatever produced the inputs / freed the buffers)
            B2S_HIP(hipStreamWaitEvent(m->aux, ready, 0));
            if (side_nets) {
                B2S_TRY(embed_nets(m->aux));
                side_done = m->next_event();
                B2S_HIP(hipEventRecord(side_done, m->aux));
            }
            if (want_wT) { B2S_TRY(transposes(m->aux)); B2S_HIP(hipEventRecord(m->enc_wT_ev, m->aux)); c->enc_wT_done = true; }
        }
        B2S_TRY(ro_embed_prep_fwd((const long*)inputs, input_lengths, m->P("encoder.embed.weight"), m->pe_enc,
                                  m->P(p + "pe_scale"), xs[0], B, S, D, make_drop(pt, seed, drop_op(DS_ENC_EMBED, 0)), st));
        if (fused) {
            // one kernel per sublayer + one row kernel (slab sum + residual + the NEXT LayerNorm) instead of 7 launches per layer (enc_fused.h)
            const int L = cf.n_encoder_layer, Dm = m->Dm, sb = m->enc_slab_bf16;
            const std::string ln0 = p + "attn_layer_norms.0";
            B2S_TRY(ro_layernorm_fwd(dt, xs[0], m->P(ln0 + ".weight"), m->P(ln0 + ".bias"), c->self_attn[0].h, D, nullptr, 0, c->self_attn[0].mean,
                                     c->self_attn[0].rstd, (int)M, D, 1e-6f, nullptr, 1, st));
#ifdef B2S_LAB
            static const int lab_skip = getenv("B2S_LAB_ENC_SKIP") ? atoi(getenv("B2S_LAB_ENC_SKIP")) : 0;     // measurement aid (lab builds only): what would a free encoder buy?
#else
            constexpr int lab_skip = 0;
#endif
            for (int l = 0; l < L && !lab_skip; ++l) {
                AttnSave& s = c->self_attn[l];
                FfnSave& f = c->ffn[l];
                s.op_attn = drop_op(DS_ENC_ATTN, l); s.op_res = drop_op(DS_ENC_ATTN_RES, l); s.mask_mode = 1;
                f.op_hid = drop_op(DS_ENC_FFN_HID, l); f.op_res = drop_op(DS_ENC_FFN_RES, l);
                const std::string lnf = p + "ffn_layer_norms." + std::to_string(l);
                if (emode == 1) {
                    EncfAttnFwd fa;
                    fa.hN = (const bf16_t*)s.h; fa.Wqkv = (const bf16_t*)m->W(nm(p, "self_attentions", l, "qkv_transform.weight"));
                    fa.Wo = (const bf16_t*)m->W(nm(p, "self_attentions", l, "output_transform.weight")); fa.klen = input_lengths; fa.B = B; fa.S = S;
                    fa.datt = make_drop(pt, seed, s.op_attn); fa.qkv = (bf16_t*)s.qkv; fa.ctx = (bf16_t*)s.ctx; fa.lse = s.lse; fa.slabs = sc.slabs;
                    B2S_TRY(b2s_encf_attn_fwd(fa, sb, st));
                    B2S_TRY(b2s_encf_reduce_ln_fwd(xs[2 * l], sc.slabs, encf::NH, sb, make_drop(pt, seed, s.op_res), m->P(lnf + ".weight"), m->P(lnf + ".bias"),
                                                   xs[2 * l + 1], (bf16_t*)f.h, D, nullptr, 0, f.mean, f.rstd, (int)M, st));
                } else {
                    // attention sublayer kernel by kernel (more than 128 keys per utterance): projection, fused attention, output projection
                    // with the residual in its epilogue, the FFN's LayerNorm
                    B2S_TRY(linear(m, st, s.h, D, m->W(nm(p, "self_attentions", l, "qkv_transform.weight")), (int)M, 3 * D, D, s.qkv, 0, 3 * D, GemmEpilogue()));
                    const char* q = (const char*)s.qkv;
                    B2S_TRY(attn_core_fwd(dt, st, q, 3 * D, q + (size_t)D * m->esz, 3 * D, q + (size_t)2 * D * m->esz, 3 * D, s.ctx, D,
                                          B, H, S, S, dh, 1, input_lengths, nullptr, 0, 0, make_drop(pt, seed, s.op_attn), sc.S, s.P, s.Pd, s.lse));
                    GemmEpilogue e; e.drop = make_drop(pt, seed, s.op_res); e.residual = xs[2 * l]; e.ldr = D;
                    B2S_TRY(linear(m, st, s.ctx, D, m->W(nm(p, "self_attentions", l, "output_transform.weight")), (int)M, D, D, xs[2 * l + 1], 1, D, e));
                    B2S_TRY(ro_layernorm_fwd(dt, xs[2 * l + 1], m->P(lnf + ".weight"), m->P(lnf + ".bias"), f.h, D, nullptr, 0, f.mean, f.rstd, (int)M, D, 1e-6f,
                                             nullptr, 1, st));
                }
                EncfFfn ff;
                ff.X = (const bf16_t*)f.h; ff.Wa = (const bf16_t*)m->W(nm(p, "ffn_layers", l, "input_layer.weight"));
                ff.Wb = (const bf16_t*)m->W(nm(p, "ffn_layers", l, "output_layer.weight")); ff.F = (bf16_t*)f.f; ff.dz = nullptr; ff.slabs = sc.slabs;
                ff.B = fB; ff.S = fS; ff.dhid = make_drop(pt, seed, f.op_hid); ff.aux_scale = 1.f;
                B2S_TRY(b2s_encf_ffn(ff, false, sb, st));
                const bool last = l + 1 == L;
                const std::string lnn = last ? p + "output_layer_norm" : p + "attn_layer_norms." + std::to_string(l + 1);
                B2S_TRY(b2s_encf_reduce_ln_fwd(xs[2 * l + 1], sc.slabs, encf::NSF, sb, make_drop(pt, seed, f.op_res), m->P(lnn + ".weight"), m->P(lnn + ".bias"),
                                               xs[2 * l + 2], (bf16_t*)(last ? c->memT : c->self_attn[l + 1].h), last ? Dm : D, last ? memory_out : nullptr, Dm,
                                               last ? c->mean_f : c->self_attn[l + 1].mean, last ? c->rstd_f : c->self_attn[l + 1].rstd, (int)M, st));
            }
        } else {
        for (int l = 0; l < cf.n_encoder_layer; ++l) {
            AttnSave& s = c->self_attn[l];
            FfnSave& f = c->ffn[l];
            float* x0 = xs[2 * l]; float* x1 = xs[2 * l + 1]; float* x2 = xs[2 * l + 2];
            const std::string lnp = p + "attn_layer_norms." + std::to_string(l);
            B2S_TRY(ro_layernorm_fwd(dt, x0, m->P(lnp + ".weight"), m->P(lnp + ".bias"), s.h, D, nullptr, 0, s.mean, s.rstd,
                                     (int)M, D, 1e-6f, nullptr, 1, st));
            B2S_TRY(linear(m, st, s.h, D, m->W(nm(p, "self_attentions", l, "qkv_transform.weight")), (int)M, 3 * D, D, s.qkv,
                           0, 3 * D, GemmEpilogue()));
            s.op_attn = drop_op(DS_ENC_ATTN, l); s.op_res = drop_op(DS_ENC_ATTN_RES, l);
            const char* q = (const char*)s.qkv;
            B2S_TRY(attn_core_fwd(dt, st, q, 3 * D, q + (size_t)D * m->esz, 3 * D, q + (size_t)2 * D * m->esz, 3 * D, s.ctx, D,
                                  B, H, S, S, dh, 1, input_lengths, nullptr, 0, 0, make_drop(pt, seed, s.op_attn), sc.S, s.P, s.Pd, s.lse));
            s.mask_mode = 1;
            GemmEpilogue e; e.drop = make_drop(pt, seed, s.op_res); e.residual = x0; e.ldr = D;
            B2S_TRY(linear(m, st, s.ctx, D, m->W(nm(p, "self_attentions", l, "output_transform.weight")), (int)M, D, D, x1, 1, D, e));
            const std::string lnf = p + "ffn_layer_norms." + std::to_string(l);
            B2S_TRY(ro_layernorm_fwd(dt, x1, m->P(lnf + ".weight"), m->P(lnf + ".bias"), f.h, D, nullptr, 0, f.mean, f.rstd,
                                     (int)M, D, 1e-6f, nullptr, 1, st));
            f.op_hid = drop_op(DS_ENC_FFN_HID, l); f.op_res = drop_op(DS_ENC_FFN_RES, l);
            GemmEpilogue e1; e1.relu = 1; e1.drop = make_drop(pt, seed, f.op_hid);
            B2S_TRY(linear(m, st, f.h, D, m->W(nm(p, "ffn_layers", l, "input_layer.weight")), (int)M, 4 * D, D, f.f, 0, 4 * D, e1));
            GemmEpilogue e2; e2.drop = make_drop(pt, seed, f.op_res); e2.residual = x1; e2.ldr = D;
            B2S_TRY(linear(m, st, f.f, 4 * D, m->W(nm(p, "ffn_layers", l, "output_layer.weight")), (int)M, D, 4 * D, x2, 1, D, e2));
        }
        const int Dm = m->Dm;
        B2S_TRY(ro_layernorm_fwd(dt, c->x_final, m->P(p + "output_layer_norm.weight"), m->P(p + "output_layer_norm.bias"),
                                 c->memT, Dm, memory_out, Dm, c->mean_f, c->rstd_f, (int)M, D, 1e-6f, nullptr, 1, st));
        }
        if (side_done) B2S_HIP(hipStreamWaitEvent(st, side_done, 0));       // the speaker / language columns (second stream, see above)
        else B2S_TRY(embed_nets(st));
        if (want_wT && !c->enc_wT_done) { B2S_TRY(transposes(st)); B2S_HIP(hipEventRecord(m->enc_wT_ev, st)); c->enc_wT_done = true; }
        c->enc_fused = emode;
        return 0;
    };
    rc = run();
    if (rc || !ctx_out) { delete c; if (ctx_out) *ctx_out = nullptr; return rc; }
    *ctx_out = c;
    return 0;
}

namespace {
// guided attention: scale = coef / sum_b min(T_b, T) * min(N_b, S)          (coef = weight / (layers * heads))
__global__ void k_ga_scale(const int* in_len, const int* tgt_len, int B, int S, int T, float coef, float* small) {
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += 64) acc += (float)min(in_len[b], S) * (float)min(tgt_len[b], T);
    acc = wave_sum(acc);
    if (threadIdx.x == 0) small[0] = coef / fmaxf(acc, 1.f);
}
// small[1] = small[0] * sum(rows)
__global__ __launch_bounds__(1024) void k_ga_reduce(const float* rows, long n, float* small) {
    __shared__ float part[16];
    float acc = 0.f;
    for (long i = threadIdx.x; i < n; i += 1024) acc += rows[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x < 64) {
        float v = threadIdx.x < 16 ? part[threadIdx.x] : 0.f;
        v = wave_sum(v);
        if (threadIdx.x == 0) small[1] = small[0] * v;
    }
}
__global__ void k_ga_bscale(float* small, const float* d_guided) { small[2] = small[0] * (d_guided ? d_guided[0] : 0.f); }
__global__ void k_ga_fetch(const float* small, float* out, float* add_to) { out[0] = small[1]; if (add_to) add_to[0] += small[1]; }
// entry of a sublayer's backward: its dY operand bf16(dropout(dx)) was either produced by the preceding LayerNorm
// backward (fused) or is cast here
int take_dy(b2s_model* m, hipStream_t st, Scratch& sc, long M, int D, const DropCfg& dres, const void** dy) {
    if (sc.dy_ready) { sc.dy_ready = false; *dy = sc.dyT; return 0; }
    *dy = sc.dx;
    sc.dyT = Scratch::rot(sc.r_dyT, sc.i_dyT);
    B2S_TRY(guard_write(m, sc.dyT, st));
    B2S_CHECK(!m->dx_bf16, "internal: a bf16 residual gradient always comes with the next sublayer's dY operand");
    if (m->dtype || dres.thresh) { B2S_TRY(ro_cast_drop(m->dtype, sc.dx, D, sc.dyT, D, (int)M, D, dres, st)); *dy = sc.dyT; }
    return 0;
}
int flush_ln_jobs(const b2s_model* m, hipStream_t st) {
    if (m->ln_jobs.n > 0) { B2S_TRY(ro_ln_param_reduce_batch(m->ln_jobs, st)); m->ln_jobs.n = 0; }
    return 0;
}
// LayerNorm backward at the exit of a sublayer (dx += ...), optionally also emitting the NEXT sublayer's dY operand
int ln_bwd_exit(b2s_model* m, hipStream_t st, Scratch& sc, const void* dh, int dh_fp32, int lddh, const float* x_in,
                const std::string& lnp, const float* mean, const float* rstd, int accumulate, long M, int D,
                const int* row_len, int rpb, const DropCfg* next) {
    void* dy2 = nullptr;
    DropCfg nd = {0, 0, 1.f};
    if (next && m->dtype == 1) {
        sc.dyT = Scratch::rot(sc.r_dyT, sc.i_dyT);
        B2S_TRY(guard_write(m, sc.dyT, st));
        dy2 = sc.dyT; nd = *next;
    }
    B2S_TRY(guard_write(m, sc.dx, st));
    // the parameter-gradient partials of up to RO_LN_BATCH LayerNorms are reduced by one launch (flush_ln_jobs: end of the stage)
    if (m->ln_jobs.n == RO_LN_BATCH) B2S_TRY(flush_ln_jobs(m, st));
    LnReduceJob& jb = m->ln_jobs.j[m->ln_jobs.n];
    jb.ws = m->dw_group ? sc.r_lnws[(size_t)sc.i_lnws++ % sc.r_lnws.size()] : sc.r_lnws[m->ln_jobs.n]; jb.D = D; jb.dgamma = m->G(lnp + ".weight"); jb.dbeta = m->G(lnp + ".bias");
    B2S_TRY(ro_layernorm_bwd(m->dtype, dh, dh_fp32, lddh, x_in, m->P(lnp + ".weight"), mean, rstd, sc.dx, accumulate,
                             jb.dgamma, jb.dbeta, (int)M, D, row_len, rpb, st, const_cast<float*>(jb.ws), dy2, nd, &jb.nblk, m->dx_bf16));
    ++m->ln_jobs.n;
    sc.dy_ready = dy2 != nullptr;
    return 0;
}
// backward of  x_out = x_in + drop(FFN(LN(x_in)))  given dx (in place: dx becomes d x_in)
int ffn_bwd(b2s_model* m, hipStream_t st, const FfnSave& f, Scratch& sc, long M, int D, float p, uint64_t seed,
            const std::string& wp_in, const std::string& wp_out, const std::string& lnp, const DropCfg* next = nullptr) {
    DropCfg dres = make_drop(p, seed, f.op_res), dhid = make_drop(p, seed, f.op_hid);
    const void* dy;
    B2S_TRY(take_dy(m, st, sc, M, D, dres, &dy));
    B2S_TRY(linear_dw(m, st, dy, D, f.f, 4 * D, (int)M, D, 4 * D, m->G(wp_out)));
    GemmEpilogue e; e.relu_aux = f.f; e.ld_aux = 4 * D; e.aux_scale = dhid.scale;
    sc.dz = Scratch::rot(sc.r_dz, sc.i_dz);
    B2S_TRY(guard_write(m, sc.dz, st));
    B2S_TRY(linear_dx(m, st, dy, D, m->W(wp_out), (int)M, 4 * D, D, sc.dz, 0, 4 * D, e));
    B2S_TRY(linear_dw(m, st, sc.dz, 4 * D, f.h, D, (int)M, 4 * D, D, m->G(wp_in)));
    B2S_TRY(linear_dx(m, st, sc.dz, 4 * D, m->W(wp_in), (int)M, D, 4 * D, sc.dh, 0, D, GemmEpilogue()));
    return ln_bwd_exit(m, st, sc, sc.dh, 0, D, f.x_in, lnp, f.mean, f.rstd, 1, M, D, nullptr, 1, next);
}
// backward of x_out = x_in + drop(SelfAttn(LN(x_in)))
int self_attn_bwd(b2s_model* m, hipStream_t st, const AttnSave& s, Scratch& sc, long M, int D, int B, int H, int L,
                  float p, uint64_t seed, const std::string& wq, const std::string& wo, const std::string& lnp,
                  const int* klen = nullptr, const DropCfg* next = nullptr) {
    const int dt = m->dtype, dh = D / H, esz = m->esz;
    DropCfg dres = make_drop(p, seed, s.op_res), datt = make_drop(p, seed, s.op_attn);
    const void* dy;
    B2S_TRY(take_dy(m, st, sc, M, D, dres, &dy));
    B2S_TRY(linear_dw(m, st, dy, D, s.ctx, D, (int)M, D, D, m->G(wo)));
    B2S_TRY(linear_dx(m, st, dy, D, m->W(wo), (int)M, D, D, sc.dctx, 0, D, GemmEpilogue()));
    sc.dqkv = Scratch::rot(sc.r_dqkv, sc.i_dqkv);
    B2S_TRY(guard_write(m, sc.dqkv, st));
    const char* q = (const char*)s.qkv; char* dq = (char*)sc.dqkv;
    B2S_TRY(attn_core_bwd(dt, st, sc.dctx, D, q, 3 * D, q + (size_t)D * esz, 3 * D, q + (size_t)2 * D * esz, 3 * D, s.P, s.Pd,
                          dq, 3 * D, dq + (size_t)D * esz, 3 * D, dq + (size_t)2 * D * esz, 3 * D, B, H, L, L, dh, datt, sc.dP, sc.dS,
                          s.lse, s.ctx, s.mask_mode, klen, nullptr, s.qskip, s.qoff, s.koff));
    B2S_TRY(linear_dw(m, st, sc.dqkv, 3 * D, s.h, D, (int)M, 3 * D, D, m->G(wq)));
    B2S_TRY(linear_dx(m, st, sc.dqkv, 3 * D, m->W(wq), (int)M, D, 3 * D, sc.dh, 0, D, GemmEpilogue()));
    return ln_bwd_exit(m, st, sc, sc.dh, 0, D, s.x_in, lnp, s.mean, s.rstd, 1, M, D, nullptr, 1, next);
}
}  // namespace

extern "C" int b2s_encoder_backward(b2s_model* m, b2s_ctx* c, const float* d_memory, void* stream) {
    B2S_TRY(check_bound(m));
    B2S_CHECK(c && c->kind == 1 && d_memory, "bad encoder context");
    const b2s_config& cf = m->cfg;
    hipStream_t st = S_(stream);
    const int B = c->B, S = c->S, D = cf.encoder_hidden, H = cf.n_attention_head, Dm = m->Dm, dt = m->dtype;
    const long M = (long)B * S;
    const float pt = c->train ? cf.transformer_dropout_rate : 0.f;
    Arena a; a.base = c->ws; a.cap = c->ws_bytes;
    b2s_ctx tmp; tmp.B = B; tmp.S = S; tmp.train = c->train;
    Scratch sc; std::vector<float*> xs;
    plan_encoder(m, tmp, a, sc, xs);
    const std::string p = "encoder.encoder.";
    int col = D;
    // (the speaker / language nets' backward reads d_memory only and produces parameter gradients only: off the main stream)
    if (cf.multi_speaker) {
        const int c0 = col, E = cf.speaker_embedding_size;
        const long* ids = (const long*)c->spk_ids; const float *e = c->spk_e, *h = c->spk_h, *W = m->P("encoder.speaker_layer.weight");
        float *gt = m->G("encoder.speaker_embed.weight"), *gW = m->G("encoder.speaker_layer.weight"), *gb = m->G("encoder.speaker_layer.bias"), *dh = c->spk_dh;
        B2S_TRY(grad_job(m, st, [=](hipStream_t s) { return ro_spk_embed_bwd(d_memory, Dm, c0, ids, e, h, W, gt, gW, gb, dh, B, S, E, s); }));
        col += cf.speaker_embedding_size;
    }
    if (cf.multi_lingual) {
        const int c0 = col, E = cf.language_embedding_size, NL = cf.max_num_language;
        const float *lv = c->lang_vecs, *e = c->lang_e, *h = c->lang_h, *Wl = m->P("encoder.language_embed.weight"), *W = m->P("encoder.language_layer.weight");
        float *gWl = m->G("encoder.language_embed.weight"), *gW = m->G("encoder.language_layer.weight"), *gb = m->G("encoder.language_layer.bias"), *dh = c->lang_dh;
        B2S_TRY(grad_job(m, st, [=](hipStream_t s) { return ro_lang_embed_bwd(d_memory, Dm, c0, lv, NL, e, h, Wl, W, gWl, gW, gb, dh, B, S, E, s); }));
    }
    // every LayerNorm backward below also emits the dY operand (bf16, residual dropout applied) of the sublayer that runs next
    DropCfg nd;
    if (cf.n_encoder_layer > 0) nd = make_drop(pt, c->seed, c->ffn[cf.n_encoder_layer - 1].op_res);
    B2S_TRY(ln_bwd_exit(m, st, sc, d_memory, 1, Dm, c->x_final, p + "output_layer_norm", c->mean_f, c->rstd_f, 0, M, D, nullptr, 1,
                        cf.n_encoder_layer > 0 ? &nd : nullptr));
    B2S_TRY(end_stage(m, st, 3 + cf.n_decoder_layer, false));
    if (c->enc_fused) {
        // (the transposed weight copies of this step: written on the second stream by the forward pass, or here if it did not)
        if (!c->enc_wT_done) {
            EncfTransposeJob jobs[24];
            int n = 0;
            for (int l = 0; l < cf.n_encoder_layer; ++l) {
                const char* leaf[4] = {"self_attentions", "self_attentions", "ffn_layers", "ffn_layers"};
                const char* w[4] = {"qkv_transform.weight", "output_transform.weight", "input_layer.weight", "output_layer.weight"};
                const int R[4] = {3 * D, D, 4 * D, D}, C[4] = {D, D, D, 4 * D};
                for (int k = 0; k < 4; ++k)
                    jobs[n++] = EncfTransposeJob{(const bf16_t*)m->W(nm(p, leaf[k], l, w[k])), (bf16_t*)m->enc_wT[(size_t)l * 4 + k], R[k], C[k]};
            }
            B2S_TRY(b2s_encf_transpose(jobs, n, st));
        } else B2S_HIP(hipStreamWaitEvent(st, m->enc_wT_ev, 0));
        const int sb = m->enc_slab_bf16;
        // slab sum + LayerNorm backward at the exit of a fused sublayer (= ln_bwd_exit with the slab sum as dy)
        auto rl_exit = [&](int ns, const float* x_in, const std::string& lnp, const float* mean, const float* rstd, const DropCfg* next) -> int {
            bf16_t* dy2 = nullptr;
            DropCfg ndc = {0, 0, 1.f};
            if (next) {
                sc.dyT = Scratch::rot(sc.r_dyT, sc.i_dyT);
                B2S_TRY(guard_write(m, sc.dyT, st));
                dy2 = (bf16_t*)sc.dyT; ndc = *next;
            }
            B2S_TRY(guard_write(m, sc.dx, st));
            if (m->ln_jobs.n == RO_LN_BATCH) B2S_TRY(flush_ln_jobs(m, st));
            LnReduceJob& jb = m->ln_jobs.j[m->ln_jobs.n];
            jb.ws = m->dw_group ? sc.r_lnws[(size_t)sc.i_lnws++ % sc.r_lnws.size()] : sc.r_lnws[m->ln_jobs.n];
            jb.D = D; jb.dgamma = m->G(lnp + ".weight"); jb.dbeta = m->G(lnp + ".bias");
            B2S_TRY(b2s_encf_reduce_ln_bwd(sc.slabs, ns, sb, x_in, m->P(lnp + ".weight"), mean, rstd, sc.dx, const_cast<float*>(jb.ws), &jb.nblk, dy2, ndc,
                                           (int)M, st, m->dx_bf16));
            ++m->ln_jobs.n;
            sc.dy_ready = dy2 != nullptr;
            return 0;
        };
#ifdef B2S_LAB
        static const int lab_skip = getenv("B2S_LAB_ENC_SKIP") ? atoi(getenv("B2S_LAB_ENC_SKIP")) : 0;
#else
        constexpr int lab_skip = 0;
#endif
        for (int l = cf.n_encoder_layer - 1; l >= 0; --l) {
            const AttnSave& s = c->self_attn[l];
            const FfnSave& f = c->ffn[l];
            if (lab_skip == 2) continue;                                                     // (no chain, no weight gradients)
            if (lab_skip == 1) {                                                             // (weight-gradient groups only)
                sc.dz = Scratch::rot(sc.r_dz, sc.i_dz); sc.dqkv = Scratch::rot(sc.r_dqkv, sc.i_dqkv);
                const void* dy0 = sc.r_dyT[0];
                B2S_TRY(linear_dw(m, st, dy0, D, f.f, 4 * D, (int)M, D, 4 * D, m->G(nm(p, "ffn_layers", l, "output_layer.weight"))));
                B2S_TRY(linear_dw(m, st, sc.dz, 4 * D, f.h, D, (int)M, 4 * D, D, m->G(nm(p, "ffn_layers", l, "input_layer.weight"))));
                B2S_TRY(linear_dw(m, st, dy0, D, s.ctx, D, (int)M, D, D, m->G(nm(p, "self_attentions", l, "output_transform.weight"))));
                B2S_TRY(linear_dw(m, st, sc.dqkv, 3 * D, s.h, D, (int)M, 3 * D, D, m->G(nm(p, "self_attentions", l, "qkv_transform.weight"))));
                B2S_TRY(end_stage(m, st, 4 + cf.n_decoder_layer + (cf.n_encoder_layer - 1 - l), false));
                continue;
            }
            const std::string lnf = p + "ffn_layer_norms." + std::to_string(l), lna = p + "attn_layer_norms." + std::to_string(l);
            const std::string w1 = nm(p, "ffn_layers", l, "input_layer.weight"), w2 = nm(p, "ffn_layers", l, "output_layer.weight");
            const std::string wq = nm(p, "self_attentions", l, "qkv_transform.weight"), wo = nm(p, "self_attentions", l, "output_transform.weight");
            const bf16_t* const* wT = (const bf16_t* const*)&m->enc_wT[(size_t)l * 4];          // {Wqkv^T, Wo^T, W1^T, W2^T}
            // ---- FFN sublayer
            const void* dy;
            B2S_TRY(take_dy(m, st, sc, M, D, make_drop(pt, c->seed, f.op_res), &dy));
            B2S_TRY(linear_dw(m, st, dy, D, f.f, 4 * D, (int)M, D, 4 * D, m->G(w2)));
            sc.dz = Scratch::rot(sc.r_dz, sc.i_dz);
            B2S_TRY(guard_write(m, sc.dz, st));
            EncfFfn fb;
            fb.X = (const bf16_t*)dy; fb.Wa = wT[3]; fb.Wb = wT[2]; fb.F = (bf16_t*)f.f; fb.dz = (bf16_t*)sc.dz; fb.slabs = sc.slabs;
            fb.B = c->enc_fused == 2 ? (int)(M / enc_ffn_cut(m, B, S)) : B; fb.S = c->enc_fused == 2 ? enc_ffn_cut(m, B, S) : S;
            fb.dhid = DropCfg{0, 0, 1.f}; fb.aux_scale = make_drop(pt, c->seed, f.op_hid).scale;
            B2S_TRY(b2s_encf_ffn(fb, true, sb, st));
            B2S_TRY(linear_dw(m, st, sc.dz, 4 * D, f.h, D, (int)M, 4 * D, D, m->G(w1)));
            nd = make_drop(pt, c->seed, s.op_res);
            B2S_TRY(rl_exit(encf::NSF, f.x_in, lnf, f.mean, f.rstd, &nd));
            // ---- attention sublayer
            if (c->enc_fused == 2) {           // kernel by kernel (the forward's choice for more than 128 keys per utterance)
                if (l > 0) nd = make_drop(pt, c->seed, c->ffn[l - 1].op_res);
                B2S_TRY(self_attn_bwd(m, st, s, sc, M, D, B, H, S, pt, c->seed, wq, wo, lna, c->in_len, l > 0 ? &nd : nullptr));
                B2S_TRY(end_stage(m, st, 4 + cf.n_decoder_layer + (cf.n_encoder_layer - 1 - l), false));
                continue;
            }
            B2S_TRY(take_dy(m, st, sc, M, D, nd, &dy));
            B2S_TRY(linear_dw(m, st, dy, D, s.ctx, D, (int)M, D, D, m->G(wo)));
            sc.dqkv = Scratch::rot(sc.r_dqkv, sc.i_dqkv);
            B2S_TRY(guard_write(m, sc.dqkv, st));
            EncfAttnBwd ab;
            ab.dY = (const bf16_t*)dy; ab.qkv = (const bf16_t*)s.qkv; ab.ctx = (const bf16_t*)s.ctx; ab.lse = s.lse; ab.WoT = wT[1]; ab.WqkvT = wT[0];
            ab.klen = c->in_len; ab.B = B; ab.S = S; ab.datt = make_drop(pt, c->seed, s.op_attn); ab.dqkv = (bf16_t*)sc.dqkv; ab.slabs = sc.slabs;
            B2S_TRY(b2s_encf_attn_bwd(ab, sb, st));
            B2S_TRY(linear_dw(m, st, sc.dqkv, 3 * D, s.h, D, (int)M, 3 * D, D, m->G(wq)));
            if (l > 0) nd = make_drop(pt, c->seed, c->ffn[l - 1].op_res);
            B2S_TRY(rl_exit(encf::NH, s.x_in, lna, s.mean, s.rstd, l > 0 ? &nd : nullptr));
            B2S_TRY(end_stage(m, st, 4 + cf.n_decoder_layer + (cf.n_encoder_layer - 1 - l), false));
        }
    } else
    for (int l = cf.n_encoder_layer - 1; l >= 0; --l) {
        const std::string lnf = p + "ffn_layer_norms." + std::to_string(l), lna = p + "attn_layer_norms." + std::to_string(l);
        nd = make_drop(pt, c->seed, c->self_attn[l].op_res);
        B2S_TRY(ffn_bwd(m, st, c->ffn[l], sc, M, D, pt, c->seed, nm(p, "ffn_layers", l, "input_layer.weight"),
                        nm(p, "ffn_layers", l, "output_layer.weight"), lnf, &nd));
        if (l > 0) nd = make_drop(pt, c->seed, c->ffn[l - 1].op_res);
        B2S_TRY(self_attn_bwd(m, st, c->self_attn[l], sc, M, D, B, H, S, pt, c->seed, nm(p, "self_attentions", l, "qkv_transform.weight"),
                              nm(p, "self_attentions", l, "output_transform.weight"), lna, c->in_len, l > 0 ? &nd : nullptr));
        B2S_TRY(end_stage(m, st, 4 + cf.n_decoder_layer + (cf.n_encoder_layer - 1 - l), false));
    }
    B2S_TRY(ro_embed_prep_bwd(sc.dx, (const long*)c->ids, c->in_len, m->pe_enc, m->G("encoder.embed.weight"), m->G(p + "pe_scale"),
                              B, S, D, make_drop(pt, c->seed, drop_op(DS_ENC_EMBED, 0)), st, m->dx_bf16));
    B2S_TRY(end_stage(m, st, 4 + cf.n_decoder_layer + cf.n_encoder_layer, true));
    return 0;
}

// ================================================================================================ decoder
extern "C" int b2s_decoder_compact_rows(b2s_model* m, const int32_t* target_lengths_host, int B) {
    B2S_CHECK(m && (B == 0 || target_lengths_host) && B >= 0, "bad argument");
    m->ragged_lens.assign(target_lengths_host, target_lengths_host + B);
    return 0;
}

extern "C" size_t b2s_decoder_ws_bytes(const b2s_model* m, int B, int S, int T) {
    if (!m || B <= 0 || S <= 0 || T <= 0) return 0;
    b2s_ctx c; c.B = B; c.S = S; c.T = T; c.train = 1;
    Arena a; Scratch sc; std::vector<float*> xs;
    plan_decoder(m, c, a, sc, xs);
    return a.off + 4096;
}

// memory_ready (hipEvent_t or NULL): `memory` is produced on ANOTHER stream (the encoder forward); this call's stream waits for the event
// right before the first kernel that reads it -- the prenet and the first decoder layer's self-attention do not, and run beside the encoder.
extern "C" int b2s_decoder_forward(b2s_model* m, const float* memory, const int32_t* input_lengths, const float* targets,
                                   const int32_t* target_lengths, int B, int S, int T, int train, uint64_t seed, void* ws,
                                   size_t ws_bytes, float* mels_out, float* stop_out, void* memory_ready, void* stream, b2s_ctx** ctx_out) {
    const bool padded_unobserved = (train & B2S_DEC_PADDED_UNOBSERVED) != 0;
    train &= 1;
    B2S_TRY(check_bound(m));
    const b2s_config& cf = m->cfg;
    B2S_CHECK(memory && input_lengths && targets && target_lengths && mels_out && stop_out && ws, "null argument");
    B2S_CHECK(B > 0 && S > 0 && T > 0, "bad shape B=%d S=%d T=%d", B, S, T);
    B2S_TRY(ensure_pe(m, T));
    hipStream_t st = S_(stream);
    b2s_ctx* c = new b2s_ctx();
    c->kind = 2; c->B = B; c->S = S; c->T = T; c->train = train; c->seed = seed;
    c->in_len = input_lengths; c->tgt_len = target_lengths;
    c->ws = (char*)ws; c->ws_bytes = ws_bytes;
    Arena a; a.base = (char*)ws; a.cap = ws_bytes;
    Scratch sc; std::vector<float*> xs;
    plan_decoder(m, *c, a, sc, xs);
    if (a.overflow || a.off > ws_bytes) { delete c; return b2s_fail(__FILE__, __LINE__, "decoder workspace too small: need %zu bytes, got %zu", a.off, ws_bytes); }
    const int D = cf.decoder_hidden, H = cf.n_attention_head, dh = D / H, dt = m->dtype, esz = m->esz;
    const int NM = cf.num_mels, HP = cf.prenet_hidden;
    const long M = (long)B * T, Mk = (long)B * S;
    const float pt = train ? cf.transformer_dropout_rate : 0.f, pd = train ? cf.decoder_dropout_rate : 0.f;
    const std::string p = "decoder.decoder.";
    // ragged rows (b2s_decoder_compact_rows handed over the host copy of target_lengths; only for callers that declared padded rows unobserved)
    std::vector<int> hoff;
    {
        std::vector<int> lens;
        lens.swap(m->ragged_lens);                        // consumed by this call, whatever happens below
        if (padded_unobserved && (int)lens.size() == B && B <= 64 && use_flash(dh)) {
            hoff.assign(1, 0);
            for (int b = 0; b < B; ++b) {
                if (lens[b] < 1 || lens[b] > T) { delete c; return b2s_fail(__FILE__, __LINE__, "b2s_decoder_compact_rows: target length %d of utterance %d outside [1, %d]", lens[b], b, T); }
                hoff.push_back(hoff.back() + lens[b]);
            }
        }
    }
    c->ragged = !hoff.empty();
    c->Mr = c->ragged ? (long)hoff.back() : M;
    const long Mr = c->Mr;                                  // rows of every row-wise launch below
    const int* roff = c->ragged ? c->rowoff : nullptr;
    auto run = [&]() -> int {
        bool have_memory = false;
        auto need_memory = [&]() -> int {          // first use of the encoder output: cast + the memory K/V projection of every layer
            if (have_memory) return 0;
            have_memory = true;
            if (memory_ready) B2S_HIP(hipStreamWaitEvent(st, (hipEvent_t)memory_ready, 0));
            B2S_TRY(ro_cast(dt, memory, c->memT, Mk * D, st));
            if (c->kvcat)       // memory K/V of every layer in one projection: [Mk, D] x [L*2D, D]^T
                B2S_TRY(linear(m, st, c->memT, D, m->kv_cat, (int)Mk, cf.n_decoder_layer * 2 * D, D, c->kvcat, 0, cf.n_decoder_layer * 2 * D,
                               GemmEpilogue()));
            return 0;
        };
        if (!memory_ready) B2S_TRY(need_memory());          // (single-stream callers keep the round-2 order)
        if (c->ragged) {
            B2S_TRY(ro_set_rowoff(hoff.data(), B + 1, c->rowoff, st));
            B2S_TRY(ro_rows_gather(dt, targets, c->tgtT, roff, B, T, NM, st));
        } else B2S_TRY(ro_cast(dt, targets, c->tgtT, M * NM, st));
        // prenet (tacotron.py:55-65)
        GemmEpilogue e0; e0.bias = m->P("decoder.prenet.dense0.bias"); e0.relu = 1; e0.drop = make_drop(pd, seed, drop_op(DS_DEC_PRENET0, 0));
        B2S_TRY(linear(m, st, c->tgtT, NM, m->W("decoder.prenet.dense0.weight"), (int)Mr, HP, NM, c->a1, 0, HP, e0));
        GemmEpilogue e1; e1.bias = m->P("decoder.prenet.dense1.bias"); e1.relu = 1; e1.drop = make_drop(pd, seed, drop_op(DS_DEC_PRENET1, 0));
        B2S_TRY(linear(m, st, c->a1, HP, m->W("decoder.prenet.dense1.weight"), (int)Mr, HP, HP, c->a2, 0, HP, e1));
        B2S_TRY(linear(m, st, c->a2, HP, m->W("decoder.prenet.dense_final.weight"), (int)Mr, D, HP, sc.a3, 1, D, GemmEpilogue()));
        B2S_TRY(ro_shift_pe_fwd(sc.a3, target_lengths, m->pe_dec, m->P(p + "pe_scale"), xs[0], B, T, D,
                                make_drop(pt, seed, drop_op(DS_DEC_EMBED, 0)), st, roff, (int)Mr));
        const bool guided = cf.guided_attention_weight > 0.f;
        // rows >= target_lengths[b] are padding: the heads mask them (row_len below) and the backward zeroes their gradient, and causal /
        // per-row sub-layers never let a valid row read them -- the attention kernels skip whole 64-row tiles of them (attention.h: qskip)
        // -- only for callers that declare those rows unobserved (B2S_DEC_PADDED_UNOBSERVED): the alignments the reference returns for
        // padded query rows of later layers depend on what earlier layers computed there
        constexpr bool no_qskip = false;
        const int* qskip = (no_qskip || !padded_unobserved) ? nullptr : target_lengths;
        if (guided) {
            hipLaunchKernelGGL(k_ga_scale, dim3(1), dim3(64), 0, st, input_lengths, target_lengths, B, S, T,
                               cf.guided_attention_weight / (float)(cf.n_decoder_layer * H), c->ga_small);
            // ragged rows: the attention kernels write the [B H, T] row sums of the frames that exist only; the reduction below walks all of them
            if (c->ragged) B2S_HIP(hipMemsetAsync(c->ga_rows, 0, (size_t)cf.n_decoder_layer * B * H * T * sizeof(float), st));
        }
        for (int l = 0; l < cf.n_decoder_layer; ++l) {
            AttnSave& s = c->self_attn[l]; AttnSave& x = c->cross_attn[l]; FfnSave& f = c->ffn[l];
            float* x0 = xs[3 * l]; float* x1 = xs[3 * l + 1]; float* x2 = xs[3 * l + 2]; float* x3 = xs[3 * l + 3];
            // causal self-attention
            const std::string lna = p + "attn_layer_norms." + std::to_string(l);
            B2S_TRY(ro_layernorm_fwd(dt, x0, m->P(lna + ".weight"), m->P(lna + ".bias"), s.h, D, nullptr, 0, s.mean, s.rstd, (int)Mr, D,
                                     1e-6f, nullptr, 1, st));
            B2S_TRY(linear(m, st, s.h, D, m->W(nm(p, "self_attentions", l, "qkv_transform.weight")), (int)Mr, 3 * D, D, s.qkv, 0, 3 * D,
                           GemmEpilogue()));
            s.op_attn = drop_op(DS_DEC_SELF_ATTN, l); s.op_res = drop_op(DS_DEC_SELF_RES, l);
            const char* q = (const char*)s.qkv;
            B2S_TRY(attn_core_fwd(dt, st, q, 3 * D, q + (size_t)D * esz, 3 * D, q + (size_t)2 * D * esz, 3 * D, s.ctx, D, B, H, T, T, dh,
                                  2, nullptr, nullptr, 0, 0, make_drop(pt, seed, s.op_attn), sc.S, s.P, s.Pd, s.lse, nullptr, qskip, roff, roff));
            s.mask_mode = 2; s.qskip = qskip; s.qoff = s.koff = roff;
            GemmEpilogue ea; ea.drop = make_drop(pt, seed, s.op_res); ea.residual = x0; ea.ldr = D;
            B2S_TRY(linear(m, st, s.ctx, D, m->W(nm(p, "self_attentions", l, "output_transform.weight")), (int)Mr, D, D, x1, 1, D, ea));
            // encoder-decoder attention
            const std::string lnx = p + "encdec_layer_norms." + std::to_string(l);
            B2S_TRY(ro_layernorm_fwd(dt, x1, m->P(lnx + ".weight"), m->P(lnx + ".bias"), x.h, D, nullptr, 0, x.mean, x.rstd, (int)Mr, D,
                                     1e-6f, nullptr, 1, st));
            B2S_TRY(linear(m, st, x.h, D, m->W(nm(p, "encdec_attentions", l, "q_transform.weight")), (int)Mr, D, D, x.qkv, 0, D, GemmEpilogue()));
            B2S_TRY(need_memory());
            if (!c->kvcat)
                B2S_TRY(linear(m, st, c->memT, D, m->W(nm(p, "encdec_attentions", l, "kv_transform.weight")), (int)Mk, 2 * D, D, x.kv, 0, 2 * D,
                               GemmEpilogue()));
            x.op_attn = drop_op(DS_DEC_CROSS_ATTN, l); x.op_res = drop_op(DS_DEC_CROSS_RES, l);
            const char* kv = (const char*)x.kv;
            GuidedArgs ga;
            if (guided) {
                ga.rows = c->ga_rows + (long)l * B * H * T; ga.qlen = target_lengths;
                ga.inv2s2 = 1.f / (2.f * cf.guided_attention_sigma * cf.guided_attention_sigma);
            }
            B2S_TRY(attn_core_fwd(dt, st, x.qkv, D, kv, x.ldkv, kv + (size_t)D * esz, x.ldkv, x.ctx, D, B, H, T, S, dh, 1, input_lengths,
                                  nullptr, 0, 0, make_drop(pt, seed, x.op_attn), sc.S, x.P, x.Pd, x.lse, guided ? &ga : nullptr, qskip, roff, nullptr));
            x.mask_mode = 1; x.qskip = qskip; x.qoff = roff;
            GemmEpilogue ex; ex.drop = make_drop(pt, seed, x.op_res); ex.residual = x1; ex.ldr = D;
            B2S_TRY(linear(m, st, x.ctx, D, m->W(nm(p, "encdec_attentions", l, "output_transform.weight")), (int)Mr, D, D, x2, 1, D, ex));
            // FFN
            const std::string lnf = p + "ffn_layer_norms." + std::to_string(l);
            B2S_TRY(ro_layernorm_fwd(dt, x2, m->P(lnf + ".weight"), m->P(lnf + ".bias"), f.h, D, nullptr, 0, f.mean, f.rstd, (int)Mr, D,
                                     1e-6f, nullptr, 1, st));
            f.op_hid = drop_op(DS_DEC_FFN_HID, l); f.op_res = drop_op(DS_DEC_FFN_RES, l);
            GemmEpilogue f1; f1.relu = 1; f1.drop = make_drop(pt, seed, f.op_hid);
            B2S_TRY(linear(m, st, f.h, D, m->W(nm(p, "ffn_layers", l, "input_layer.weight")), (int)Mr, 4 * D, D, f.f, 0, 4 * D, f1));
            GemmEpilogue f2; f2.drop = make_drop(pt, seed, f.op_res); f2.residual = x2; f2.ldr = D;
            B2S_TRY(linear(m, st, f.f, 4 * D, m->W(nm(p, "ffn_layers", l, "output_layer.weight")), (int)Mr, D, 4 * D, x3, 1, D, f2));
        }
        B2S_TRY(need_memory());                             // (a model without decoder layers still has to order itself behind the event)
        B2S_TRY(ro_layernorm_fwd(dt, c->x_final, m->P(p + "output_layer_norm.weight"), m->P(p + "output_layer_norm.bias"), c->outT, D,
                                 nullptr, 0, c->mean_f, c->rstd_f, (int)Mr, D, 1e-6f, c->ragged ? nullptr : target_lengths, T, st));
        GemmEpilogue em;
        if (!c->ragged) { em.row_len = target_lengths; em.rows_per_batch = T; }
        B2S_TRY(linear(m, st, c->outT, D, m->W("decoder.mel_net.weight"), (int)Mr, NM, D, c->ragged ? c->melc : mels_out, 1, NM, em));
        if (c->ragged)      // the caller's tensors stay padded [B, T, *], zeros beyond target_lengths (tacotron.py:112-115 impute): stop projection + both scatters in one launch
            B2S_TRY(ro_heads_scatter(dt, c->outT, D, m->P("decoder.stop_net.weight"), m->P("decoder.stop_net.bias"), c->melc, mels_out, stop_out, roff, B, T, NM, D, st));
        else
            B2S_TRY(ro_rowdot_fwd(dt, c->outT, D, m->P("decoder.stop_net.weight"), m->P("decoder.stop_net.bias"), stop_out, (int)Mr, D, target_lengths, T, st));
        if (guided) hipLaunchKernelGGL(k_ga_reduce, dim3(1), dim3(1024), 0, st, c->ga_rows, (long)cf.n_decoder_layer * B * H * T, c->ga_small);
        B2S_LAUNCH_CHECK();
        return 0;
    };
    int rc = run();
    if (rc || !ctx_out) { delete c; if (ctx_out) *ctx_out = nullptr; return rc; }
    *ctx_out = c;
    return 0;
}

namespace {
__global__ void k_rowmask_copy(const float* in, float* out, const int* lens, int T, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { int b = (int)(i / T), t = (int)(i - (long)b * T); out[i] = t < lens[b] ? in[i] : 0.f; }
}
}  // namespace

extern "C" int b2s_decoder_guided_loss(b2s_model* m, b2s_ctx* c, float* out, float* add_to, void* stream) {
    B2S_CHECK(m && c && c->kind == 2 && out, "bad decoder context");
    B2S_CHECK(c->ga_small, "guided_attention_weight is 0: this forward has no guided-attention term");
    hipLaunchKernelGGL(k_ga_fetch, dim3(1), dim3(1), 0, S_(stream), c->ga_small, out, add_to);
    B2S_LAUNCH_CHECK();
    return 0;
}
// dmem_done (hipEvent_t or NULL): recorded on `stream` as soon as d_memory_out is complete -- right after the first decoder layer's
// encoder-decoder attention backward, before that layer's self-attention, the input / prenet backward and their weight gradients.  A caller
// that runs the encoder backward on another stream makes that stream wait for the event only (B2S_DEC_BWD_FLUSH_TAIL must then be set:
// this call hands its last stages' weight-gradient work to the second stream itself instead of leaving it to the next entry point).
extern "C" int b2s_decoder_backward(b2s_model* m, b2s_ctx* c, const float* d_mels, const float* d_stop, const float* d_guided,
                                    int flags, float* d_memory_out, void* dmem_done, void* stream) {
    B2S_TRY(check_bound(m));
    const bool want_dmem = !(flags & B2S_DEC_BWD_NO_DMEMORY);
    B2S_CHECK(c && c->kind == 2 && d_mels && (d_memory_out || !want_dmem), "bad decoder context");
    const b2s_config& cf = m->cfg;
    hipStream_t st = S_(stream);
    const int B = c->B, S = c->S, T = c->T, D = cf.decoder_hidden, H = cf.n_attention_head, dh = D / H, dt = m->dtype, esz = m->esz;
    const int NM = cf.num_mels, HP = cf.prenet_hidden;
    const long Mp = (long)B * T, Mk = (long)B * S;
    const long M = c->ragged ? c->Mr : Mp;                  // rows of every row-wise launch below (ragged rows: sum(target_lengths))
    const int* roff = c->ragged ? c->rowoff : nullptr;
    const float pt = c->train ? cf.transformer_dropout_rate : 0.f, pd = c->train ? cf.decoder_dropout_rate : 0.f;
    Arena a; a.base = c->ws; a.cap = c->ws_bytes;
    b2s_ctx tmp; tmp.B = B; tmp.S = S; tmp.T = T; tmp.train = c->train;
    Scratch sc; std::vector<float*> xs;
    plan_decoder(m, tmp, a, sc, xs);
    const std::string p = "decoder.decoder.";
    const bool guided = c->ga_small != nullptr && d_guided != nullptr;
    if (guided) hipLaunchKernelGGL(k_ga_bscale, dim3(1), dim3(1), 0, st, c->ga_small, d_guided);
    // heads (tacotron.py:112-115)
    if (c->ragged) B2S_TRY(ro_rows_gather(dt, d_mels, sc.dmelT, roff, B, T, NM, st, d_stop, sc.dstop_m, 1));      // (d_stop rides along: one launch)
    else B2S_TRY(ro_cast(dt, d_mels, sc.dmelT, M * NM, st));
    B2S_TRY(linear_dw(m, st, sc.dmelT, NM, c->outT, D, (int)M, NM, D, m->G("decoder.mel_net.weight")));
    GemmEpilogue eo;
    if (!c->ragged) { eo.row_len = c->tgt_len; eo.rows_per_batch = T; }
    B2S_TRY(linear_dx(m, st, sc.dmelT, NM, m->W("decoder.mel_net.weight"), (int)M, D, NM, sc.doutT, 0, D, eo));
    if (d_stop) {
        if (!c->ragged) hipLaunchKernelGGL(k_rowmask_copy, dim3(cdiv(M, 256)), dim3(256), 0, st, d_stop, sc.dstop_m, c->tgt_len, T, M);
        B2S_TRY(grad_colsum(m, st, dt, c->outT, 0, D, sc.dstop_m, m->G("decoder.stop_net.weight"), 1, (int)M, D));
        B2S_TRY(grad_colsum(m, st, 0, sc.dstop_m, 1, 1, nullptr, m->G("decoder.stop_net.bias"), 1, (int)M, 1));
    }
    DropCfg nd;
    if (cf.n_decoder_layer > 0) nd = make_drop(pt, c->seed, c->ffn[cf.n_decoder_layer - 1].op_res);
    B2S_TRY(ln_bwd_exit(m, st, sc, sc.doutT, 0, D, c->x_final, p + "output_layer_norm", c->mean_f, c->rstd_f, 0, M, D, c->ragged ? nullptr : c->tgt_len, T,
                        cf.n_decoder_layer > 0 ? &nd : nullptr));
    B2S_TRY(end_stage(m, st, 1, false));
    // tail policy (engine.h: dw_hold_from): B2S_DW_TAIL_LAYERS decoder layers' (and the prenet's) weight-gradient groups wait for the end
    // of this call and are launched capped at B2S_DW_TAIL_CAP workgroups, beside the encoder backward on the caller's other stream
    // (measured, profiles/NOTES_r03.md: 2 layers / 200 tiles: 8.04 -> 7.88 ms; 1 layer 7.95; 4 layers 7.98; caps <= 176 lose what the holding wins)
#ifdef B2S_LAB
    static const int tail_layers = getenv("B2S_LAB_DW_TAIL_LAYERS") ? atoi(getenv("B2S_LAB_DW_TAIL_LAYERS")) : 1;
    static const int tail_cap = getenv("B2S_LAB_DW_TAIL_CAP") ? atoi(getenv("B2S_LAB_DW_TAIL_CAP")) : 200;
#else
    constexpr int tail_layers = 1;      // (re-measured with the persistent GEMM kernel, round 5: 2 layers 7.30, 1 layer 7.25, none 7.35-7.38 ms on the slower box pair)
    constexpr int tail_cap = 200;
#endif
    m->dw_hold_from = -1; m->dw_tail_cap = 0;
    // (whatever way this call is left -- a failing B2S_TRY included -- the tail policy does not outlive it: a following stand-alone
    // b2s_encoder_backward must not find its stages held)
    struct TailPolicyReset { b2s_model* m; ~TailPolicyReset() { m->dw_hold_from = -1; m->dw_tail_cap = 0; m->dw_flush_capped = false; } } tail_policy_reset{m};
    if ((flags & B2S_DEC_BWD_FLUSH_TAIL) && m->dw_group && tail_layers > 0 && tail_cap > 0 && dt == 1) {
        m->dw_hold_from = 2 + std::max(0, cf.n_decoder_layer - tail_layers);      // stage 1 = heads, 2 + k = decoder layer L-1-k
        m->dw_tail_cap = tail_cap;
    }
    bool first_mem = true, dmem_finished = false;
    // side stream (engine.h): the dK / dV kernels of the encoder-decoder attentions leave this stream
    m->side_ev = nullptr;
    struct SideReset { b2s_model* m; ~SideReset() { m->side_ev = nullptr; } } side_reset{m};
    const bool side = m->side && m->side != st && sc.dkvcat && sc.dctx_x && dt == 1 && use_flash(dh);
    if (side && m->side_evs.empty()) {
        m->side_evs.resize(16);
        for (auto& e : m->side_evs) B2S_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    auto finish_dmem = [&]() -> int {
        if (dmem_finished) return 0;
        dmem_finished = true;
        // every layer's dK / dV; from here on this stream's events cover the side stream's work
        if (m->side_ev) { B2S_HIP(hipStreamWaitEvent(st, m->side_ev, 0)); m->side_ev = nullptr; }
        if (want_dmem && sc.dkvcat && cf.n_decoder_layer > 0) {
            // d(memory) = [dKV_0 .. dKV_{L-1}] [Mk, L*2D] x Wcat [L*2D, D]: one GEMM with K = L*2D instead of L accumulating launches.
            // 56 output tiles only -> K split over 4 workgroups each (slab workspace; free here: with grouped weight gradients
            // nothing on the second stream uses it)
            const int Kc = cf.n_decoder_layer * 2 * D;
            GemmArgs g;
            g.A.p = sc.dkvcat; g.A.ld = Kc; g.A.R = (int)Mk; g.A.C = Kc;
            g.B.p = m->kv_cat; g.B.ld = D; g.B.R = Kc; g.B.C = D;
            g.M = (int)Mk; g.N = D; g.K = Kc; g.C = d_memory_out; g.c_fp32 = 1; g.ldc = D;
            if (m->dw_group) {
                B2S_HIP(hipMemsetAsync(d_memory_out, 0, (size_t)Mk * D * 4, st));
                g.epi.accumulate = 1; g.splitk = 4;
                m->set_ws(g, st);
            }
            B2S_TRY(b2s_gemm_launch(g, dt, false, true, st));
        } else if (first_mem && want_dmem) B2S_HIP(hipMemsetAsync(d_memory_out, 0, (size_t)Mk * D * 4, st));
        if (dmem_done) B2S_HIP(hipEventRecord((hipEvent_t)dmem_done, st));
        return 0;
    };
    for (int l = cf.n_decoder_layer - 1; l >= 0; --l) {
        const std::string lnf = p + "ffn_layer_norms." + std::to_string(l), lnx = p + "encdec_layer_norms." + std::to_string(l),
                          lna = p + "attn_layer_norms." + std::to_string(l);
        nd = make_drop(pt, c->seed, c->cross_attn[l].op_res);
        B2S_TRY(ffn_bwd(m, st, c->ffn[l], sc, M, D, pt, c->seed, nm(p, "ffn_layers", l, "input_layer.weight"),
                        nm(p, "ffn_layers", l, "output_layer.weight"), lnf, &nd));
        {   // encoder-decoder attention backward
            const AttnSave& x = c->cross_attn[l];
            const std::string wq = nm(p, "encdec_attentions", l, "q_transform.weight"), wkv = nm(p, "encdec_attentions", l, "kv_transform.weight"),
                              wo = nm(p, "encdec_attentions", l, "output_transform.weight");
            DropCfg dres = make_drop(pt, c->seed, x.op_res), datt = make_drop(pt, c->seed, x.op_attn);
            const void* dy;
            B2S_TRY(take_dy(m, st, sc, M, D, dres, &dy));
            B2S_TRY(linear_dw(m, st, dy, D, x.ctx, D, (int)M, D, D, m->G(wo)));
            void* const dctx = side ? sc.dctx_x : sc.dctx;
            float* const dsum = side ? sc.dsum_x : sc.dP;
            if (side && m->side_ev) B2S_HIP(hipStreamWaitEvent(st, m->side_ev, 0));      // the layer above has read dctx_x / dsum_x (long done: it ran a whole layer ago)
            B2S_TRY(linear_dx(m, st, dy, D, m->W(wo), (int)M, D, D, dctx, 0, D, GemmEpilogue()));
            sc.dqkv = Scratch::rot(sc.r_dqkv, sc.i_dqkv);
            B2S_TRY(guard_write(m, sc.dqkv, st));
            int lddkv = 2 * D;
            if (sc.dkvcat) { sc.dkv = (char*)sc.dkvcat + (size_t)l * 2 * D * esz; lddkv = cf.n_decoder_layer * 2 * D; }     // own columns per layer: no reuse hazard
            else { sc.dkv = Scratch::rot(sc.r_dkv, sc.i_dkv); B2S_TRY(guard_write(m, sc.dkv, st)); }
            const char* kv = (const char*)x.kv; char* dkv = (char*)sc.dkv;
            GuidedArgs ga;
            if (guided) {
                ga.rows = c->ga_rows + (long)l * B * H * T; ga.qlen = c->tgt_len; ga.scale = c->ga_small + 2;
                ga.inv2s2 = 1.f / (2.f * cf.guided_attention_sigma * cf.guided_attention_sigma);
            }
            B2S_TRY(attn_core_bwd(dt, st, dctx, D, x.qkv, D, kv, x.ldkv, kv + (size_t)D * esz, x.ldkv, x.P, x.Pd, sc.dqkv, D, dkv, lddkv,
                                  dkv + (size_t)D * esz, lddkv, B, H, T, S, dh, datt, dsum, sc.dS, x.lse, x.ctx, x.mask_mode, c->in_len,
                                  guided ? &ga : nullptr, x.qskip, x.qoff, x.koff, side ? m->side : nullptr, side ? m->next_event() : nullptr));
            if (side) {
                hipEvent_t e = m->side_evs[m->side_next++ % m->side_evs.size()];
                B2S_HIP(hipEventRecord(e, m->side));
                m->side_ev = e;
            }
            B2S_TRY(linear_dw(m, st, sc.dqkv, D, x.h, D, (int)M, D, D, m->G(wq)));
            B2S_TRY(linear_dx(m, st, sc.dqkv, D, m->W(wq), (int)M, D, D, sc.dh, 0, D, GemmEpilogue()));
            B2S_TRY(linear_dw(m, st, sc.dkv, lddkv, c->memT, D, (int)Mk, 2 * D, D, m->G(wkv)));
            if (want_dmem && !sc.dkvcat) {
                GemmEpilogue em; em.accumulate = first_mem ? 0 : 1;
                B2S_TRY(linear_dx(m, st, sc.dkv, 2 * D, m->W(wkv), (int)Mk, D, 2 * D, d_memory_out, 1, D, em));
                first_mem = false;
            }
            if (l == 0) B2S_TRY(finish_dmem());              // every layer's dK / dV is in: d(memory) does not wait for the rest of this call
            nd = make_drop(pt, c->seed, c->self_attn[l].op_res);
            B2S_TRY(ln_bwd_exit(m, st, sc, sc.dh, 0, D, x.x_in, lnx, x.mean, x.rstd, 1, M, D, nullptr, 1, &nd));
        }
        if (l > 0) nd = make_drop(pt, c->seed, c->ffn[l - 1].op_res);
        B2S_TRY(self_attn_bwd(m, st, c->self_attn[l], sc, M, D, B, H, T, pt, c->seed, nm(p, "self_attentions", l, "qkv_transform.weight"),
                              nm(p, "self_attentions", l, "output_transform.weight"), lna, nullptr, l > 0 ? &nd : nullptr));
        B2S_TRY(end_stage(m, st, 2 + (cf.n_decoder_layer - 1 - l), false));
    }
    B2S_TRY(finish_dmem());                                  // (no decoder layers)
    B2S_TRY(ro_shift_pe_bwd(dt, sc.dx, c->tgt_len, m->pe_dec, sc.da3, m->G(p + "pe_scale"), B, T, D, make_drop(pt, c->seed, drop_op(DS_DEC_EMBED, 0)), st, m->dx_bf16,
                            roff, (int)M));
    // prenet backward
    DropCfg d1 = make_drop(pd, c->seed, drop_op(DS_DEC_PRENET0, 0)), d2 = make_drop(pd, c->seed, drop_op(DS_DEC_PRENET1, 0));
    B2S_TRY(linear_dw(m, st, sc.da3, D, c->a2, HP, (int)M, D, HP, m->G("decoder.prenet.dense_final.weight")));
    GemmEpilogue e2; e2.relu_aux = c->a2; e2.ld_aux = HP; e2.aux_scale = d2.scale;
    B2S_TRY(linear_dx(m, st, sc.da3, D, m->W("decoder.prenet.dense_final.weight"), (int)M, HP, D, sc.dz2, 0, HP, e2));
    B2S_TRY(grad_colsum(m, st, dt, sc.dz2, 0, HP, nullptr, m->G("decoder.prenet.dense1.bias"), 1, (int)M, HP));
    B2S_TRY(linear_dw(m, st, sc.dz2, HP, c->a1, HP, (int)M, HP, HP, m->G("decoder.prenet.dense1.weight")));
    GemmEpilogue e1; e1.relu_aux = c->a1; e1.ld_aux = HP; e1.aux_scale = d1.scale;
    B2S_TRY(linear_dx(m, st, sc.dz2, HP, m->W("decoder.prenet.dense1.weight"), (int)M, HP, HP, sc.dz1, 0, HP, e1));
    B2S_TRY(grad_colsum(m, st, dt, sc.dz1, 0, HP, nullptr, m->G("decoder.prenet.dense0.bias"), 1, (int)M, HP));
    B2S_TRY(linear_dw(m, st, sc.dz1, HP, c->tgtT, NM, (int)M, HP, NM, m->G("decoder.prenet.dense0.weight")));
    B2S_LAUNCH_CHECK();
    // (deferred join: the caller's next call is b2s_encoder_backward on this stream, whose last stage joins the second stream and fires
    // this stage's hook -- the main stream does not idle here until the prenet's weight-gradient group has finished, ~0.12 ms)
    if (flags & B2S_DEC_BWD_FLUSH_TAIL) {
#ifdef B2S_LAB
        // (timing lab, tools/early_adam_lab.py: the mark in front of the held groups -- the update of the held layer then races with its gradients)
        static const bool lab_early_mark = getenv("B2S_LAB_EARLY_MARK") != nullptr;
        if (lab_early_mark && m->aux) {
            if (!m->grads_mark_ev) B2S_HIP(hipEventCreateWithFlags(&m->grads_mark_ev, hipEventDisableTiming));
            B2S_HIP(hipEventRecord(m->grads_mark_ev, m->aux));
            m->lab_early_marked = true;
        }
#endif
        // hand everything that is still queued to the second stream behind an event of THIS stream (the operands were produced here), but
        // leave the join to the caller's next entry point -- which may run on another stream and must not be the one that orders them
        m->dw_flush_capped = m->dw_hold_from >= 0;
        m->dw_hold_from = -1;
        const int rc = end_stage(m, st, 2 + cf.n_decoder_layer, false, true);
        m->dw_flush_capped = false;
        return rc;
    }
    m->dw_hold_from = -1;
    B2S_TRY(end_stage(m, st, 2 + cf.n_decoder_layer, !(flags & B2S_DEC_BWD_DEFER_JOIN)));
    return 0;
}

extern "C" int b2s_decoder_alignment(b2s_model* m, b2s_ctx* c, int which, int layer, float* out, void* stream) {
    B2S_CHECK(m && c && c->kind == 2 && out, "bad decoder context");
    B2S_CHECK(layer >= 0 && layer < m->cfg.n_decoder_layer && (which == 0 || which == 1), "bad alignment selector");
    const AttnSave& s = which ? c->cross_attn[layer] : c->self_attn[layer];
    if (s.lse) {
        const int D = m->cfg.decoder_hidden, H = m->cfg.n_attention_head, dh = D / H, esz = m->esz;
        AttnArgs a;
        if (which) a = flash_args(s.qkv, D, s.kv, s.ldkv, nullptr, 0, c->B, H, s.Lq, s.Lk, dh, 1, c->in_len, DropCfg{0, 0, 1.f}, s.lse);
        else a = flash_args(s.qkv, 3 * D, (const char*)s.qkv + (size_t)D * esz, 3 * D, nullptr, 0, c->B, H, s.Lq, s.Lk, dh, 2, nullptr, DropCfg{0, 0, 1.f}, s.lse);
        a.qskip = s.qskip; a.qoff = s.qoff; a.koff = s.koff;
        return b2s_flash_align(m->dtype, a, dh, out, S_(stream));
    }
    return ro_align_transpose(m->dtype, s.P, out, c->B * m->cfg.n_attention_head, s.Lq, s.Lk, s.ldp, S_(stream));
}

// ================================================================================================ postnet
extern "C" size_t b2s_postnet_ws_bytes(const b2s_model* m, int B, int T) {
    if (!m || B <= 0 || T <= 0) return 0;
    b2s_ctx c; c.B = B; c.T = T;
    Arena a; PostScratch ps;
    plan_postnet(m, c, a, ps);
    return a.off + 4096;
}

extern "C" int b2s_postnet_forward(b2s_model* m, const float* inputs, const int32_t* lengths, const float* add, int B, int T,
                                   int train, uint64_t seed, void* ws, size_t ws_bytes, float* out, void* stream,
                                   b2s_ctx** ctx_out) {
    B2S_TRY(check_bound(m));
    const b2s_config& cf = m->cfg;
    B2S_CHECK(inputs && lengths && out && ws && B > 0 && T > 0, "bad argument");
    hipStream_t st = S_(stream);
    b2s_ctx* c = new b2s_ctx();
    c->kind = 3; c->B = B; c->T = T; c->train = train; c->seed = seed; c->tgt_len = lengths;
    c->ws = (char*)ws; c->ws_bytes = ws_bytes;
    Arena a; a.base = (char*)ws; a.cap = ws_bytes;
    PostScratch ps;
    plan_postnet(m, *c, a, ps);
    if (a.overflow || a.off > ws_bytes) { delete c; return b2s_fail(__FILE__, __LINE__, "postnet workspace too small: need %zu bytes, got %zu", a.off, ws_bytes); }
    const long M = (long)B * T;
    const int n = cf.n_postnet_layer, dt = m->dtype;
    const float pd = train ? cf.decoder_dropout_rate : 0.f;
    // Training: the batch statistics are column sums taken by the conv GEMM's epilogue (GemmEpilogue::colstat) and turned into mean /
    // rstd / running statistics by the normalisation kernel itself: conv + one row kernel per layer (was conv + memset + two reduction
    // passes + finalize + apply).  B2S_BN_SEPARATE=1 keeps the separate two-pass statistics (A/B switch).
    constexpr bool bn_separate = false;
    const bool fused_stats = train && !bn_separate && M > 1;
    auto run = [&]() -> int {
        B2S_TRY(ro_cast(dt, inputs, c->u[0], M * cf.num_mels, st));
        if (fused_stats) B2S_HIP(hipMemsetAsync(ps.stat, 0, sizeof(float) * (size_t)n * ps.stat_stride, st));
        for (int i = 0; i < n; ++i) {
            const int cin = i == 0 ? cf.num_mels : cf.postnet_hidden, cout = i == n - 1 ? cf.num_mels : cf.postnet_hidden;
            const std::string q = "postnet.batchnorm_layers." + std::to_string(i) + ".";
            float* sums = ps.stat + (size_t)i * ps.stat_stride;
            GemmArgs g;       // y[m, co] = sum_{j,ci} x[m + j - 2, ci] * w[co, ci, j]   (impute + Conv1d k5 p2)
            g.A.p = c->u[i]; g.A.ld = cin; g.A.R = (int)M; g.A.C = 5 * cin; g.A.g_cin = cin; g.A.g_T = T; g.A.g_len = lengths;
            g.B.p = m->conv_wf[i]; g.B.ld = 5 * cin; g.B.R = cout; g.B.C = 5 * cin;
            g.M = (int)M; g.N = cout; g.K = 5 * cin; g.C = c->y[i]; g.c_fp32 = 1; g.ldc = cout;
            if (fused_stats) g.epi.colstat = sums;
            B2S_TRY(b2s_gemm_launch(g, dt, false, false, st));
            DropCfg d = make_drop(pd, seed, drop_op(DS_POST_CONV, i));
            const bool last = i == n - 1;
            if (fused_stats) {
                B2S_TRY(ro_bn_apply_train(dt, c->y[i], sums, c->bn_mean[i], c->bn_rstd[i], 1e-5f, m->P(q + "running_mean"), m->P(q + "running_var"),
                                          (long*)m->data[m->id(q + "num_batches_tracked")], 0.1f, m->P(q + "weight"), m->P(q + "bias"), last ? 0 : 1,
                                          last ? nullptr : c->u[i + 1], last ? out : nullptr, last ? add : nullptr, (int)M, cout, d, st));
                continue;
            }
            if (train)
                B2S_TRY(ro_bn_stats(c->y[i], (int)M, cout, c->bn_mean[i], c->bn_rstd[i], 1e-5f, m->P(q + "running_mean"),
                                    m->P(q + "running_var"), (long*)m->data[m->id(q + "num_batches_tracked")], 0.1f, sums, st));
            else
                B2S_TRY(ro_bn_eval_stats(m->P(q + "running_mean"), m->P(q + "running_var"), c->bn_mean[i], c->bn_rstd[i], 1e-5f, cout, st));
            if (!last)
                B2S_TRY(ro_bn_apply(dt, c->y[i], c->bn_mean[i], c->bn_rstd[i], m->P(q + "weight"), m->P(q + "bias"), 1, c->u[i + 1],
                                    nullptr, nullptr, (int)M, cout, d, st));
            else
                B2S_TRY(ro_bn_apply(dt, c->y[i], c->bn_mean[i], c->bn_rstd[i], m->P(q + "weight"), m->P(q + "bias"), 0, nullptr, out,
                                    add, (int)M, cout, d, st));
        }
        return 0;
    };
    int rc = run();
    if (rc || !ctx_out) { delete c; if (ctx_out) *ctx_out = nullptr; return rc; }
    *ctx_out = c;
    return 0;
}

extern "C" int b2s_postnet_backward(b2s_model* m, b2s_ctx* c, const float* d_out, float* d_inputs_out, int flags, void* stream) {
    B2S_TRY(check_bound(m));
    B2S_CHECK(c && c->kind == 3 && d_out && d_inputs_out, "bad postnet context");
    B2S_CHECK(c->train, "postnet backward requires a train-mode forward (batch statistics)");
    const b2s_config& cf = m->cfg;
    hipStream_t st = S_(stream);
    const int B = c->B, T = c->T, n = cf.n_postnet_layer, dt = m->dtype;
    const long M = (long)B * T;
    const float pd = cf.decoder_dropout_rate;
    Arena a; a.base = c->ws; a.cap = c->ws_bytes;
    b2s_ctx tmp; tmp.B = B; tmp.T = T;
    PostScratch ps;
    plan_postnet(m, tmp, a, ps);
    // The conv weight gradients (5 split-K GEMMs over all tokens + their slab reductions, ~0.3 ms) are not on the path to d_inputs:
    // with a second stream they are queued and run there after the pass, behind one event, while the caller's decoder backward
    // proceeds on the main stream.
    std::vector<GemmArgs> dws;
    for (int i = n - 1; i >= 0; --i) {
        const int cin = i == 0 ? cf.num_mels : cf.postnet_hidden, cout = i == n - 1 ? cf.num_mels : cf.postnet_hidden;
        const std::string q = "postnet.batchnorm_layers." + std::to_string(i) + ".";
        DropCfg d = make_drop(pd, c->seed, drop_op(DS_POST_CONV, i));
        const void* dout = i == n - 1 ? (const void*)d_out : ps.du[i + 1];
        void* dy = ps.dy[i];
        B2S_TRY(ro_bn_bwd(dt, dout, i == n - 1 ? 1 : 0, c->y[i], c->bn_mean[i], c->bn_rstd[i], m->P(q + "weight"), m->P(q + "bias"),
                          i < n - 1 ? 1 : 0, m->G(q + "weight"), m->G(q + "bias"), dy, (int)M, cout, d, st));
        {   // dW[co, ci, j] = sum_m dy[m, co] * xg[m, j*cin + ci]
            GemmArgs g;
            g.A.p = dy; g.A.ld = cout; g.A.R = (int)M; g.A.C = cout;
            g.B.p = c->u[i]; g.B.ld = cin; g.B.R = (int)M; g.B.C = 5 * cin; g.B.g_cin = cin; g.B.g_T = T; g.B.g_len = c->tgt_len;
            g.M = cout; g.N = 5 * cin; g.K = (int)M;
            g.C = m->G("postnet.conv_layers." + std::to_string(i) + ".weight"); g.c_fp32 = 1; g.ldc = 5 * cin;
            g.epi.conv_dw_cin = cin; g.epi.accumulate = 1; g.splitk = pick_splitk(cout, 5 * cin, (int)M, dt);
            if (m->aux) dws.push_back(g);
            else { m->set_ws(g, st); B2S_TRY(b2s_gemm_launch(g, dt, true, true, st)); }
        }
        {   // dx[m, ci] = mask(m) * sum_{j', co} dy[m + j' - 2, co] * w[co, ci, 4 - j']
            GemmArgs g;
            g.A.p = dy; g.A.ld = cout; g.A.R = (int)M; g.A.C = 5 * cout; g.A.g_cin = cout; g.A.g_T = T; g.A.g_len = nullptr;
            g.B.p = m->conv_wb[i]; g.B.ld = 5 * cout; g.B.R = cin; g.B.C = 5 * cout;
            g.M = (int)M; g.N = cin; g.K = 5 * cout;
            g.epi.row_len = c->tgt_len; g.epi.rows_per_batch = T;
            if (i == 0) { g.C = d_inputs_out; g.c_fp32 = 1; } else { g.C = ps.du[i]; g.c_fp32 = 0; }
            g.ldc = cin;
            B2S_TRY(b2s_gemm_launch(g, dt, false, false, st));
        }
    }
    if (!dws.empty()) {
        hipEvent_t ready = m->next_event();
        B2S_HIP(hipEventRecord(ready, st));
        B2S_HIP(hipStreamWaitEvent(m->aux, ready, 0));
        constexpr bool in_kernel_gather = false;          // A/B switch: the previous form
        for (GemmArgs& g : dws) {
            if (ps.col && !in_kernel_gather && g.B.g_cin % 8 == 0) {
                B2S_TRY(ro_im2col5(dt, g.B.p, g.B.g_len, g.B.g_T, g.B.g_cin, ps.col, (long)g.B.R, m->aux));
                g.B.p = ps.col; g.B.ld = 5 * g.B.g_cin; g.B.g_cin = 0; g.B.g_T = 0; g.B.g_len = nullptr;
            }
            m->set_ws(g, m->aux); B2S_TRY(b2s_gemm_launch(g, dt, true, true, m->aux));
        }
        hipEvent_t done = m->next_event();
        B2S_HIP(hipEventRecord(done, m->aux));
        m->aux_dirty = true;
        m->pending_ev = done;                             // (the stage hook, if any, waits for it: end_stage)
    }
    // (deferred join: the caller's next call on this model and stream is b2s_decoder_backward(_ex), whose stages -- or final join --
    // take care of this stage's hook and of the second stream)
    if ((flags & B2S_POST_BWD_DEFER_JOIN) && m->aux) {
        B2S_TRY(flush_ln_jobs(m, st));
        m->pending_stages.push_back(0);                   // fires at the decoder backward's first hand-over (or its join)
        return 0;
    }
    B2S_TRY(join_aux(m, st));
    B2S_TRY(hook_after_stream(m, st));
    fire_stages(m, m->pending_stages);
    m->stage_done(0);
    m->pending_ev = nullptr;
    return 0;
}

// ================================================================================================ loss / optimizer
extern "C" int b2s_loss_forward(b2s_model* m, const float* mel_bef, const float* mel_aft, const float* stop_logits,
                                const float* mel_targets, const int32_t* target_lengths, int B, int T, float* losses_out,
                                float* aft_losses_out, float* scratch, void* stream) {
    B2S_TRY(check_bound(m));
    B2S_CHECK(mel_bef && mel_aft && stop_logits && mel_targets && target_lengths && losses_out && aft_losses_out && scratch, "null argument");
    hipStream_t st = S_(stream);
    float* l2 = scratch;                      // scratch[0] = l2, scratch[1..] partial sums
    const bool fused_zero = m->l2_fresh;
    if (m->l2_fresh) {
        B2S_TRY(ro_sum_scaled(m->l2_part, m->n_adam_chunks, 0.5f * m->cfg.reg_weight, l2, st, scratch + 1, 3 + B));
    } else {
        B2S_HIP(hipMemsetAsync(l2, 0, sizeof(float), st));
        if (m->n_l2_chunks) B2S_TRY(ro_mt_sumsq(m->l2_chunks, m->n_l2_chunks, l2, 0.5f * m->cfg.reg_weight, st));
    }
    return ro_loss_fwd(mel_bef, mel_aft, stop_logits, mel_targets, target_lengths, l2, losses_out, aft_losses_out, B, T,
                       m->cfg.num_mels, 5.0f, scratch + 1, st, fused_zero);
}
extern "C" int b2s_loss_backward(b2s_model* m, const float* mel_bef, const float* mel_aft, const float* stop_logits,
                                 const float* mel_targets, const int32_t* target_lengths, int B, int T, const float* grad_scale,
                                 float* d_bef, float* d_aft, float* d_stop, void* stream) {
    B2S_CHECK(m && mel_bef && mel_aft && stop_logits && mel_targets && target_lengths && d_bef && d_aft && d_stop, "null argument");
    return ro_loss_bwd(mel_bef, mel_aft, stop_logits, mel_targets, target_lengths, grad_scale, d_bef, d_aft, d_stop, B, T,
                       m->cfg.num_mels, 5.0f, S_(stream));
}
extern "C" int b2s_l2_backward(b2s_model* m, const float* grad_scale, void* stream) {
    B2S_TRY(check_bound(m));
    for (size_t i = 0; i < m->tinfo.size(); ++i)
        B2S_CHECK(!m->tinfo[i].l2 || m->grad[i], "gradient of %s is not bound", m->tinfo[i].name.c_str());
    return ro_mt_axpy(m->l2_chunks, m->n_l2_chunks, m->cfg.reg_weight, grad_scale, S_(stream));
}
extern "C" int b2s_adam_bind(b2s_model* m, void* const* exp_avg_host, void* const* exp_avg_sq_host, int n) {
    B2S_TRY(check_bound(m));
    B2S_CHECK(n == (int)m->tinfo.size() && exp_avg_host && exp_avg_sq_host, "expected %d state pointers", (int)m->tinfo.size());
    for (int i = 0; i < n; ++i) {
        m->exp_avg[i] = exp_avg_host[i]; m->exp_avg_sq[i] = exp_avg_sq_host[i];
        B2S_CHECK(m->tinfo[i].kind != 1 || !m->grad[i] || (m->exp_avg[i] && m->exp_avg_sq[i]), "missing Adam state for %s", m->tinfo[i].name.c_str());
    }
    return rebuild_adam_chunks(m);
}
extern "C" int b2s_adam_set_grad_wire(b2s_model* m, const void* wire_bf16, const float* grad_base) {
    B2S_CHECK(m && (!wire_bf16 || grad_base), "null argument");
    B2S_CHECK(!wire_bf16 || (((size_t)wire_bf16 & 15) == 0 && ((size_t)grad_base & 15) == 0), "wire / gradient buffers must be 16-byte aligned");
    m->adam_wire = wire_bf16; m->adam_gbase = wire_bf16 ? grad_base : nullptr;
    return 0;
}
namespace {
// bias corrections of `step` (kernel arguments: rowops.h)
AdamHyper adam_hyper(float lr, int step, float beta1, float beta2) {
    return AdamHyper{lr, (float)(1.0 - std::pow((double)beta1, step)), (float)std::sqrt(1.0 - std::pow((double)beta2, step))};
}
}  // namespace
extern "C" int b2s_adam_step(b2s_model* m, float lr, int step, float beta1, float beta2, float eps, float l2, float grad_scale,
                             void* stream) {
    B2S_TRY(check_bound(m));
    B2S_CHECK(m->adam_chunks && step >= 1, "Adam state not bound or bad step");
    hipStream_t st = S_(stream);
    const AdamHyper dhp = adam_hyper(lr, step, beta1, beta2);
    // l2 is applied to the L2 member set only (chunk flag), i.e. g = grad*grad_scale + l2*p for members
    // the partial sums cover the L2 regulariser only if every member is in the chunk table (not with a frozen encoder)
    const bool cover = !m->cfg.freeze_encoder;
    if (!m->shard_ranges.empty()) {
        // sharded (b2s_adam_shard): this rank updates the element ranges it owns; the caller then packs them into the parameter wire, all-gathers
        // it and scatters the other ranks' updates back (b2s_param_wire).  The regulariser's sum of squares is recomputed from the complete
        // parameters by the next loss call (l2_fresh = false): this rank's Adam pass sees only its shard.
        B2S_TRY(ro_mt_adam(m->shard_chunks, m->n_shard_chunks, dhp, beta1, beta2, eps, l2, grad_scale, nullptr, st, m->adam_wire, m->adam_gbase));
        m->adam_step_no = step; m->adam_step_mask = 7;
        m->l2_fresh = false;
        return 0;
    }
    B2S_TRY(ro_mt_adam(m->adam_chunks, m->n_adam_chunks, dhp, beta1, beta2, eps, l2, grad_scale, cover ? m->l2_part : nullptr, st, m->adam_wire, m->adam_gbase));
    // fp32 (parity) mode: the Adam kernel does not write the conv GEMM images (in bf16 mode it does); refresh them here so that
    // an eval / synthesis forward between two training steps sees the weights the step just produced
    if (m->dtype == 0) B2S_TRY(relayout_convs(m, st));
    m->adam_step_no = step; m->adam_step_mask = 7;
    m->l2_fresh = cover;
    return 0;
}
extern "C" int b2s_adam_step_groups(b2s_model* m, float lr, int step, float beta1, float beta2, float eps, float l2, float grad_scale,
                                    int groups, int behind_mark, void* stream) {
    B2S_TRY(check_bound(m));
    B2S_CHECK(m->adam_chunks && step >= 1 && groups > 0 && groups < 8 && (behind_mark == 0 || behind_mark == 1), "Adam state not bound, bad step, group mask or placement");
    B2S_CHECK(m->shard_ranges.empty(), "b2s_adam_step_groups: the optimizer is sharded (b2s_adam_shard): one b2s_adam_step per step");
    if (step != m->adam_step_no) { m->adam_step_no = step; m->adam_step_mask = 0; }
    B2S_CHECK((m->adam_step_mask & groups) == 0, "parameter group mask %d was already updated in step %d", groups & m->adam_step_mask, step);
    hipStream_t st = S_(stream);
    const bool cover = !m->cfg.freeze_encoder;
    // A backward entry point called with a deferred join leaves the last stages' weight-gradient groups, column sums and LayerNorm
    // reductions queued for the next entry point: an update issued now would consume incomplete gradients (and those parameters would
    // never train -- the late gradients are zeroed at the next step).  Refuse instead of guessing.
    B2S_CHECK(m->dw_pending.empty() && m->aux_jobs.empty() && m->ln_jobs.n == 0 && m->dw_stages_pending == 0 && m->unflushed_stages.empty(),
              "b2s_adam_step_groups: gradient work of the last backward stages is still queued (the preceding backward call deferred its "
              "join): call it without B2S_DEC_BWD_DEFER_JOIN / B2S_POST_BWD_DEFER_JOIN before a partial optimizer step");
    if (behind_mark) {
        // behind the mark only (b2s_model_mark_grads_ready): whatever the second stream was given after the mark -- the encoder's
        // weight-gradient groups -- belongs to groups that are not in `groups`
        B2S_CHECK(m->grads_marked, "b2s_adam_step_groups(behind_mark = 1) needs b2s_model_mark_grads_ready after the last backward call of these groups");
        B2S_CHECK(!(groups & B2S_ADAM_ENCODER), "behind_mark = 1 is for the groups whose gradients were complete at the mark");
    }
    const AdamHyper dhp = adam_hyper(lr, step, beta1, beta2);
    if (behind_mark) {
        B2S_HIP(hipStreamWaitEvent(st, m->grads_mark_ev, 0));
        m->grads_marked = false;
    } else {
        // the weight-gradient groups a backward entry point handed to the second stream without joining it (B2S_DEC_BWD_FLUSH_TAIL)
        // must have landed first
        B2S_TRY(join_aux(m, st));
    }
    static const int order[3] = {1, 2, 0};                     // decoder, postnet, encoder
    // behind the mark the update runs beside the encoder backward (on its own stream) and the last weight-gradient groups: a grid capped
    // at 512 workgroups -- the full one saturates HBM and the latency-bound encoder kernels take 3-4x as long while it runs (measured,
    // profiles/NOTES_r04.md: caps 64 / 128 / 256 / 384 / 512 / 768 / 1024: 7.98 / 7.54 / 7.46 / 7.44 / 7.39 / 7.39 / 7.46 ms per step)
#ifdef B2S_LAB
    static const int kTailWorkgroups = getenv("B2S_LAB_TAIL_ADAM_WG") ? atoi(getenv("B2S_LAB_TAIL_ADAM_WG")) : 512;
#else
    constexpr int kTailWorkgroups = 512;
#endif
    for (int k = 0; k < 3; ++k) {
        const int g = order[k], lo = m->adam_grp[g], n = m->adam_grp[g + 1] - lo;
        if (!(groups >> g & 1) || n <= 0) continue;
        B2S_TRY(ro_mt_adam(m->adam_chunks + lo, n, dhp, beta1, beta2, eps, l2, grad_scale, cover ? m->l2_part + lo : nullptr, st, m->adam_wire, m->adam_gbase,
                           behind_mark ? kTailWorkgroups : 0));
        if (g == 2 && m->dtype == 0) B2S_TRY(relayout_convs(m, st));
    }
    m->adam_step_mask |= groups;
    // the per-chunk sums of squares cover the regulariser once every group has been stepped (groups whose chunk range is
    // empty -- a frozen encoder -- never are: `cover` is false then)
    m->l2_fresh = cover && m->adam_step_mask == 7;
    return 0;
}
// Sharded optimizer (reduce-scatter + all-gather data parallelism; train.py:125,130-131,188-189 semantics: the mean gradient feeds Adam, every rank ends the step
// with the same parameters).  lo / hi: the n sorted, disjoint element ranges of the flat gradient buffer starting at grad_base that THIS rank owns (the slices the
// reduce-scatter leaves it); b2s_adam_step then updates exactly those.  n = 0 un-shards.
extern "C" int b2s_adam_shard(b2s_model* m, const float* grad_base, const int64_t* lo, const int64_t* hi, int n) {
    B2S_TRY(check_bound(m));
    B2S_CHECK(m->adam_chunks, "b2s_adam_shard: bind the Adam state first (b2s_adam_bind)");
    B2S_CHECK(n == 0 || (grad_base && lo && hi && n > 0), "null argument");
    std::vector<std::pair<long, long>> r;
    for (int i = 0; i < n; ++i) {
        B2S_CHECK(lo[i] >= 0 && hi[i] > lo[i] && (i == 0 || lo[i] >= hi[i - 1]), "b2s_adam_shard: ranges must be sorted, disjoint and non-empty (range %d)", i);
        r.push_back({(long)lo[i], (long)hi[i]});
    }
    m->shard_ranges = r;
    m->shard_gbase = n ? grad_base : nullptr;
    m->l2_fresh = false;
    return build_shard_tables(m);
}
// The parameter wire of the sharded optimizer: `wire` is a flat fp32 buffer laid out like the flat gradient buffer (element i <-> gradient element i).
// direction 0: pack -- the parameters this rank's b2s_adam_step just updated are copied into their wire positions (ahead of the all-gather of the wire);
// direction 1: scatter -- every parameter element this rank does NOT own is overwritten from the wire (after the all-gather), together with its compute-dtype
// shadow and the conv GEMM images, exactly as the Adam kernel would have written them: all ranks end the step with identical masters and shadows.
extern "C" int b2s_param_wire(b2s_model* m, float* wire, int direction, void* stream) {
    B2S_TRY(check_bound(m));
    B2S_CHECK(wire && (direction == 0 || direction == 1), "bad argument");
    B2S_CHECK(!m->shard_ranges.empty(), "b2s_param_wire: the optimizer is not sharded (b2s_adam_shard)");
    B2S_CHECK(((size_t)wire & 15) == 0, "the parameter wire must be 16-byte aligned");
    hipStream_t st = S_(stream);
    if (direction == 0) return ro_mt_param_wire(m->shard_chunks, m->n_shard_chunks, wire, m->shard_gbase, false, st);
    B2S_TRY(ro_mt_param_wire(m->other_chunks, m->n_other_chunks, wire, m->shard_gbase, true, st));
    if (m->dtype == 0) B2S_TRY(relayout_convs(m, st));                // fp32 mode: the conv GEMM images follow the (now complete) masters
    return 0;
}
// Marks the point of the second stream behind all gradient work handed to it so far (a decoder backward called with
// B2S_DEC_BWD_FLUSH_TAIL leaves its last weight-gradient groups running there).  b2s_adam_step_groups(on_aux = 2) waits for the mark
// instead of joining the second stream -- the caller may enqueue the encoder backward (on another stream) in between, so that a
// failure there still finds no part of the optimizer step applied.
extern "C" int b2s_model_mark_grads_ready(b2s_model* m) {
    B2S_TRY(check_bound(m));
    B2S_CHECK(m->aux, "no second stream");
    B2S_CHECK(m->dw_pending.empty() && m->aux_jobs.empty() && m->ln_jobs.n == 0 && m->dw_stages_pending == 0,
              "b2s_model_mark_grads_ready: gradient work is still queued on the host side (call the backward entry point with B2S_DEC_BWD_FLUSH_TAIL)");
    if (!m->grads_mark_ev) B2S_HIP(hipEventCreateWithFlags(&m->grads_mark_ev, hipEventDisableTiming));
#ifdef B2S_LAB
    if (m->lab_early_marked) { m->lab_early_marked = false; m->grads_marked = true; return 0; }
#endif
    B2S_HIP(hipEventRecord(m->grads_mark_ev, m->aux));
    m->grads_marked = true;
    return 0;
}
// A backward pass that was given up between entry points (a failed call, an exception in the caller between postnet / decoder / encoder
// backward): drop whatever the deferred-join protocol still holds -- queued weight-gradient GEMMs, bias / LayerNorm reductions and stage
// hooks, all of which point into contexts the caller is about to free -- without launching or firing any of it, and make `stream` wait
// for what the second stream already runs.  The gradient buffers are left incomplete: zero them before the next backward.
extern "C" int b2s_model_backward_abort(b2s_model* m, void* stream) {
    B2S_CHECK(m, "null model");
    m->dw_pending.clear(); m->dw_stages_pending = 0;
    m->aux_jobs.clear();
    m->ln_jobs.n = 0;
    m->pending_stages.clear(); m->unflushed_stages.clear();
    m->pending_ev = nullptr;
    m->dw_hold_from = -1; m->dw_tail_cap = 0; m->dw_flush_capped = false;      // (a decoder backward that failed mid-call leaves its tail policy set)
    m->grads_marked = false;
    if (m->side_ev) { B2S_HIP(hipStreamWaitEvent(S_(stream), m->side_ev, 0)); m->side_ev = nullptr; }
    if (m->aux) { m->aux_dirty = true; B2S_TRY(join_aux(m, S_(stream))); }
    return 0;
}
extern "C" int b2s_model_set_grad_slot_padding(b2s_model* m, int bytes) {
    B2S_CHECK(m && bytes >= 0 && bytes <= 4096, "grad slot padding must be 0..4096 bytes");
    m->grad_pad_bytes = (size_t)bytes;
    return 0;
}
extern "C" int b2s_zero_grads(b2s_model* m, void* stream, int flags) {
    B2S_TRY(check_bound(m));
    // a new backward pass starts here: nothing of an earlier one may still be queued (it would run on freed contexts)
    if (!m->dw_pending.empty() || !m->aux_jobs.empty() || m->ln_jobs.n > 0 || !m->pending_stages.empty() || !m->unflushed_stages.empty())
        B2S_TRY(b2s_model_backward_abort(m, stream));
    m->dw_overwrite_pass = false;
    if ((flags & B2S_ZERO_GRADS_OVERWRITE_DW) && m->dw_group && !m->dw_ow_ptrs.empty()) {
        // the caller runs the WHOLE backward pass (every segment, once): layer weight gradients are stored by their grouped launch, only
        // the accumulating rest is cleared
        m->dw_overwrite_pass = true;
        return ro_mt_zero(m->zero_chunks, m->n_zero_chunks, S_(stream));
    }
    // coalesce adjacent gradient buffers (the host normally binds one flat buffer) into few memsets
    std::vector<std::pair<char*, size_t>> r;
    for (size_t i = 0; i < m->tinfo.size(); ++i)
        if (m->tinfo[i].kind == 1 && m->grad[i]) r.push_back({(char*)m->grad[i], (size_t)m->tinfo[i].numel * 4});
    std::sort(r.begin(), r.end());
    size_t i = 0;
    while (i < r.size()) {
        char* lo = r[i].first; char* hi = lo + r[i].second;
        size_t j = i + 1;
        // Gaps between gradient ranges are bridged only when the caller declared them padding of ONE flat buffer
        // (b2s_model_set_grad_slot_padding: the engine's own 256-byte slots).  Separately bound gradient tensors -- views into a larger
        // buffer with live data in between, sub-allocations of a custom allocator -- are cleared range by range.
        const size_t bridge = m->grad_pad_bytes;
        while (j < r.size() && r[j].first <= hi + bridge) { hi = std::max(hi, r[j].first + r[j].second); ++j; }
        B2S_HIP(hipMemsetAsync(lo, 0, (size_t)(hi - lo), S_(stream)));
        i = j;
    }
    return 0;
}

int b2s_ensure_pe_export(b2s_model* m, int len) { return ensure_pe(m, len); }

// exported to capi_ops.hip
int b2s_attn_core_fwd_export(int dtype, hipStream_t st, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv,
                             void* ctx, int ldc, int B, int H, int Lq, int Lk, int dh, int mask_mode, const int* klen,
                             const float* bias, long bias_sb, long bias_sq, DropCfg drop, float* S, void* P, void* Pd) {
    return attn_core_fwd(dtype, st, q, ldq, k, ldk, v, ldv, ctx, ldc, B, H, Lq, Lk, dh, mask_mode, klen, bias, bias_sb, bias_sq, drop, S, P, Pd);
}
int b2s_attn_core_bwd_export(int dtype, hipStream_t st, const void* dctx, int ldc, const void* q, int ldq, const void* k, int ldk,
                             const void* v, int ldv, const void* P, const void* Pd, void* dq, int lddq, void* dk, int lddk,
                             void* dv, int lddv, int B, int H, int Lq, int Lk, int dh, DropCfg drop, float* dP, void* dS) {
    return attn_core_bwd(dtype, st, dctx, ldc, q, ldq, k, ldk, v, ldv, P, Pd, dq, lddq, dk, lddk, dv, lddv, B, H, Lq, Lk, dh, drop, dP, dS);
}
