#!/usr/bin/env python3
"""Generate golden fixtures by running the REFERENCE (imported from /root/reference).

Runs only in the build container (the reference never travels to the GPU box); its outputs
(tests/golden/*.npz, *.json) are committed as data.  Inputs and weights come from
oracle/synth.py (NumPy default_rng) so the fixtures hold outputs only.

    python tests/golden/make_goldens.py            # writes all groups

Groups: G1 helpers, G2 tiny model fwd/loss/grads/Adam/eval-mode,
G3 per-module, G4 autoregressive decode, G5 full-size spot checks, G6 LR schedule + packer,
G7 batch producer (packer caps, collate contract, adapt-rate ramp), G8 feeders, G9 full-size autoregressive decode.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("B2S_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)
for _m in ("librosa", "librosa.filters", "librosa.effects", "soundfile", "fastdtw"):
    sys.modules.setdefault(_m, types.ModuleType(_m))        # only utils.audio/infolog plotting needs them
sys.modules["fastdtw"].fastdtw = None                       # name imported by utils/infolog.py:8, never called here

from hyperparams import hparams as hp                       # noqa: E402  (reference)
from transformer import common as rcommon                   # noqa: E402
from transformer import attention as rattn                  # noqa: E402
from transformer import modules as rmod                     # noqa: E402
from transformer import tacotron as rtaco                   # noqa: E402
import synthesize as rsynth                                 # noqa: E402
import dataloader as rdata                                  # noqa: E402

from oracle import synth, TINY, TINY96, make_config         # noqa: E402

torch.set_num_threads(8)
_DEFAULT_HP = dict(hp.values())


def reset_hp(overrides=""):
    hp.override_from_dict(_DEFAULT_HP)
    if overrides:
        hp.parse(overrides)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote %s.npz (%d arrays, %.1f KB)" % (name, len(out),
                                                 os.path.getsize(os.path.join(HERE, name + ".npz")) / 1024))


def build_model(overrides, seed=1234):
    reset_hp(overrides)
    cfg = make_config(overrides)
    m = rtaco.Tacotron(hp)
    st = synth.synthetic_state(cfg, seed)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in st.items()}, strict=True)
    return m, cfg


def tbatch(b):
    return {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in b.items()}


# ------------------------------------------------------------------------------- G1
def g1():
    mask = torch.tensor([[1, 1, 1, 0, 0], [1, 1, 1, 1, 1]], dtype=torch.bool)
    x3 = torch.arange(2 * 5 * 3, dtype=torch.float32).reshape(2, 5, 3) + 1
    xc = torch.arange(2 * 3 * 5, dtype=torch.float32).reshape(2, 3, 5) + 1
    lens = torch.tensor([3, 5])
    loss = torch.arange(10, dtype=torch.float32).reshape(2, 5) * 0.25 + 1
    save("g1_helpers",
         pe_37_64=rcommon.get_sinusoid_encoding_table(37, 64),
         pe_9_7=rcommon.get_sinusoid_encoding_table(9, 7),
         pe_1100_768_rows=rcommon.get_sinusoid_encoding_table(1100, 768)[[0, 1, 599, 1099]],
         bias_causal_6=rcommon.attention_bias(6, "causal"),
         bias_masking=rcommon.attention_bias(mask, "masking"),
         impute_cl=rcommon.impute(x3, lens), impute_cf=rcommon.impute(xc, lens, channels_last=False),
         impute_2d=rcommon.impute(loss, lens),
         mask_reduce_all=rcommon.mask_reduce(loss, lens), mask_reduce_ps=rcommon.mask_reduce(loss, lens, True))


# ------------------------------------------------------------------------------- G2
def g2(tag, overrides, full_grads):
    m, cfg = build_model(overrides)
    names_shapes = [[k, list(v.shape)] for k, v in m.state_dict().items()]
    with open(os.path.join(HERE, "state_layout_%s.json" % tag), "w") as f:
        json.dump(names_shapes, f)
    nb = synth.synthetic_batch(cfg, B=3, S=11, T=23, seed=7, in_lens=[11, 7, 4], tgt_lens=[23, 15, 9])
    b = tbatch(nb)
    arrs = {}
    # eval-mode forward (BN running stats)
    m.eval()
    with torch.no_grad():
        o = m(**b)
    arrs.update(eval_mel_bef=o["mel_bef"], eval_mel_aft=o["mel_aft"], eval_stop=o["stop_logits"])
    # train-mode forward / loss / grads / Adam
    m.train()
    optim = torch.optim.Adam(m.parameters(), lr=hp.max_lr, eps=hp.adam_eps)
    from functools import partial
    sched = torch.optim.lr_scheduler.LambdaLR(optim, lr_lambda=partial(rtaco.learning_rate_schedule, hp=hp))
    for step in range(3):
        o = m(**b)
        losses = rtaco.compute_loss(m, b["mel_targets"], b["target_lengths"], o, hp)
        optim.zero_grad()
        losses["loss"].backward()
        if step == 0:
            arrs.update(mel_bef=o["mel_bef"], mel_aft=o["mel_aft"], stop=o["stop_logits"])
            for i in range(cfg.n_decoder_layer):
                arrs["align_self_%d" % i] = o["alignments"]["self"][i]
                arrs["align_encdec_%d" % i] = o["alignments"]["encdec"][i]
            for k, v in losses.items():
                arrs["loss_" + k] = v
            for n, p in m.named_parameters():
                g = p.grad if p.grad is not None else torch.zeros_like(p)
                arrs["gnorm/" + n] = g.double().norm()
                if full_grads:
                    arrs["grad/" + n] = g
                else:
                    arrs["gslice/" + n] = g.flatten()[:16]
        optim.step()
        sched.step()
        if step in (0, 2):
            sd = m.state_dict()
            for n, v in sd.items():
                if full_grads and (v.numel() <= 20000 or "stop_net" in n or "batchnorm" in n):
                    arrs["after%d/%s" % (step + 1, n)] = v.detach().clone()
                arrs["after%d_norm/%s" % (step + 1, n)] = v.double().norm()
            arrs["after%d_loss" % (step + 1)] = losses["loss"]
    save("g2_model_%s" % tag, **arrs)


# ------------------------------------------------------------------------------- G3
def g3():
    reset_hp(TINY)
    rng = np.random.default_rng(99)
    arrs = {}

    def rnd(*shape, scale=1.0):
        return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))

    # MultiheadAttention self (C=64,H=2) and cross (C=128,H=2), with grads
    for kind, C, Lq, Lk in (("self", 64, 9, 9), ("cross", 128, 10, 7)):
        mha = rattn.MultiheadAttention(C, C, kind == "self", 2, dropout_rate=0.0)
        for n, p in mha.named_parameters():
            p.data = rnd(*p.shape, scale=C ** -0.5)
            arrs["mha_%s/w/%s" % (kind, n)] = p.data.clone()
        q = rnd(2, Lq, C).requires_grad_(True)
        mem = None if kind == "self" else rnd(2, Lk, C).requires_grad_(True)
        lens = torch.tensor([Lk, Lk - 3])
        mask = torch.arange(Lk)[None, :] < lens[:, None]
        bias = rcommon.attention_bias(mask, "masking") if kind == "cross" else rcommon.attention_bias(Lq, "causal")
        out = mha(q, mem, bias)
        go = rnd(*out["outputs"].shape)
        out["outputs"].backward(go)
        arrs.update({"mha_%s/q" % kind: q.detach(), "mha_%s/lens" % kind: lens, "mha_%s/out" % kind: out["outputs"],
                     "mha_%s/align" % kind: out["align"], "mha_%s/go" % kind: go, "mha_%s/dq" % kind: q.grad})
        if mem is not None:
            arrs["mha_cross/mem"] = mem.detach()
            arrs["mha_cross/dmem"] = mem.grad
        for n, p in mha.named_parameters():
            arrs["mha_%s/dw/%s" % (kind, n)] = p.grad

    # FFNLayer
    f = rmod.FFNLayer(64, 256, 64, dropout_rate=0.0)
    for n, p in f.named_parameters():
        p.data = rnd(*p.shape, scale=0.1)
        arrs["ffn/w/" + n] = p.data.clone()
    x = rnd(2, 5, 64)
    arrs.update({"ffn/x": x, "ffn/out": f(x)})

    # Whole-module checks on the TINY model (weights from synth)
    m, cfg = build_model(TINY)
    m.eval()
    nb = synth.synthetic_batch(cfg, B=2, S=9, T=14, seed=3, in_lens=[9, 5], tgt_lens=[14, 8])
    b = tbatch(nb)
    with torch.no_grad():
        arrs["prenet/out"] = m.decoder.prenet(b["mel_targets"])
        enc = m.encoder(b["inputs"], b["input_lengths"], b["input_spk_ids"], b["input_language_vecs"])
        arrs["encoder/out"] = enc
        for lo in (False, True):
            mels, stop, al = m.decoder(enc, b["input_lengths"], b["mel_targets"], b["target_lengths"], leave_one=lo)
            arrs["decoder_lo%d/mels" % lo] = mels
            arrs["decoder_lo%d/stop" % lo] = stop
            arrs["decoder_lo%d/align_encdec_1" % lo] = al["encdec"][1]
        arrs["postnet_eval/out"] = m.postnet(b["mel_targets"], b["target_lengths"])
        m.postnet.train()
        arrs["postnet_train/out"] = m.postnet(b["mel_targets"], b["target_lengths"])
        for i in range(cfg.n_postnet_layer):
            arrs["postnet_train/running_mean_%d" % i] = m.postnet.batchnorm_layers[i].running_mean
            arrs["postnet_train/running_var_%d" % i] = m.postnet.batchnorm_layers[i].running_var
    save("g3_modules", **arrs)


def pick_staggered_bias(raw, lo, hi):
    """raw [B, T]: stop logits without the bias, from a run that never stops (batch rows are independent and a row's
    logits up to its stop do not depend on the bias).  Returns a bias b such that the first frames with raw + b > 0 are
    pairwise different and lie in [lo, hi) (one row may never stop when no bias separates all of them): the staggered-stop /
    zero-tail logic then runs for several frames."""
    cands = np.unique(-raw[:, lo:hi].ravel())
    cands = np.concatenate([(cands[:-1] + cands[1:]) * 0.5, cands[-1:] + 1.0])
    best = None
    for b in cands:
        hit = raw + b > 0
        first = np.where(hit.any(1), hit.argmax(1), raw.shape[1])
        finite = first[first < raw.shape[1]]
        # pairwise different stop frames, all >= lo; at most one row may never stop (that row reports frames + 1)
        if len(set(first.tolist())) == raw.shape[0] and first.min() >= lo and len(finite) >= raw.shape[0] - 1 and finite.max() < hi:
            spread = (len(finite), np.diff(np.sort(first)).min())
            if best is None or spread > best[0]:
                best = (spread, float(b), first)
    assert best is not None, "no bias staggers the stops"
    print("   staggered stops at frames", best[2].tolist(), "bias %.5f" % best[1])
    return best[1]


# ------------------------------------------------------------------------------- G9
def g9():
    """Full-size autoregressive decode (BASELINE configs[3] at a CPU-feasible size): default hparams, dropout rates 0 (so
    that decoder.train() synthesis is deterministic), B=3, S=40, max_generation_frames=48,
    stop bias chosen so the three utterances stop at three different frames >= 8."""
    over = "transformer_dropout_rate=0.0,decoder_dropout_rate=0.0,max_generation_frames=48"
    m, cfg = build_model(over, seed=77)
    nb = synth.synthetic_batch(cfg, B=3, S=40, T=4, seed=21, in_lens=[40, 31, 36], n_spk=572, n_lang=38)
    b = tbatch(nb)
    b.pop("mel_targets"); b.pop("target_lengths")
    m.eval()
    m.decoder.stop_net.bias.data.fill_(-100.0)
    r = rsynth.eval_batch(m, b, use_bar=False, bar_interval=-1)
    with torch.no_grad():
        enc = m.encoder(b["inputs"], b["input_lengths"], b["input_spk_ids"], b["input_language_vecs"])
        _, sl, _ = m.decoder(enc, b["input_lengths"], torch.from_numpy(r["mel_pre"]), torch.full([3], 48, dtype=torch.int32))
    bias = pick_staggered_bias((sl + 100.0).numpy(), lo=8, hi=46)
    m.decoder.stop_net.bias.data.fill_(bias)
    r = rsynth.eval_batch(m, b, use_bar=False, bar_interval=-1)
    arrs = {"stop_bias": np.float32(bias), "mel_pre": r["mel_pre"], "mel_aft": r["mel_aft"],
            "generated_lengths": np.asarray(r["generated_lengths"])}
    for i in range(cfg.n_decoder_layer):
        a = r["alignments"]["encdec"][i]                             # [B,H,S,T]
        arrs["align_argmax_%d" % i] = a.argmax(axis=2).astype(np.int32)
        top2 = np.sort(a, axis=2)[:, :, -2:, :]
        arrs["align_margin_%d" % i] = (top2[:, :, 1] - top2[:, :, 0]).astype(np.float32)      # arg-max margin (ties are not comparable)
    print("g9 generated_lengths", r["generated_lengths"])
    save("g9_decode_fullsize", **arrs)


# ------------------------------------------------------------------------------- G4
def g4():
    arrs = {}
    for tag, over in (("tiny", TINY), ("tiny96", TINY96)):
        for case, stop_bias in (("never", -100.0), ("mixed", None), ("first", 100.0)):
            m, cfg = build_model(over + ",max_generation_frames=40")
            nb = synth.synthetic_batch(cfg, B=3, S=10, T=4, seed=11, in_lens=[10, 6, 8])
            b = tbatch(nb)
            b.pop("mel_targets"); b.pop("target_lengths")
            m.eval()
            if stop_bias is not None:
                m.decoder.stop_net.bias.data.fill_(stop_bias)
            else:
                # pick a bias that makes the samples stop at different steps: run once with a
                # never-stopping bias, read the per-step logits, and place the threshold between them
                m.decoder.stop_net.bias.data.fill_(-100.0)
                r = rsynth.eval_batch(m, b, use_bar=False, bar_interval=-1)
                with torch.no_grad():
                    enc = m.encoder(b["inputs"], b["input_lengths"], b["input_spk_ids"], b["input_language_vecs"])
                    mels = torch.from_numpy(r["mel_pre"])
                    dec_in = torch.cat([torch.zeros(3, 0, 80), mels], dim=1)
                    _, sl, _ = m.decoder(enc, b["input_lengths"], dec_in,
                                         torch.full([3], 40, dtype=torch.int32))
                raw = sl + 100.0                                    # logits without the bias
                bias = pick_staggered_bias(raw.numpy(), lo=8, hi=38)
                m.decoder.stop_net.bias.data.fill_(bias)
                arrs["%s_%s/stop_bias" % (tag, case)] = np.float32(bias)
            r = rsynth.eval_batch(m, b, use_bar=False, bar_interval=-1)
            arrs["%s_%s/mel_pre" % (tag, case)] = r["mel_pre"]
            arrs["%s_%s/mel_aft" % (tag, case)] = r["mel_aft"]
            arrs["%s_%s/generated_lengths" % (tag, case)] = np.asarray(r["generated_lengths"])
            for i in range(cfg.n_decoder_layer):
                a = r["alignments"]["encdec"][i]                     # [B,H,S,T]
                arrs["%s_%s/align_argmax_%d" % (tag, case, i)] = a.argmax(axis=2).astype(np.int32)
            print(tag, case, "generated_lengths", r["generated_lengths"])
    save("g4_decode", **arrs)


# ------------------------------------------------------------------------------- G5
def g5():
    overrides = "transformer_dropout_rate=0.0,decoder_dropout_rate=0.0"
    m, cfg = build_model(overrides, seed=4321)
    nb = synth.synthetic_batch(cfg, B=4, S=100, T=600, seed=0, in_lens=[100, 90, 80, 70],
                               tgt_lens=[600, 550, 500, 450], n_spk=572, n_lang=38)
    b = tbatch(nb)
    m.train()
    o = m(**b)
    losses = rtaco.compute_loss(m, b["mel_targets"], b["target_lengths"], o, hp)
    losses["loss"].backward()
    arrs = {"mel_bef_slice": o["mel_bef"][:, :4, :8], "mel_aft_slice": o["mel_aft"][:, :4, :8],
            "mel_bef_tail": o["mel_bef"][:, 440:452, :4], "stop_slice": o["stop_logits"][:, :16],
            "mel_bef_norm": o["mel_bef"].double().norm(), "mel_aft_norm": o["mel_aft"].double().norm(),
            "stop_argmax": o["stop_logits"].argmax(-1),
            "align_encdec5_argmax": o["alignments"]["encdec"][5].argmax(2)[:, :, ::25].to(torch.int32),
            "align_encdec5_max": o["alignments"]["encdec"][5].max(2)[0][:, :, ::25]}
    for k, v in losses.items():
        arrs["loss_" + k] = v
    for n, p in m.named_parameters():
        arrs["gnorm/" + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).double().norm()
    save("g5_fullsize", **arrs)


# ------------------------------------------------------------------------------- G6
def g6():
    reset_hp()
    steps = [0, 50000, 50001, 325000, 600000, 2000000]
    lrs = [rtaco.learning_rate_schedule(s, hp) for s in steps]
    rng = np.random.default_rng(5)
    tl = np.sort(rng.integers(120, 860, size=400))
    il = np.maximum(8, (tl * 0.196 + rng.integers(-8, 8, size=400)).astype(np.int64))
    ex = [{"input": np.zeros(int(i)), "mel_target": np.zeros((int(t), 1))} for i, t in zip(il, tl)]
    batches = rdata._pack_into_batches(ex, hparams=hp)
    sizes = [len(b) for b in batches]
    # L2 membership and parameter count at default hparams
    m = rtaco.Tacotron(hp)
    l2 = [n for n, p in m.named_parameters() if 'weight' in n and 'layer_norm' not in n and 'batchnorm' not in n
          and 'encoder.speaker_embed' not in n and 'encoder.embed' not in n]
    with open(os.path.join(HERE, "state_layout_default.json"), "w") as f:
        json.dump([[k, list(v.shape)] for k, v in m.state_dict().items()], f)
    with open(os.path.join(HERE, "l2_members_default.json"), "w") as f:
        json.dump(l2, f)
    save("g6_misc", lr_steps=np.asarray(steps), lr_values=np.asarray(lrs, dtype=np.float64),
         pack_in_lens=il, pack_tgt_lens=tl, pack_sizes=np.asarray(sizes),
         n_params=np.asarray(sum(p.numel() for p in m.parameters())))


# ------------------------------------------------------------------------------- G7
def g7():
    """Batch producer: packer on unsorted / capped inputs, collate contract, adapt rate, sharding."""
    from types import SimpleNamespace
    reset_hp()
    rng = np.random.default_rng(11)
    ex = []
    for i in range(9):
        L, T = int(rng.integers(5, 30)), int(rng.integers(20, 90))
        lv = np.zeros([hp.max_num_language]); lv[int(rng.integers(0, 5))] = 1
        ex.append({"name": "spk%d_utt%d" % (i % 3, i), "input": rng.integers(1, 255, size=L).astype(np.int32),
                   "mel_target": rng.standard_normal((T, hp.num_mels)).astype(np.float32), "target_length": T,
                   "language_vec": lv, "speaker_id": int(rng.integers(0, 50))})
    b = rdata._prepare_batch(ex, hp)
    proto = rdata.get_input_proto(hp)
    tb = {k: proto[k](b[k]) for k in proto if k in b}                 # exactly what Feeder._enqueue_next_group enqueues
    out = {"n": np.asarray(len(ex))}
    for i, e in enumerate(ex):
        out["ex%d_input" % i] = e["input"]; out["ex%d_mel" % i] = e["mel_target"]
        out["ex%d_lang" % i] = e["language_vec"]; out["ex%d_spk" % i] = np.asarray(e["speaker_id"])
    for k, v in tb.items():
        if k == "names":
            out["names"] = np.asarray(v)
        else:
            out["batch_" + k] = v.numpy(); out["dtype_" + k] = np.asarray(str(v.dtype))
    # packer: unsorted examples, tight caps, `single`
    tl = rng.integers(40, 400, size=60); il = rng.integers(8, 90, size=60)
    exs = [{"input": np.zeros(int(i)), "mel_target": np.zeros((int(t), 1))} for i, t in zip(il, tl)]
    caps = SimpleNamespace(batch_frame_limit=1200, batch_frame_quad_limit=400000)
    out["pk_in"] = il; out["pk_tgt"] = tl
    out["pk_sizes_tight"] = np.asarray([len(x) for x in rdata._pack_into_batches(exs, hparams=caps)])
    out["pk_sizes_single"] = np.asarray([len(x) for x in rdata._pack_into_batches(exs, single=True, hparams=caps)])
    noT = [{"input": np.zeros(int(i))} for i in il]                   # synthesis-time packing: target length = 1.5 x input
    out["pk_sizes_notarget"] = np.asarray([len(x) for x in rdata._pack_into_batches(noT, hparams=caps)])
    # adapt-rate ramp (Feeder._adapt_rate)
    steps = [0, 999, 1000, 1500, 2000, 5000]
    h2 = SimpleNamespace(adapt_start_step=1000, adapt_end_step=2000, final_adapt_rate=0.25)
    out["adapt_steps"] = np.asarray(steps)
    out["adapt_rates"] = np.asarray([rdata.Feeder._adapt_rate(SimpleNamespace(global_step=s_, _hparams=h2)) for s_ in steps])
    h3 = SimpleNamespace(adapt_start_step=30000, adapt_end_step=30000, final_adapt_rate=0.25)      # reference default: a step
    out["adapt_rates_default"] = np.asarray([rdata.Feeder._adapt_rate(SimpleNamespace(global_step=s_, _hparams=h3)) for s_ in (29999, 30000)])
    save("g7_batching", **out)


# ------------------------------------------------------------------------------- G8
def g8():
    """On-disk formats + feeders: the reference's Feeder / FeederEval run on the synthetic
    corpus of oracle/synth.py; the fixture holds batch compositions (names) and checksums."""
    import tempfile
    out = {}
    CFGS = {
        "balanced": "bucket_size=16,batch_frame_limit=400,batch_frame_quad_limit=60000,data_warmup_steps=5,"
                    "target_length_lower_bound=40,target_length_upper_bound=100",
        "plain": "bucket_size=16,batch_frame_limit=400,batch_frame_quad_limit=60000,balanced_training=false,data_warmup_steps=0",
        "adapt": "bucket_size=16,batch_frame_limit=400,batch_frame_quad_limit=60000,data_warmup_steps=0,"
                 "adapt_start_step=0,adapt_end_step=0,final_adapt_rate=0.5",
    }
    with tempfile.TemporaryDirectory() as d:
        c = synth.synthetic_corpus(d, seed=3, n=48)
        for tag, over in CFGS.items():
            for rank, world in ((0, 1), (1, 2)):
                reset_hp(over)
                rdata.zip_cache.clear()
                kw = dict(adapt_lang=["fr-fr"]) if tag == "adapt" else {}
                f = rdata.Feeder(c["zip"], c["meta"], hp, spk_to_id=c["spk_ids"], lang_to_id=c["lang_ids"], rank=rank, world_size=world, **kw)
                seq = []
                for _ in range(3):
                    f._enqueue_next_group()
                    while not f.queue.empty():
                        b = f.queue.get()
                        seq.append({"names": b["names"], "in_sum": int(b["inputs"].sum()), "mel_sum": float(b["mel_targets"].double().sum()),
                                    "tl": b["target_lengths"].tolist(), "spk": b["input_spk_ids"].tolist(),
                                    "lang": b["input_language_vecs"].argmax(1).tolist()})
                out["train/%s/r%dw%d" % (tag, rank, world)] = seq
        reset_hp("batch_frame_limit=400,batch_frame_quad_limit=60000")
        rdata.zip_cache.clear()
        fe = rdata.FeederEval(c["zip"], c["meta"], hp, spk_to_id=c["spk_ids"], lang_to_id=c["lang_ids"], shuffle=True, keep_order=True,
                              pick_partial=True)
        out["eval/partial"] = [{"names": b["names"], "tl": b["target_lengths"].tolist()} for b in fe.fetch_data()]
        rdata.zip_cache.clear()
        fe = rdata.FeederEval(c["zip"], c["meta"], hp, spk_to_id=c["spk_ids"], lang_to_id=c["lang_ids"], eval_lang=["de-de"], shuffle=False,
                              target_spk="spkB")
        out["eval/de_as_spkB"] = [{"names": b["names"], "spk": b["input_spk_ids"].tolist()} for b in fe.fetch_data()]
        meta = rdata._read_meta(open(c["meta"], encoding="utf-8"), "nlti")
        out["meta_head"] = meta[:3]
    reset_hp()
    with open(os.path.join(HERE, "g8_feeders.json"), "w") as f:
        json.dump(out, f, indent=0, ensure_ascii=False)
    print("wrote g8_feeders.json (%d sequences)" % len(out))


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9"]
    torch.manual_seed(0)
    if "g1" in which: g1()
    if "g2" in which:
        g2("tiny", TINY, full_grads=True)
        g2("tiny96", TINY96, full_grads=False)
    if "g3" in which: g3()
    if "g4" in which: g4()
    if "g5" in which: g5()
    if "g6" in which: g6()
    if "g7" in which: g7()
    if "g8" in which: g8()
    if "g9" in which: g9()
