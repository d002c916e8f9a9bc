"""Pin the CPU oracle against the reference-generated golden fixtures (no GPU needed).

Tolerance: 1e-5 abs on activations (fp32 re-ordering noise measured 7e-6 at full size,
SURVEY.md section 8c); gradient / parameter checks are relative to the tensor norm.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import b2s_oracle as O
from oracle import synth, make_config, TINY, TINY96

G = os.path.join(os.path.dirname(__file__), "golden")
torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))


def load(name):
    return dict(np.load(os.path.join(G, name + ".npz")))


def close(a, b, atol=1e-5, rtol=0.0):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


# ------------------------------------------------------------------ G1 helpers
def test_g1_helpers():
    g = load("g1_helpers")
    close(O.sinusoid_table(37, 64), g["pe_37_64"], atol=0)
    close(O.sinusoid_table(9, 7), g["pe_9_7"], atol=0)
    close(O.sinusoid_table(1100, 768)[[0, 1, 599, 1099]], g["pe_1100_768_rows"], atol=0)
    close(O.attention_bias_causal(6), g["bias_causal_6"], atol=0)
    mask = torch.tensor([[1, 1, 1, 0, 0], [1, 1, 1, 1, 1]], dtype=torch.bool)
    close(O.attention_bias_masking(mask), g["bias_masking"], atol=0)
    x3 = torch.arange(30, dtype=torch.float32).reshape(2, 5, 3) + 1
    xc = torch.arange(30, dtype=torch.float32).reshape(2, 3, 5) + 1
    lens = torch.tensor([3, 5])
    loss = torch.arange(10, dtype=torch.float32).reshape(2, 5) * 0.25 + 1
    close(O.impute(x3, lens), g["impute_cl"], atol=0)
    close(O.impute(xc, lens, channels_last=False), g["impute_cf"], atol=0)
    close(O.impute(loss, lens), g["impute_2d"], atol=0)
    close(O.mask_reduce(loss, lens), g["mask_reduce_all"], atol=1e-6)
    close(O.mask_reduce(loss, lens, True), g["mask_reduce_ps"], atol=1e-6)


# ------------------------------------------------------------------ state layout
@pytest.mark.parametrize("tag,over", [("tiny", TINY), ("tiny96", TINY96), ("default", "")])
def test_state_layout(tag, over):
    ref = json.load(open(os.path.join(G, "state_layout_%s.json" % tag)))
    mine = [[n, list(s)] for n, s, _ in synth.param_shapes(make_config(over))]
    assert mine == ref


def test_l2_membership_and_param_count():
    cfg = make_config("")
    ref = json.load(open(os.path.join(G, "l2_members_default.json")))
    mine = [n for n, s, k in synth.param_shapes(cfg) if O.is_parameter(n) and O.l2_member(n)]
    assert mine == ref
    n = sum(int(np.prod(s)) for nme, s, k in synth.param_shapes(cfg) if O.is_parameter(nme))
    assert n == int(load("g6_misc")["n_params"]) == 83477155


# ------------------------------------------------------------------ G2 model
def _g2_setup(over):
    cfg = make_config(over)
    nb = synth.synthetic_batch(cfg, B=3, S=11, T=23, seed=7, in_lens=[11, 7, 4], tgt_lens=[23, 15, 9])
    return cfg, synth.synthetic_state(cfg, 1234), O.to_torch_batch(nb)


@pytest.mark.parametrize("tag,over", [("tiny", TINY), ("tiny96", TINY96)])
def test_g2_forward_loss_grads_adam(tag, over):
    g = load("g2_model_" + tag)
    cfg, st, b = _g2_setup(over)
    # eval mode
    P = O.to_torch_state(st)
    with torch.no_grad():
        o = O.tacotron_forward(P, cfg, b, train=False)
    close(o["mel_bef"], g["eval_mel_bef"]); close(o["mel_aft"], g["eval_mel_aft"]); close(o["stop_logits"], g["eval_stop"])
    # three training steps
    P = O.to_torch_state(st, requires_grad=True)
    opt = {}
    for step in range(3):
        o, losses, grads = O.train_step(P, cfg, b, opt, step, train=True)
        if step == 0:
            close(o["mel_bef"], g["mel_bef"]); close(o["mel_aft"], g["mel_aft"]); close(o["stop_logits"], g["stop"])
            for i in range(cfg.n_decoder_layer):
                close(o["alignments"]["self"][i], g["align_self_%d" % i], atol=1e-6)
                close(o["alignments"]["encdec"][i], g["align_encdec_%d" % i], atol=1e-6)
            for k in ("loss", "bef_loss", "aft_loss", "aft_losses", "mse_loss", "l2", "stop_loss"):
                close(losses[k], g["loss_" + k], atol=1e-6, rtol=1e-5)
            for n, gr in grads.items():
                ref_norm = float(g["gnorm/" + n])
                assert abs(float(gr.double().norm()) - ref_norm) <= 1e-4 * ref_norm + 1e-7, n
                if "grad/" + n in g:
                    close(gr, g["grad/" + n], atol=1e-5 * max(1.0, ref_norm))
                else:
                    close(gr.flatten()[:16], g["gslice/" + n], atol=1e-5 * max(1.0, ref_norm))
        if step in (0, 2):
            # own-gradient trajectory: Adam's first steps are +-lr*sign(g), so elements with |g| ~ 0 are
            # ill-conditioned; the trajectory is checked through norms and the loss only
            for n, v in P.items():
                rn = float(g["after%d_norm/%s" % (step + 1, n)])
                assert abs(float(v.detach().double().norm()) - rn) <= 1e-4 * rn + 1e-5, (step, n)
            close(losses["loss"], g["after%d_loss" % (step + 1)], atol=1e-4)
    if tag == "tiny":
        # Adam formula pinned exactly: feed the reference's own gradients through adam_step
        P = O.to_torch_state(st)
        grads = {n: torch.from_numpy(g["grad/" + n]) for n in P if O.is_parameter(n)}
        with torch.no_grad():
            O.adam_step(P, grads, {}, 0, cfg)
        for n in grads:
            if "after1/" + n in g:
                close(P[n], g["after1/" + n], atol=1e-6)


# ------------------------------------------------------------------ G3 modules
def test_g3_modules():
    g = load("g3_modules")
    for kind, C in (("self", 64), ("cross", 128)):
        P = {"a." + k.split("/w/")[1]: torch.from_numpy(v).requires_grad_(True)
             for k, v in g.items() if k.startswith("mha_%s/w/" % kind)}
        q = torch.from_numpy(g["mha_%s/q" % kind]).requires_grad_(True)
        lens = torch.from_numpy(g["mha_%s/lens" % kind])
        if kind == "self":
            mem, bias = None, O.attention_bias_causal(q.shape[1])
        else:
            mem = torch.from_numpy(g["mha_cross/mem"]).requires_grad_(True)
            bias = O.attention_bias_masking(O.length_mask(lens, mem.shape[1]))
        out, align = O.multihead_attention(P, "a", q, mem, bias, 2)
        close(out, g["mha_%s/out" % kind]); close(align, g["mha_%s/align" % kind], atol=1e-6)
        out.backward(torch.from_numpy(g["mha_%s/go" % kind]))
        close(q.grad, g["mha_%s/dq" % kind], atol=2e-5)
        if mem is not None:
            close(mem.grad, g["mha_cross/dmem"], atol=2e-5)
        for k, p in P.items():
            close(p.grad, g["mha_%s/dw/%s" % (kind, k[2:])], atol=5e-5)
    P = {"f." + k.split("/w/")[1]: torch.from_numpy(v) for k, v in g.items() if k.startswith("ffn/w/")}
    close(O.ffn(P, "f", torch.from_numpy(g["ffn/x"])), g["ffn/out"])

    cfg = make_config(TINY)
    P = O.to_torch_state(synth.synthetic_state(cfg, 1234))
    b = O.to_torch_batch(synth.synthetic_batch(cfg, B=2, S=9, T=14, seed=3, in_lens=[9, 5], tgt_lens=[14, 8]))
    with torch.no_grad():
        close(O.prenet(P, cfg, b["mel_targets"]), g["prenet/out"])
        enc = O.encoder_forward(P, cfg, b["inputs"], b["input_lengths"], b["input_spk_ids"], b["input_language_vecs"])
        close(enc, g["encoder/out"])
        for lo in (0, 1):
            mels, stop, al = O.decoder_forward(P, cfg, enc, b["input_lengths"], b["mel_targets"],
                                               b["target_lengths"], leave_one=bool(lo))
            close(mels, g["decoder_lo%d/mels" % lo]); close(stop, g["decoder_lo%d/stop" % lo])
            close(al["encdec"][1], g["decoder_lo%d/align_encdec_1" % lo], atol=1e-6)
        close(O.postnet_forward(P, cfg, b["mel_targets"], b["target_lengths"], train=False), g["postnet_eval/out"])
        bn = {}
        close(O.postnet_forward(P, cfg, b["mel_targets"], b["target_lengths"], train=True, bn_state=bn),
              g["postnet_train/out"], atol=2e-5)
        for i in range(cfg.n_postnet_layer):
            close(bn["postnet.batchnorm_layers.%d.running_mean" % i], g["postnet_train/running_mean_%d" % i], atol=1e-6)
            close(bn["postnet.batchnorm_layers.%d.running_var" % i], g["postnet_train/running_var_%d" % i], atol=1e-6)


# ------------------------------------------------------------------ G4 decode
@pytest.mark.parametrize("tag,over", [("tiny", TINY), ("tiny96", TINY96)])
@pytest.mark.parametrize("case", ["never", "mixed", "first"])
def test_g4_decode(tag, over, case):
    g = load("g4_decode")
    cfg = make_config(over + ",max_generation_frames=40")
    st = synth.synthetic_state(cfg, 1234)
    bias = {"never": -100.0, "first": 100.0}.get(case)
    if bias is None:
        bias = float(g["%s_%s/stop_bias" % (tag, case)])
    st["decoder.stop_net.bias"] = np.full((1,), bias, dtype=np.float32)
    P = O.to_torch_state(st)
    nb = synth.synthetic_batch(cfg, B=3, S=10, T=4, seed=11, in_lens=[10, 6, 8])
    nb.pop("mel_targets"); nb.pop("target_lengths")
    r = O.eval_batch(P, cfg, O.to_torch_batch(nb))
    pre = "%s_%s/" % (tag, case)
    assert r["generated_lengths"].tolist() == g[pre + "generated_lengths"].tolist()
    close(r["mel_pre"], g[pre + "mel_pre"], atol=2e-5); close(r["mel_aft"], g[pre + "mel_aft"], atol=2e-5)
    for i in range(cfg.n_decoder_layer):
        am = r["alignments"]["encdec"][i].argmax(dim=2).numpy()
        assert (am == g[pre + "align_argmax_%d" % i]).all()


# ------------------------------------------------------------------ G9 full-size decode (BASELINE configs[3] at CPU size)
G9_OVER = "transformer_dropout_rate=0.0,decoder_dropout_rate=0.0,max_generation_frames=48"


def g9_inputs(g):
    cfg = make_config(G9_OVER)
    st = synth.synthetic_state(cfg, 77)
    st["decoder.stop_net.bias"] = np.full((1,), float(g["stop_bias"]), dtype=np.float32)
    nb = synth.synthetic_batch(cfg, B=3, S=40, T=4, seed=21, in_lens=[40, 31, 36], n_spk=572, n_lang=38)
    nb.pop("mel_targets"); nb.pop("target_lengths")
    return cfg, st, nb


def check_align_argmax(aligns, g, n_layers, min_margin=1e-4):
    """encoder-decoder alignment arg-max equals the reference's wherever the reference's own top-2 margin exceeds min_margin
    (at random init some frames attend almost uniformly; a tie-break there is not a property of either implementation)."""
    for i in range(n_layers):
        am = np.asarray(aligns[i]).argmax(axis=2)
        ok = (am == g["align_argmax_%d" % i]) | (g["align_margin_%d" % i] < min_margin)
        assert ok.all(), "layer %d: %d arg-max positions differ" % (i, int((~ok).sum()))


def test_g9_decode_fullsize():
    g = load("g9_decode_fullsize")
    cfg, st, nb = g9_inputs(g)
    r = O.eval_batch(O.to_torch_state(st), cfg, O.to_torch_batch(nb))
    assert r["generated_lengths"].tolist() == g["generated_lengths"].tolist() and len(set(g["generated_lengths"].tolist())) == 3
    close(r["mel_pre"], g["mel_pre"], atol=5e-5); close(r["mel_aft"], g["mel_aft"], atol=5e-5)
    check_align_argmax([a.numpy() for a in r["alignments"]["encdec"]], g, cfg.n_decoder_layer)


# ------------------------------------------------------------------ G5 full size (slow-ish: ~10 s)
def test_g5_fullsize():
    g = load("g5_fullsize")
    cfg = make_config("transformer_dropout_rate=0.0,decoder_dropout_rate=0.0")
    P = O.to_torch_state(synth.synthetic_state(cfg, 4321), requires_grad=True)
    b = O.to_torch_batch(synth.synthetic_batch(cfg, B=4, S=100, T=600, seed=0, in_lens=[100, 90, 80, 70],
                                               tgt_lens=[600, 550, 500, 450], n_spk=572, n_lang=38))
    o = O.tacotron_forward(P, cfg, b, train=True)
    losses = O.compute_loss(P, cfg, b["mel_targets"], b["target_lengths"], o)
    names = [n for n in P if O.is_parameter(n)]
    grads = torch.autograd.grad(losses["loss"], [P[n] for n in names], allow_unused=True)
    close(o["mel_bef"][:, :4, :8], g["mel_bef_slice"], atol=2e-5)
    close(o["mel_aft"][:, :4, :8], g["mel_aft_slice"], atol=2e-5)
    close(o["mel_bef"][:, 440:452, :4], g["mel_bef_tail"], atol=2e-5)
    close(o["stop_logits"][:, :16], g["stop_slice"], atol=2e-5)
    assert (o["stop_logits"].argmax(-1).numpy() == g["stop_argmax"]).all()
    al = o["alignments"]["encdec"][5]
    assert (al.argmax(2)[:, :, ::25].numpy() == g["align_encdec5_argmax"]).all()
    for k in ("loss", "bef_loss", "aft_loss", "aft_losses", "mse_loss", "l2", "stop_loss"):
        close(losses[k], g["loss_" + k], atol=1e-6, rtol=1e-5)
    for n, gr in zip(names, grads):
        ref = float(g["gnorm/" + n])
        mine = 0.0 if gr is None else float(gr.double().norm())
        assert abs(mine - ref) <= 2e-4 * ref + 1e-8, (n, mine, ref)


# ------------------------------------------------------------------ G6
def test_g6_lr_schedule():
    g = load("g6_misc")
    cfg = make_config("")
    for s, v in zip(g["lr_steps"], g["lr_values"]):
        assert O.learning_rate_schedule(int(s), cfg) == pytest.approx(float(v), rel=1e-12)
