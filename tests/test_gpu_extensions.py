"""GPU parity of the north-star extensions (guided-attention loss, frozen encoder) against the oracle's
restatement (tests/test_extensions.py pins that restatement by properties; the reference has no counterpart).
Tolerances as in test_gpu_model.py: fp32 mode, losses 2e-4 relative, gradients 2e-4 * max(1, |g|)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import b2s_oracle as O                     # checker only
from oracle import synth, make_config, TINY, TINY96
from gpu_util import DEV
from test_gpu_model import build, dev_batch

EXT = ",guided_attention_weight=2.0,guided_attention_sigma=0.3"


def _batch(cfg):
    return synth.synthetic_batch(cfg, B=3, S=11, T=23, seed=7, in_lens=[11, 7, 4], tgt_lens=[23, 15, 9])


@pytest.mark.parametrize("tag,over", [("tiny", TINY), ("tiny96", TINY96)])
def test_guided_attention_loss_and_grads_fp32(tag, over):
    from transformer.tacotron import compute_loss
    m, cfg, st, hp = build(over + EXT)
    nb = _batch(cfg)
    b = dev_batch(nb)
    m.train()
    o = m(**b)
    losses = compute_loss(m, b["mel_targets"], b["target_lengths"], o, hp)
    losses["loss"].backward()
    torch.cuda.synchronize()
    P = O.to_torch_state(st, requires_grad=True)
    ob = O.to_torch_batch(nb)
    out = O.tacotron_forward(P, cfg, ob, train=True, bn_state={})
    ref = O.compute_loss(P, cfg, ob["mel_targets"], ob["target_lengths"], out, ob["input_lengths"])
    names = [n for n in P if O.is_parameter(n)]
    gl = torch.autograd.grad(ref["loss"], [P[n] for n in names], allow_unused=True)
    for k in ("loss", "ga_loss", "bef_loss", "stop_loss"):
        assert abs(float(losses[k]) - float(ref[k])) < 1e-5 + 2e-4 * abs(float(ref[k])), (k, float(losses[k]), float(ref[k]))
    # the value also equals the published formula evaluated on the alignments the model returns
    al = [a.cpu() for a in o["alignments"]["encdec"]]
    direct = cfg.guided_attention_weight * float(O.guided_attention_loss(al, ob["input_lengths"], ob["target_lengths"],
                                                                         cfg.guided_attention_sigma))
    assert abs(direct - float(losses["ga_loss"])) < 1e-5
    bad = []
    params = dict(m.named_parameters())
    for n, g in zip(names, gl):
        g = g if g is not None else torch.zeros_like(P[n])
        got = params[n].grad.detach().cpu()
        err = float((got - g).abs().max())
        if err > 2e-4 * max(1.0, float(g.norm())):
            bad.append((n, err, float(g.norm())))
    assert not bad, bad[:8]
    # the guided term really contributes to the gradient being compared
    n = "decoder.decoder.encdec_attentions.0.q_transform.weight"
    P0 = O.to_torch_state(st, requires_grad=True)
    cfg0 = make_config(over)
    out0 = O.tacotron_forward(P0, cfg0, ob, train=True, bn_state={})
    g0 = torch.autograd.grad(O.compute_loss(P0, cfg0, ob["mel_targets"], ob["target_lengths"], out0)["loss"], [P0[n]])[0]
    assert float((g0 - gl[names.index(n)]).abs().max()) > 1e-4


def test_guided_term_without_gradient_request():
    """The guided loss is reported but not differentiated (loss without it is back-propagated): gradients equal the
    weight-0 model's."""
    from transformer.tacotron import compute_loss
    grads = []
    for over in (TINY + EXT, TINY):
        m, cfg, st, hp = build(over)
        b = dev_batch(_batch(cfg))
        m.train()
        o = m(**b)
        losses = compute_loss(m, b["mel_targets"], b["target_lengths"], o, hp)
        base = losses["bef_loss"] + losses["aft_loss"] + losses["l2"] + losses["stop_loss"]
        base.backward()
        torch.cuda.synchronize()
        grads.append({n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()})
    for n in grads[0]:
        assert float((grads[0][n] - grads[1][n]).abs().max()) < 1e-6, n


@pytest.mark.parametrize("tag,over", [("tiny", TINY), ("tiny96", TINY96)])
def test_fused_trainer_guided_frozen_encoder(tag, over):
    """configs[4] of BASELINE.json in miniature: frozen encoder + guided-attention loss through the fused trainer,
    three steps against the oracle; encoder parameters stay bit-identical."""
    from b2s_hip.trainer import HipTrainer
    m, cfg, st, hp = build(over + EXT + ",freeze_encoder=true")
    assert not any(p.requires_grad for p in m.encoder.parameters())
    m.train()
    nb = _batch(cfg)
    b = dev_batch(nb)
    tr = HipTrainer(m, hp)
    P = O.to_torch_state(st, requires_grad=True)
    ob = O.to_torch_batch(nb)
    opt = {}
    for step in range(3):
        vals = tr.train_step(b)
        _, losses, _ = O.train_step(P, cfg, ob, opt, step, train=True)
        torch.cuda.synchronize()
        v = vals.cpu().numpy()
        for i, k in enumerate(("loss", "bef_loss", "aft_loss", "mse_loss", "l2", "stop_loss")):
            assert abs(float(v[i]) - float(losses[k])) < 2e-4 + 2e-4 * abs(float(losses[k])), (step, k, float(v[i]), float(losses[k]))
        assert abs(float(tr.last_ga_loss) - float(losses["ga_loss"])) < 1e-5 + 2e-4 * float(losses["ga_loss"])
    sd = m.state_dict()
    worst = 0.0
    for n, ref in P.items():
        got = sd[n].detach().cpu().double()
        ref = ref.detach().double()
        if n.endswith("num_batches_tracked"):
            assert int(got) == int(ref)
            continue
        if n.startswith("encoder."):
            assert torch.equal(sd[n].detach().cpu(), torch.from_numpy(np.array(st[n]))), n
            continue
        assert abs(float(got.norm()) - float(ref.norm())) <= 2e-4 * float(ref.norm()) + 1e-5, n
        worst = max(worst, float((got - ref).abs().max()))
    assert worst < 6.5e-3, worst


def test_guided_autograd_frozen_encoder_drop_in_loop():
    """train.py-style loop (torch.optim.Adam over the parameters that require grad) with the frozen encoder."""
    from transformer.tacotron import compute_loss
    m, cfg, st, hp = build(TINY + EXT + ",freeze_encoder=true")
    b = dev_batch(_batch(cfg))
    m.train()
    optim = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=hp.max_lr, eps=hp.adam_eps)
    first = None
    for step in range(4):
        o = m(**b)
        losses = compute_loss(m, b["mel_targets"], b["target_lengths"], o, hp)
        optim.zero_grad()
        losses["loss"].backward()
        optim.step()
        first = float(losses["loss"]) if first is None else first
    assert float(losses["loss"]) < first
    for n, p in m.encoder.named_parameters():
        assert p.grad is None
        assert torch.equal(p.detach().cpu(), torch.from_numpy(np.array(st["encoder." + n]))), n


def test_guided_bf16_dropout_on_full_heads():
    """bf16 + dropout on, default head size 96 at a multi-tile length: the reported loss equals the formula evaluated
    on the (pre-dropout) alignments the model returns, and the step runs finite."""
    from transformer.tacotron import compute_loss
    over = TINY96.replace("transformer_dropout_rate=0.0", "transformer_dropout_rate=0.1") + EXT
    m, cfg, st, hp = build(over, compute_dtype="bf16")
    nb = synth.synthetic_batch(cfg, B=4, S=70, T=150, seed=3, in_lens=[70, 64, 33, 9], tgt_lens=[150, 128, 65, 17])
    b = dev_batch(nb)
    m.train()
    o = m(**b)
    losses = compute_loss(m, b["mel_targets"], b["target_lengths"], o, hp)
    losses["loss"].backward()
    torch.cuda.synchronize()
    ob = O.to_torch_batch(nb)
    al = [a.cpu() for a in o["alignments"]["encdec"]]
    direct = cfg.guided_attention_weight * float(O.guided_attention_loss(al, ob["input_lengths"], ob["target_lengths"],
                                                                         cfg.guided_attention_sigma))
    assert abs(direct - float(losses["ga_loss"])) < 2e-3 * direct, (direct, float(losses["ga_loss"]))
    for n, p in m.named_parameters():
        assert torch.isfinite(p.grad).all(), n


def test_dropout_on_decode_statistics_vs_recompute_loop():
    """The reference synthesises with decoder.train() (eval.py:116-117): prenet dropout 0.5 and transformer dropout 0.1 are
    live and, because its loop recomputes the whole prefix every frame (synthesize.py:35-45), every earlier position gets
    fresh masks at every step.  The KV-cached loop draws each position's masks once (when that frame is generated) and
    keeps its K / V.  RNG streams differ anyway, so the comparison is statistical: 256 copies of ONE utterance (the only
    randomness is dropout) through the HIP hipGraph loop and through the oracle's recompute-every-frame loop on the same
    weights; the distributions of generated length and of the per-frame mel mean / spread must agree within sampling
    noise + the bounds below (the measured numbers are printed and quoted in INTEGRATION.md section 3)."""
    import synthesize
    from test_gpu_model import build, dev_batch, load
    g = load("g4_decode")
    over = TINY96.replace("transformer_dropout_rate=0.0,decoder_dropout_rate=0.0", "transformer_dropout_rate=0.1,decoder_dropout_rate=0.5")
    assert over != TINY96
    bias = float(g["tiny96_mixed/stop_bias"]) + 0.6

    def edit(st):
        st["decoder.stop_net.bias"] = np.full((1,), bias, dtype=np.float32)
    m, cfg, st, hp = build(over + ",max_generation_frames=40", state_edit=edit)
    m.eval()
    m.decoder.train()
    B = 256
    one = synth.synthetic_batch(cfg, B=1, S=10, T=4, seed=11, in_lens=[10])
    nb = {k: (np.repeat(np.asarray(v), B, axis=0) if not isinstance(v, list) else v * B) for k, v in one.items()}
    nb.pop("mel_targets"); nb.pop("target_lengths")
    r = synthesize.eval_batch(m, dev_batch(nb), use_bar=False, bar_interval=-1)
    P = O.to_torch_state(st)
    torch.manual_seed(0)
    ro = O.eval_batch(P, cfg, O.to_torch_batch(nb), decoder_train=True)
    lh = np.minimum(np.asarray(r["generated_lengths"], dtype=np.float64), 41)
    lo = np.minimum(ro["generated_lengths"].numpy().astype(np.float64), 41)
    # per-frame statistics over utterances still running at that frame
    def frame_stats(mel, lens, t):
        alive = lens > t
        x = mel[alive, t, :]
        return alive.mean(), x.mean(), x.std()
    mh, mo = r["mel_pre"], ro["mel_pre"].numpy()
    T = min(mh.shape[1], mo.shape[1])
    rows = []
    for t in range(0, T, 4):
        ah, uh, sh = frame_stats(mh, lh, t)
        ao, uo, so = frame_stats(mo, lo, t)
        rows.append((t, ah, ao, uh, uo, sh, so))
    print("dropout-on decode, 256 x one utterance: generated length HIP %.2f +- %.2f | oracle (recompute loop) %.2f +- %.2f"
          % (lh.mean(), lh.std(), lo.mean(), lo.std()))
    for t, ah, ao, uh, uo, sh, so in rows:
        print("  frame %2d: alive %.2f / %.2f  mel mean %+.4f / %+.4f  mel std %.4f / %.4f" % (t, ah, ao, uh, uo, sh, so))
    se = np.sqrt(lh.var() / B + lo.var() / B)
    assert abs(lh.mean() - lo.mean()) < 4 * se + 0.1 * lo.mean(), (lh.mean(), lo.mean())
    assert 0.6 < (lh.std() + 0.5) / (lo.std() + 0.5) < 1.67
    for t, ah, ao, uh, uo, sh, so in rows:
        if min(ah, ao) > 0.3:
            assert abs(uh - uo) < 0.25 * max(so, sh) + 0.02, (t, uh, uo)
            assert 0.7 < sh / so < 1.43, (t, sh, so)
