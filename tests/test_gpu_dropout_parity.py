"""Dropout-ON parity: the arithmetic bench.py actually times (the reference trains AND synthesises with dropout live:
transformer/modules.py:18,55,64,67,120,132,138,141, attention.py:89, tacotron.py:58,62,89, eval.py:116-117).

The reference draws its masks from torch's global generator, the engine from a counter RNG -- the streams cannot match.  So the
comparison runs the other way round: the CPU oracle takes the ENGINE's masks (oracle/rng.py: DeviceMasks -- the hash is pinned bit for
bit by test_gpu_ops.py::test_dropout_mask_matches_host_restatement, the op-id table is read from the library through
b2s_dropout_site, nothing is restated here) and everything downstream of a mask -- forward, the 7 loss terms, every parameter gradient
incl. the masks regenerated in backward, LayerNorm-backward's emitted dY operands, ReLU + dropout epilogues, prenet / postnet p = 0.5,
the fused encoder's replay, padded-tile skipping -- is held to the fp32 bars (mel <= 1e-3 abs, losses 2e-4, per-tensor gradient
relative L2 <= 1e-3) and, in bf16, to the recorded-drift gates.  A deliberately shifted op id at ONE site must make the test fail."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import b2s_oracle as O                     # checker only
from oracle import rng as R
from oracle import synth, make_config, TINY, TINY96
from gpu_util import DEV, record_drift, drift_gate
from test_gpu_model import build, dev_batch, load

DROP = "transformer_dropout_rate=0.1,decoder_dropout_rate=0.5"         # the reference's rates (hyperparams.py:33,35)


def with_dropout(over):
    out = over.replace("transformer_dropout_rate=0.0,decoder_dropout_rate=0.0", DROP)
    assert out != over
    return out


def site_info(site, layer, decode):
    """(op id, index kind, frame-salt rule) of a dropout site, from the library's own table."""
    from b2s_hip import lib as L
    op, kind, salt = C.c_uint32(), C.c_int(), C.c_int()
    L.check(L.load().b2s_dropout_site(site.encode(), int(layer), int(bool(decode)), C.byref(op), C.byref(kind), C.byref(salt)))
    return op.value, kind.value, salt.value


def oracle_step(cfg, st, nb, seeds, overrides=None):
    """Oracle forward + loss + gradients with the engine's masks."""
    P = O.to_torch_state(st, requires_grad=True)
    ob = O.to_torch_batch(nb)
    src = R.DeviceMasks(seeds, site_info, overrides=overrides)
    with O.device_masks(src):
        out = O.tacotron_forward(P, cfg, ob, train=True)
    losses = O.compute_loss(P, cfg, ob["mel_targets"], ob["target_lengths"], out)
    names = [n for n in P if O.is_parameter(n)]
    gl = torch.autograd.grad(losses["loss"], [P[n] for n in names], allow_unused=True)
    grads = {n: (g.detach().float() if g is not None else torch.zeros_like(P[n])) for n, g in zip(names, gl)}
    return out, {k: float(v) for k, v in losses.items() if v.dim() == 0}, grads, src


def worst_direction(g, ref):
    we = (0.0, None)
    for n, r in ref.items():
        rn = float(r.double().norm())
        if rn < 1e-9:
            continue
        e = float((g[n].double().reshape(-1) - r.double().reshape(-1)).norm()) / rn
        if e > we[0]:
            we = (e, n)
    return we


def hip_autograd_step(over, st, nb, compute_dtype="fp32"):
    from transformer.tacotron import compute_loss
    m, cfg, _, hp = build(over, compute_dtype=compute_dtype, state_edit=lambda s: s.update(st))
    b = dev_batch(nb)
    m.train()
    o = m(**b)
    losses = compute_loss(m, b["mel_targets"], b["target_lengths"], o, hp)
    losses["loss"].backward()
    torch.cuda.synchronize()
    seeds = dict(m.engine().seeds_used)
    grads = {n: (p.grad.detach().float().cpu() if p.grad is not None else torch.zeros_like(p).cpu()) for n, p in m.named_parameters()}
    outs = {k: o[k].detach().float().cpu() for k in ("mel_bef", "mel_aft", "stop_logits")}
    outs["align"] = {k: [a.cpu() for a in v] for k, v in o["alignments"].items()}
    ls = {k: float(losses[k]) for k in ("loss", "bef_loss", "aft_loss", "mse_loss", "l2", "stop_loss")}
    return cfg, outs, ls, grads, seeds


def check_fp32(tag, outs, ls, grads, ref_out, ref_l, ref_g, mel_tol):
    for k in ("mel_bef", "mel_aft", "stop_logits"):
        d = float((outs[k] - ref_out[k].detach()).abs().max())
        print("%s dropout on, fp32: max|%s - oracle| = %.2e" % (tag, k, d))
        assert d < mel_tol, (tag, k, d)
    for k in ("loss", "bef_loss", "aft_loss", "mse_loss", "l2", "stop_loss"):
        assert abs(ls[k] - ref_l[k]) <= 2e-4 * abs(ref_l[k]) + 1e-6, (tag, k, ls[k], ref_l[k])
    # (1-element parameters -- the two pe_scale, stop_net.bias -- are cancelling sums over all tokens accumulated with fp32 atomics: run-to-run
    # spread of a few 1e-4 at the full-size shape; they get the 5e-3 bar the dropout-off tests give them)
    e, n = worst_direction({k: v for k, v in grads.items() if v.numel() > 1}, {k: v for k, v in ref_g.items() if v.numel() > 1})
    e1, n1 = worst_direction({k: v for k, v in grads.items() if v.numel() == 1}, {k: v for k, v in ref_g.items() if v.numel() == 1})
    print("%s dropout on, fp32: worst per-tensor relative L2 gradient error %.3e (%s); 1-element tensors %.3e (%s)" % (tag, e, n, e1, n1))
    assert e < 1e-3 and e1 < 5e-3, (tag, e, n, e1, n1)


@pytest.mark.parametrize("tag,over", [("tiny", TINY), ("tiny96", TINY96)])
def test_dropout_on_forward_loss_grads_fp32_tiny(tag, over):
    """The two tiny golden configurations (head sizes 32 / 64 and 64 / 96: every attention goes through the fused kernels),
    reference rates 0.1 / 0.5, the golden batch: outputs, alignments (pre-dropout: attention.py:88), losses, every gradient."""
    over = with_dropout(over)
    cfg0 = make_config(over)
    st = synth.synthetic_state(cfg0, 1234)
    nb = synth.synthetic_batch(cfg0, B=3, S=11, T=23, seed=7, in_lens=[11, 7, 4], tgt_lens=[23, 15, 9])
    cfg, outs, ls, grads, seeds = hip_autograd_step(over, st, nb)
    assert set(seeds) >= {"encoder", "decoder", "postnet"}
    ref_out, ref_l, ref_g, src = oracle_step(cfg, st, nb, seeds)
    assert len(src.calls) == 1 + 4 * cfg.n_encoder_layer + 3 + 6 * cfg.n_decoder_layer + cfg.n_postnet_layer    # every reference dropout call took a device mask
    check_fp32(tag, outs, ls, grads, ref_out, ref_l, ref_g, 2e-4)
    for kind in ("self", "encdec"):
        for i in range(cfg.n_decoder_layer):
            a, r = outs["align"][kind][i], ref_out["alignments"][kind][i].detach()
            assert float((a - r).abs().max()) < 1e-5 and bool((a.argmax(2) == r.argmax(2)).all()), (kind, i)
    # the dropout really is on: the same step without masks is far away
    P = O.to_torch_state(st)
    with torch.no_grad():
        plain = O.tacotron_forward(P, make_config(over.replace(DROP, "transformer_dropout_rate=0.0,decoder_dropout_rate=0.0")), O.to_torch_batch(nb), train=True)
    assert float((plain["mel_bef"] - outs["mel_bef"]).abs().max()) > 0.05


def test_shifted_op_id_at_one_site_is_noticed():
    """Test of the test: the oracle with ONE site's op id off by one (a different, equally valid mask at that site only) no longer agrees
    with the device -- for a residual site of the decoder, an attention-weight site of the encoder and the postnet."""
    over = with_dropout(TINY96)
    cfg0 = make_config(over)
    st = synth.synthetic_state(cfg0, 1234)
    nb = synth.synthetic_batch(cfg0, B=3, S=11, T=23, seed=7, in_lens=[11, 7, 4], tgt_lens=[23, 15, 9])
    cfg, outs, ls, grads, seeds = hip_autograd_step(over, st, nb)
    for site, layer, key in (("decoder.ffn_res", 1, "mel_bef"), ("encoder.attn", 0, "mel_bef"), ("postnet.conv", 2, "mel_aft")):
        op = site_info(site, layer, False)[0]
        ref_out, ref_l, ref_g, _ = oracle_step(cfg, st, nb, seeds, overrides={(site, layer): op + 1})
        d = float((outs[key] - ref_out[key].detach()).abs().max())
        e, n = worst_direction(grads, ref_g)
        print("op id of %s[%d] shifted: max|%s - oracle| = %.2e, worst gradient error %.2e" % (site, layer, key, d, e))
        assert d > 5e-3 and e > 1e-2, (site, d, e)


def _fullsize(shape, seed_state=11, seed_batch=2):
    over = DROP
    cfg = make_config(over)
    st = synth.synthetic_state(cfg, seed_state)
    B, S, T, n_spk, n_lang = shape
    nb = synth.synthetic_batch(cfg, B, S, T, seed=seed_batch, n_spk=n_spk, n_lang=n_lang)
    return over, cfg, st, nb


def _trainer_step(over, st, batch, compute_dtype):
    from b2s_hip.trainer import HipTrainer
    m, cfg, _, hp = build(over, compute_dtype=compute_dtype, state_edit=lambda s: s.update(st))
    m.train()
    tr = HipTrainer(m, hp)
    v = tr.train_step(batch).cpu().numpy()
    torch.cuda.synchronize()
    grads = {n: tr.eng.grad_view(n).detach().float().cpu().clone() for n, _ in m.named_parameters()}
    ls = dict(zip(("loss", "bef_loss", "aft_loss", "mse_loss", "l2", "stop_loss"), [float(x) for x in v[:6]]))
    seeds = dict(tr.eng.seeds_used)
    del tr, m
    return ls, grads, seeds


@pytest.mark.parametrize("tag,shape", [("lj", (14, 114, 582, 1, 1)), ("c3", (4, 256, 300, 572, 38))])
def test_dropout_on_fullsize_vs_oracle(tag, shape):
    """Default hparams, dropout 0.1 / 0.5, against the fp32 oracle under the engine's masks.  lj: BASELINE configs[1] exactly as bench.py times
    it (B = 14, S = 114, T = 582: resident-key encoder-decoder attention, fused encoder sublayers in bf16); c3: the configs[2] row shape
    (S = 256 text bytes, 572 speakers / 38 languages, at B = 4, T = 300 to keep the CPU oracle affordable: generic attention kernels,
    the kernel-by-kernel encoder).  fp32 mode through the module surface (outputs + losses + gradients at the fp32 bars) and through
    the fused trainer (padded query tiles skipped); bf16 mode through the fused trainer at the recorded-drift gates."""
    over, cfg, st, nb = _fullsize(shape)
    _, outs, ls, grads, seeds = hip_autograd_step(over, st, nb)
    ref_out, ref_l, ref_g, _ = oracle_step(cfg, st, nb, seeds)
    check_fp32("%s shape (module surface)" % tag, outs, ls, grads, ref_out, ref_l, ref_g, 1e-3)
    batch = dev_batch(nb)
    for mode in ("fp32", "bf16"):
        ls, grads, seeds = _trainer_step(over, st, batch, mode)
        _, ref_l, ref_g, _ = oracle_step(cfg, st, nb, seeds)
        lworst = max(abs(ls[k] - ref_l[k]) / (abs(ref_l[k]) + 1e-9) for k in ("loss", "bef_loss", "aft_loss", "stop_loss", "l2"))
        e, n = worst_direction({k: v for k, v in grads.items() if v.numel() > 1}, {k: v for k, v in ref_g.items() if v.numel() > 1})
        print("%s shape dropout on, fused trainer %s: worst loss-term error %.3e, worst per-tensor relative L2 gradient error %.3e (%s)" % (tag, mode, lworst, e, n))
        if mode == "fp32":
            assert lworst < 2e-4 and e < 1e-3, (lworst, e, n)
        else:
            record_drift(tag + "_dropout/loss_terms_rel", lworst)
            record_drift(tag + "_dropout/worst_grad_dir_rel", e)
            assert lworst < drift_gate(tag + "_dropout/loss_terms_rel", 0.01, floor=1e-3) and e < drift_gate(tag + "_dropout/worst_grad_dir_rel", 0.15, floor=0.05), (lworst, e, n)


def test_dropout_on_decode_replayed_against_recompute_loop():
    """synthesize.eval_batch with decoder.train() (eval.py:116-117) in fp32 on the tiny model, stop bias inside the logit range: the KV-cached
    hipGraph loop draws every position's masks once, in the frame that generates it.  The oracle's loop is the reference's algorithm
    (synthesize.py:35-45: the whole prefix recomputed every frame) given the SAME per-frame masks for every prefix row -- equal to the
    cached loop by causality.  Generated lengths exact, mels <= 1e-3, encoder-decoder alignment arg-max exact."""
    import synthesize
    g = load("g4_decode")
    over = with_dropout(TINY96)
    bias = float(g["tiny96_mixed/stop_bias"]) + 0.6

    def edit(st):
        st["decoder.stop_net.bias"] = np.full((1,), bias, dtype=np.float32)
    m, cfg, st, hp = build(over + ",max_generation_frames=40", state_edit=edit)
    m.eval()
    m.decoder.train()
    nb = synth.synthetic_batch(cfg, B=6, S=10, T=4, seed=11, in_lens=[10, 9, 8, 10, 7, 6])
    nb.pop("mel_targets"); nb.pop("target_lengths")
    r = synthesize.eval_batch(m, dev_batch(nb), use_bar=False, bar_interval=-1, keep_self_alignments=True)
    seed = m.engine().seeds_used["decode"]
    src = R.DeviceMasks({"decoder": seed}, site_info, decode=True)
    P = O.to_torch_state(st)
    with O.device_masks(src):
        ro = O.eval_batch(P, cfg, O.to_torch_batch(nb), decoder_train=True)
    lh, lo = [int(x) for x in r["generated_lengths"]], [int(x) for x in ro["generated_lengths"]]
    print("dropout-on decode replay: generated lengths HIP %s | oracle %s" % (lh, lo))
    assert lh == lo
    assert len(set(lh)) >= 3, "stop frames are not staggered: %s" % lh
    mo = ro["mel_pre"].numpy()
    assert r["mel_pre"].shape == mo.shape
    d = float(np.abs(r["mel_pre"] - mo).max())
    print("dropout-on decode replay: max|mel_pre - oracle| = %.2e" % d)
    assert d < 1e-3
    assert float(np.abs(r["mel_aft"] - ro["mel_aft"].numpy()).max()) < 1e-3
    for l in range(cfg.n_decoder_layer):
        a, b = r["alignments"]["encdec"][l], ro["alignments"]["encdec"][l].numpy()
        assert a.shape == b.shape and np.abs(a - b).max() < 1e-4
        assert (a.argmax(2) == b.argmax(2)).all()
