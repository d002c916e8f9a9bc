"""Ragged decoder rows (include/b2s_hip.h: b2s_decoder_compact_rows): with the host copy of target_lengths in the batch the fused trainer runs the
decoder segment over sum(target_lengths) token rows instead of B x T.  The reference computes the padded rows and masks them out
(transformer/common.py:51-70, modules.py:142-144, tacotron.py:112-115), so nothing observable may change:

  * dropout off: the losses and the mel / stop outputs are BIT-IDENTICAL with and without the ragged layout (padded rows exactly zero), every
    gradient agrees to the fp32 summation order of its weight-gradient K walk, and so do the parameters after the steps;
  * dropout on: the ragged step matches the CPU oracle run under the engine's own masks (oracle/rng.py: DeviceMasks with the ragged row index),
    fp32 mode at the 1e-3 bar, and a ragged index that is off is noticed.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import b2s_oracle as O
from oracle import rng as R
from oracle import synth, make_config, TINY, TINY96
from test_gpu_model import build, dev_batch
from test_gpu_dropout_parity import with_dropout, site_info, worst_direction


def _trainer_steps(over, st, nb, compute_dtype, ragged, steps=2):
    from b2s_hip.trainer import HipTrainer
    m, cfg, _, hp = build(over, compute_dtype=compute_dtype, state_edit=lambda s: s.update(st))
    m.train()
    tr = HipTrainer(m, hp, dist=False)
    b = dev_batch(nb)
    if ragged:
        b["target_lengths_host"] = [int(x) for x in nb["target_lengths"]]
    grads, vals = [], []
    tr.grad_probe = lambda flat, wire: grads.append(flat.detach().clone())
    for _ in range(steps):
        vals.append(tr.train_step(b).detach().clone())
    torch.cuda.synchronize()
    names = list(tr.eng.param_offsets)
    g0 = {n: grads[0][tr.eng.param_offsets[n][0]:tr.eng.param_offsets[n][0] + tr.eng.param_offsets[n][1]].cpu() for n in names}
    params = {n: p.detach().float().cpu() for n, p in m.named_parameters()}
    return vals, g0, params, dict(tr.eng.seeds_used), cfg, m


@pytest.mark.parametrize("compute_dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("tag,over,lens", [("tiny", TINY, [23, 15, 9]), ("tiny96", TINY96, [17, 23, 1]), ("tiny96-long", TINY96, None)])
def test_ragged_rows_change_nothing_observable(tag, over, lens, compute_dtype):
    cfg0 = make_config(over)
    st = synth.synthetic_state(cfg0, 1234)
    if lens is None:            # rows that cross the attention kernels' 64- / 128-row tiles, ragged ends inside and at tile boundaries
        nb = synth.synthetic_batch(cfg0, B=5, S=37, T=200, seed=3, in_lens=[37, 30, 11, 37, 5], tgt_lens=[200, 129, 64, 63, 1])
    else:
        nb = synth.synthetic_batch(cfg0, B=3, S=11, T=23, seed=7, in_lens=[11, 7, 4], tgt_lens=lens)
    va, ga, pa, _, _, _ = _trainer_steps(over, st, nb, compute_dtype, ragged=False)
    vb, gb, pb, _, _, _ = _trainer_steps(over, st, nb, compute_dtype, ragged=True)
    # The whole step is not bit-reproducible run to run in EITHER layout (fp32 atomics in the BatchNorm statistics, the bias / LayerNorm parameter
    # sums and the L2 term; in bf16 a last-bit difference there moves roundings downstream), so the trainer-level comparison has the bars of two
    # runs of one layout; bit-identity of the segment itself is test_ragged_outputs_are_bit_identical_and_zero_on_padded_rows below.
    for i in range(va[0].numel()):
        assert abs(float(va[0][i]) - float(vb[0][i])) <= (1e-6 if compute_dtype == "fp32" else 2e-5) * abs(float(va[0][i])) + 1e-9, (i, va[0], vb[0])
    # gradients: same values up to the order in which a weight gradient's K walk meets the rows
    e, n = worst_direction(gb, ga)
    print("%s %s: worst per-tensor gradient difference ragged vs padded %.2e (%s)" % (tag, compute_dtype, e, n))
    assert e < (2e-5 if compute_dtype == "fp32" else 1e-2), (e, n)
    for k in pa:
        d = float((pa[k] - pb[k]).abs().max())
        # (two Adam steps at lr 1e-3: an element whose gradient is ~0 moves by up to lr whatever the sign of the rounding noise)
        assert d <= (3e-4 if compute_dtype == "fp32" else 4e-3), (k, d)


def test_ragged_outputs_are_bit_identical_and_zero_on_padded_rows():
    """The segment call itself: mel / stop outputs of b2s_decoder_forward with and without the hand-over, and d(memory) of its backward."""
    over = TINY96
    cfg0 = make_config(over)
    st = synth.synthetic_state(cfg0, 1234)
    nb = synth.synthetic_batch(cfg0, B=4, S=19, T=150, seed=5, in_lens=[19, 12, 19, 3], tgt_lens=[150, 70, 128, 2])
    m, cfg, _, hp = build(over, compute_dtype="bf16", state_edit=lambda s: s.update(st))
    m.train()
    eng = m.engine()
    b = dev_batch(nb)
    in32, tgt32 = b["input_lengths"].int(), b["target_lengths"].int()
    mem, c_enc = eng.encoder_forward(b["inputs"], in32, b.get("input_spk_ids"), b.get("input_language_vecs"), True, 11, False)
    outs = []
    for host in (None, [int(x) for x in nb["target_lengths"]]):
        mels, stop, c = eng.decoder_forward(mem, in32, b["mel_targets"], tgt32, True, 13, True, padded_unobserved=True, target_lengths_host=host)
        g = torch.Generator(device=mels.device).manual_seed(1)
        dm = torch.randn(mels.shape, generator=g, device=mels.device)
        ds = torch.randn(stop.shape, generator=g, device=mels.device)
        valid = (torch.arange(mels.shape[1], device=mels.device)[None, :] < b["target_lengths"][:, None])
        dm = dm * valid[..., None]; ds = ds * valid
        eng.begin_backward()
        dmem = eng.decoder_backward(c, dm, ds, mem.shape)
        torch.cuda.synchronize()
        outs.append((mels.clone(), stop.clone(), dmem.clone()))
        c.free()
    (m0, s0, d0), (m1, s1, d1) = outs
    assert torch.equal(m0, m1) and torch.equal(s0, s1), "forward outputs are bit-identical"
    valid = (torch.arange(m0.shape[1], device=m0.device)[None, :] < b["target_lengths"][:, None])
    assert float(m1[~valid].abs().max()) == 0.0 and float(s1[~valid].abs().max()) == 0.0
    assert torch.equal(d0, d1), "d(memory) is bit-identical (its sums run over query tiles of one utterance: the same order in both layouts)"


def test_ragged_dropout_on_matches_oracle_under_device_masks():
    """fp32 mode, reference dropout rates, ragged rows: outputs, losses and every gradient against the oracle with the engine's masks -- whose row
    sites index RAGGED rows (DeviceMasks(ragged_lengths=...)); the same comparison with the padded row index fails."""
    over = with_dropout(TINY96)
    cfg0 = make_config(over)
    st = synth.synthetic_state(cfg0, 1234)
    lens = [23, 15, 9]
    nb = synth.synthetic_batch(cfg0, B=3, S=11, T=23, seed=7, in_lens=[11, 7, 4], tgt_lens=lens)
    vals, g0, _, seeds, cfg, m = _trainer_steps(over, st, nb, "fp32", ragged=True, steps=1)
    P = O.to_torch_state(st, requires_grad=True)
    ob = O.to_torch_batch(nb)
    res = {}
    for tag, rl in (("ragged", {"decoder": lens}), ("padded", None)):
        src = R.DeviceMasks(seeds, site_info, ragged_lengths=rl)
        with O.device_masks(src):
            out = O.tacotron_forward(P, cfg, ob, train=True)
        losses = O.compute_loss(P, cfg, ob["mel_targets"], ob["target_lengths"], out)
        names = [n for n in P if O.is_parameter(n)]
        gl = torch.autograd.grad(losses["loss"], [P[n] for n in names], allow_unused=True)
        ref_g = {n: (g.detach().float() if g is not None else torch.zeros_like(P[n])) for n, g in zip(names, gl)}
        e, n = worst_direction({k: v for k, v in g0.items() if v.numel() > 1}, {k: v for k, v in ref_g.items() if v.numel() > 1})
        res[tag] = (abs(float(vals[0][0]) - float(losses["loss"])) / abs(float(losses["loss"])), e, n)
        print("ragged step vs oracle with the %s row index: loss rel %.2e, worst gradient %.2e (%s)" % (tag, res[tag][0], e, n))
    assert res["ragged"][0] < 2e-4 and res["ragged"][1] < 1e-3, res["ragged"]
    assert res["padded"][1] > 1e-2, "the padded row index is a different (equally valid) mask: the comparison must notice"


def test_ragged_rows_with_guided_attention_and_frozen_encoder():
    """The few-shot fine-tune configuration (guided-attention loss on the encoder-decoder alignments, frozen encoder: both absent upstream) on ragged
    rows: the guided loss sums per-frame row sums over [B H, T] -- frames that do not exist in the ragged layout must contribute nothing -- and
    every loss term and gradient agrees with the padded layout at dropout 0."""
    over = TINY96 + ",guided_attention_weight=1.0,guided_attention_sigma=0.2,freeze_encoder=true"
    cfg0 = make_config(TINY96)
    st = synth.synthetic_state(cfg0, 1234)
    nb = synth.synthetic_batch(cfg0, B=4, S=19, T=150, seed=5, in_lens=[19, 12, 19, 3], tgt_lens=[150, 70, 128, 2])
    res = []
    for ragged in (False, True):
        from b2s_hip.trainer import HipTrainer
        m, cfg, _, hp = build(over, compute_dtype="fp32", state_edit=lambda s: s.update(st))
        m.train()
        tr = HipTrainer(m, hp, dist=False)
        b = dev_batch(nb)
        if ragged:
            b["target_lengths_host"] = [int(x) for x in nb["target_lengths"]]
        grads = []
        tr.grad_probe = lambda flat, wire: grads.append(flat.detach().clone())
        v = tr.train_step(b).detach().clone()
        torch.cuda.synchronize()
        res.append((v, float(tr.last_ga_loss), {n: grads[0][o:o + c].cpu() for n, (o, c) in tr.eng.param_offsets.items() if not n.startswith("encoder.")}))
    (va, gaa, ga), (vb, gab, gb) = res
    assert gaa > 0 and abs(gaa - gab) <= 1e-6 * gaa, (gaa, gab)
    for i in range(va.numel()):
        assert abs(float(va[i]) - float(vb[i])) <= 1e-6 * abs(float(va[i])) + 1e-9, (i, va, vb)
    e, n = worst_direction(gb, ga)
    print("guided attention + frozen encoder: ragged vs padded, guided loss %.6f / %.6f, worst gradient difference %.2e (%s)" % (gaa, gab, e, n))
    assert e < 2e-5, (e, n)
