"""Tacotron.forward runs encoder + decoder as ONE autograd node (b2s_hip/engine.py: EncDecFn) so that the encoder runs beside the decoder as it does under
HipTrainer; both streams are joined inside the node.  Same kernels, same seeds: outputs are bit-identical to the two-node path (B2S_DROPIN_OVERLAP=0:
EncoderFn + DecoderFn, tacotron.py:126-129 literally), gradients agree to the run-to-run level of either path (fp32 atomics in the BatchNorm statistics,
LayerNorm / bias sums), alignments and the frozen-encoder / eval flows work."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import synth, make_config, TINY96
from test_gpu_model import build, dev_batch
from test_gpu_dropout_parity import with_dropout, worst_direction


def _run(monkeypatch, overlap, over, compute_dtype, freeze=False):
    from transformer.tacotron import compute_loss
    monkeypatch.setenv("B2S_DROPIN_OVERLAP", "1" if overlap else "0")
    torch.manual_seed(1234)
    m, cfg, st, hp = build(over + (",freeze_encoder=true" if freeze else ""), compute_dtype=compute_dtype)
    m.train()
    nb = synth.synthetic_batch(cfg, B=4, S=19, T=150, seed=5, in_lens=[19, 12, 19, 3], tgt_lens=[150, 70, 128, 2])
    b = dev_batch(nb)
    out = m(**b)
    losses = compute_loss(m, b["mel_targets"], b["target_lengths"], out, hp)
    losses["loss"].backward()
    torch.cuda.synchronize()
    grads = {n: (p.grad.detach().float().cpu().clone() if p.grad is not None else None) for n, p in m.named_parameters()}
    al = out["alignments"]["encdec"][0].detach().cpu()
    return {k: out[k].detach().cpu().clone() for k in ("mel_bef", "mel_aft", "stop_logits")}, float(losses["loss"].detach()), grads, al, dict(m.engine().seeds_used)


@pytest.mark.parametrize("compute_dtype", ["fp32", "bf16"])
def test_one_node_equals_two_nodes(monkeypatch, compute_dtype):
    over = with_dropout(TINY96)
    o1, l1, g1, a1, s1 = _run(monkeypatch, True, over, compute_dtype)
    o0, l0, g0, a0, s0 = _run(monkeypatch, False, over, compute_dtype)
    assert s1 == s0, "same dropout seeds in both paths"
    for k in ("mel_bef", "stop_logits"):
        assert torch.equal(o1[k], o0[k]), k
    # (the postnet's BatchNorm statistics are fp32 atomics: not bit-reproducible run to run in either path)
    assert float((o1["mel_aft"] - o0["mel_aft"]).abs().max()) <= (1e-4 if compute_dtype == "fp32" else 5e-2)
    assert torch.equal(a1, a0)
    assert abs(l1 - l0) <= (1e-6 if compute_dtype == "fp32" else 1e-4) * abs(l0)
    e, n = worst_direction({k: v for k, v in g1.items() if v is not None and v.numel() > 1}, {k: v for k, v in g0.items() if v is not None and v.numel() > 1})
    print("worst per-tensor gradient difference one node vs two (%s): %.2e (%s)" % (compute_dtype, e, n))
    assert e < (2e-5 if compute_dtype == "fp32" else 1e-2), (e, n)


def test_one_node_frozen_encoder_and_eval(monkeypatch):
    o1, l1, g1, a1, _ = _run(monkeypatch, True, TINY96, "fp32", freeze=True)
    o0, l0, g0, a0, _ = _run(monkeypatch, False, TINY96, "fp32", freeze=True)
    for k in ("mel_bef", "stop_logits"):
        assert torch.equal(o1[k], o0[k]), k
    assert all(v is None for k, v in g1.items() if k.startswith("encoder.")), "a frozen encoder gets no gradients"
    assert all(v is not None for k, v in g1.items() if not k.startswith("encoder."))
    e, n = worst_direction({k: v for k, v in g1.items() if v is not None and v.numel() > 1}, {k: v for k, v in g0.items() if v is not None and v.numel() > 1})
    assert e < 2e-5, (e, n)
    # eval / no_grad through the one-node path
    monkeypatch.setenv("B2S_DROPIN_OVERLAP", "1")
    m, cfg, st, hp = build(TINY96)
    m.eval()
    nb = synth.synthetic_batch(cfg, B=3, S=11, T=23, seed=7, in_lens=[11, 7, 4], tgt_lens=[23, 15, 9])
    with torch.no_grad():
        a = m(**dev_batch(nb))
    monkeypatch.setenv("B2S_DROPIN_OVERLAP", "0")
    with torch.no_grad():
        b = m(**dev_batch(nb))
    for k in ("mel_bef", "mel_aft", "stop_logits"):
        assert torch.equal(a[k], b[k]), k
