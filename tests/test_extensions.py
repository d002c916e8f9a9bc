"""Extensions named by the north-star but absent upstream: guided-attention loss on the
encoder-decoder alignments and the frozen-encoder few-shot fine-tuning mode.  The reference has no counterpart
(PARITY UNPINNED): the oracle's restatement of the published formula is checked by properties here (CPU), and the
HIP path is checked against that oracle in test_gpu_extensions.py.
"""
import math

import numpy as np
import torch

from oracle import b2s_oracle as O
from oracle import synth, make_config, TINY


def test_guided_attention_closed_forms():
    B, H, S, T = 2, 3, 8, 20
    in_len, tgt_len = torch.tensor([8, 5]), torch.tensor([20, 13])
    sigma = 0.2
    # uniform attention over the valid keys: loss = mean over valid (b, n, t) of W / N_b ... computed directly
    A = torch.zeros(B, H, S, T)
    for b in range(B):
        A[b, :, :in_len[b], :] = 1.0 / float(in_len[b])
    got = float(O.guided_attention_loss([A, A], in_len, tgt_len, sigma))
    num, den = 0.0, 0.0
    for b in range(B):
        N, Tb = int(in_len[b]), int(tgt_len[b])
        for n in range(N):
            for t in range(Tb):
                num += (1.0 - math.exp(-((n / N - t / Tb) ** 2) / (2 * sigma ** 2))) / N
        den += N * Tb
    assert abs(got - num / den) < 1e-6
    # a perfectly diagonal alignment costs (almost) nothing; an anti-diagonal one is heavily penalised
    diag, anti = torch.zeros(B, H, S, T), torch.zeros(B, H, S, T)
    for b in range(B):
        N, Tb = int(in_len[b]), int(tgt_len[b])
        for t in range(Tb):
            n = min(N - 1, int(round(t / Tb * N)))
            diag[b, :, n, t] = 1.0
            anti[b, :, N - 1 - n, t] = 1.0
    ld = float(O.guided_attention_loss([diag], in_len, tgt_len, sigma))
    la = float(O.guided_attention_loss([anti], in_len, tgt_len, sigma))
    assert ld < 0.01 and la > 10 * ld
    # padded frames / bytes never contribute
    noisy = diag.clone()
    noisy[0, :, :, 20:] = 7.0
    noisy[1, :, 5:, :] = 7.0
    noisy[1, :, :, 13:] = 7.0
    assert abs(float(O.guided_attention_loss([noisy], in_len, tgt_len, sigma)) - ld) < 1e-7


def test_guided_and_frozen_training_step_oracle():
    cfg = make_config(TINY + ",guided_attention_weight=2.0,freeze_encoder=true")
    st = synth.synthetic_state(cfg, 5)
    P = O.to_torch_state(st, requires_grad=True)
    before = {n: p.detach().clone() for n, p in P.items()}
    b = O.to_torch_batch(synth.synthetic_batch(cfg, B=3, S=11, T=23, seed=7, in_lens=[11, 7, 4], tgt_lens=[23, 15, 9]))
    out, losses, grads = O.train_step(P, cfg, b, {}, 0, train=True)
    assert "ga_loss" in losses and float(losses["ga_loss"]) > 0
    base = losses["bef_loss"] + losses["aft_loss"] + losses["l2"] + losses["stop_loss"]
    assert abs(float(losses["loss"]) - float(base + losses["ga_loss"])) < 1e-6
    moved = [n for n in P if O.is_parameter(n) and not torch.equal(P[n].detach(), before[n])]
    assert moved and not any(n.startswith("encoder.") for n in moved)
    assert all(not n.startswith("encoder.") for n in grads)
    # the guided term reaches the encoder-decoder query / key projections
    cfg0 = make_config(TINY)
    P0 = O.to_torch_state(st, requires_grad=True)
    _, _, g0 = O.train_step(P0, cfg0, b, {}, 0, train=True)
    n = "decoder.decoder.encdec_attentions.0.q_transform.weight"
    assert float((grads[n] - g0[n]).abs().max()) > 1e-6
