"""Checkpoint interchange between the fused HipTrainer and a train.py-style torch.optim.Adam + LambdaLR loop
(reference format, utils/checkpoint.py:19-58): resume in either direction continues the same trajectory."""
import os
from functools import partial

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import synth, TINY
from test_gpu_model import build, dev_batch


def _torch_loop(m, hp):
    from transformer.tacotron import learning_rate_schedule
    optim = torch.optim.Adam(m.parameters(), lr=hp.max_lr, eps=hp.adam_eps)
    sched = torch.optim.lr_scheduler.LambdaLR(optim, lr_lambda=partial(learning_rate_schedule, hp=hp))
    return optim, sched


def _torch_step(m, hp, b, optim, sched):
    from transformer.tacotron import compute_loss
    o = m(**b)
    losses = compute_loss(m, b["mel_targets"], b["target_lengths"], o, hp)
    optim.zero_grad()
    losses["loss"].backward()
    optim.step()
    sched.step()
    return float(losses["loss"])


def _compare(ma, mb, tol):
    worst = 0.0
    for (n, a), (_, b) in zip(ma.state_dict().items(), mb.state_dict().items()):
        a, b = a.detach().cpu().double(), b.detach().cpu().double()
        if n.endswith("num_batches_tracked"):
            assert int(a) == int(b)
            continue
        assert abs(float(a.norm()) - float(b.norm())) <= 2e-4 * float(b.norm()) + 1e-5, n
        worst = max(worst, float((a - b).abs().max()))
    assert worst < tol, worst


def test_fused_trainer_checkpoint_resumes_under_torch_loop(tmp_path):
    from b2s_hip.trainer import HipTrainer
    from utils import checkpoint
    ma, cfg, st, hp = build(TINY)
    b = dev_batch(synth.synthetic_batch(cfg, B=3, S=11, T=23, seed=7, in_lens=[11, 7, 4], tgt_lens=[23, 15, 9]))
    ma.train()
    tr = HipTrainer(ma, hp)
    for _ in range(2):
        tr.train_step(b)
    path = checkpoint.save_model(str(tmp_path), ma, tr, tr.sched, tr.global_step)
    assert os.path.basename(path) == "model.ckpt-2"
    saved = torch.load(path, map_location="cpu")
    ref_opt = torch.optim.Adam(torch.nn.Linear(2, 2).parameters()).state_dict()
    assert set(saved) == {"model", "optim", "sched", "step"}
    assert set(ref_opt["param_groups"][0]) <= set(saved["optim"]["param_groups"][0]) | {"params"}
    mb, _, _, _ = build(TINY, seed=99)                 # different weights: everything must come from the file
    mb.train()
    optim, sched = _torch_loop(mb, hp)
    assert checkpoint.load_model(path, mb, optim, sched, "cuda") == 2
    assert sched.last_epoch == 2
    _torch_step(mb, hp, b, optim, sched)
    tr.train_step(b)
    torch.cuda.synchronize()
    _compare(ma, mb, 2.5e-3)


def test_torch_loop_checkpoint_resumes_under_fused_trainer(tmp_path):
    from b2s_hip.trainer import HipTrainer
    from utils import checkpoint
    ma, cfg, st, hp = build(TINY)
    b = dev_batch(synth.synthetic_batch(cfg, B=3, S=11, T=23, seed=7, in_lens=[11, 7, 4], tgt_lens=[23, 15, 9]))
    ma.train()
    optim, sched = _torch_loop(ma, hp)
    for _ in range(2):
        _torch_step(ma, hp, b, optim, sched)
    path = checkpoint.save_model(str(tmp_path), torch.nn.DataParallel(ma), optim, sched, 2)     # wrapper is unwrapped
    mb, _, _, _ = build(TINY, seed=99)
    mb.train()
    tr = HipTrainer(mb, hp)
    assert checkpoint.load_model(path, mb, tr, tr.sched, "cuda") == 2
    assert tr.global_step == 2
    tr.train_step(b)
    _torch_step(ma, hp, b, optim, sched)
    torch.cuda.synchronize()
    _compare(mb, ma, 2.5e-3)
    # and the trainer's own state survives a save / load cycle bit-exactly
    path2 = checkpoint.save_model(str(tmp_path), mb, tr, tr.sched, tr.global_step)
    mc, _, _, _ = build(TINY, seed=5)
    trc = HipTrainer(mc, hp)
    assert checkpoint.load_model(path2, mc, trc, trc.sched, "cuda") == 3
    assert torch.equal(trc.exp_avg, tr.exp_avg) and torch.equal(trc.exp_avg_sq, tr.exp_avg_sq)
    va, vb = tr.train_step(b), trc.train_step(b)
    torch.cuda.synchronize()
    assert torch.allclose(va, vb, rtol=1e-5, atol=1e-6)      # conv weight gradients use fp32 atomics: last-bit order effects
