"""The autograd path hands autograd VIEWS of the engine's flat gradient buffer (b2s_hip/engine.py: grad_out).  In the reference's loop (train.py:171-174:
forward, compute_loss, optim.zero_grad() -- set_to_none -- backward, optim.step()) autograd keeps such a view as .grad instead of cloning it (162 copy
kernels / 334 MB per step saved); every flow in which the view would outlive the buffer's next clearing gets a copy first (_reclaim_lent):

  * the reference's loop: .grad lives inside the flat buffer, values = the gradients of the cloning path, torch.optim.Adam steps as before;
  * gradient accumulation (two backward passes, no zero_grad in between): .grad = g1 + g2, as torch computes it with cloned gradients;
  * zero_grad(set_to_none=False) loops: .grad keeps its own storage from the second pass on.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import synth, make_config, TINY96
from test_gpu_model import build, dev_batch


def _setup(compute_dtype="fp32"):
    from transformer.tacotron import compute_loss
    m, cfg, st, hp = build(TINY96, compute_dtype=compute_dtype)
    m.train()
    nbs = [synth.synthetic_batch(cfg, B=3, S=11, T=23, seed=s, in_lens=[11, 7, 4], tgt_lens=[23, 15, 9]) for s in (7, 8)]
    def run(nb):
        b = dev_batch(nb)
        out = m(**b)
        return compute_loss(m, b["mel_targets"], b["target_lengths"], out, hp)["loss"]
    return m, hp, nbs, run


def _in_flat(eng, t):
    lo = eng._gflat.data_ptr()
    return lo <= t.data_ptr() < lo + eng._gflat.numel() * 4


def test_reference_loop_keeps_views_and_steps_like_cloned_gradients():
    m, hp, nbs, run = _setup()
    eng = m.engine()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    # pass 1: gradients through the lending path
    loss = run(nbs[0]); opt.zero_grad(); loss.backward()
    torch.cuda.synchronize()
    lent = {n: p.grad for n, p in m.named_parameters()}
    assert all(g is not None for g in lent.values())
    assert all(_in_flat(eng, g) for g in lent.values()), "every .grad is a view of the flat buffer (no clone)"
    ref = {n: g.detach().clone() for n, g in lent.items()}
    # the same pass with .grad pre-set (autograd then ADDS the cached view to it): identical values, own storage
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    loss = run(nbs[0]); loss.backward()
    torch.cuda.synchronize()
    for n, p in m.named_parameters():
        assert not _in_flat(eng, p.grad)
        assert torch.equal(p.grad, ref[n]) or float((p.grad - ref[n]).abs().max()) <= 1e-6 * float(ref[n].abs().max() + 1e-30), n
    # an optimizer step on lent gradients, then the next iteration of the loop
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    loss = run(nbs[0]); opt.zero_grad(); loss.backward(); opt.step()
    loss2 = run(nbs[1]); opt.zero_grad(); loss2.backward(); opt.step()
    torch.cuda.synchronize()
    moved = sum(int(not torch.equal(before[n], p.detach())) for n, p in m.named_parameters())
    assert moved >= 0.9 * len(before) and np.isfinite(float(loss2))


def test_accumulation_without_zero_grad_adds_up():
    m, hp, nbs, run = _setup()
    eng = m.engine()
    # the two gradients one by one (each zeroed with set_to_none, cloned out)
    gs = []
    for nb in nbs:
        for p in m.parameters():
            p.grad = None
        run(nb).backward()
        torch.cuda.synchronize()
        gs.append({n: p.grad.detach().clone() for n, p in m.named_parameters()})
    # accumulated: the first pass lends views, the second must find them replaced by copies before the buffer is cleared
    for p in m.parameters():
        p.grad = None
    run(nbs[0]).backward()
    assert all(_in_flat(eng, p.grad) for p in m.parameters())
    run(nbs[1]).backward()
    torch.cuda.synchronize()
    for n, p in m.named_parameters():
        assert not _in_flat(eng, p.grad), n
        want = gs[0][n] + gs[1][n]
        assert float((p.grad - want).abs().max()) <= 1e-5 * float(want.abs().max() + 1e-30) + 1e-12, n


def test_zero_in_place_loop_keeps_own_storage():
    m, hp, nbs, run = _setup()
    eng = m.engine()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    run(nbs[0]).backward()                                  # lent
    opt.zero_grad(set_to_none=False)                        # zeroes the views in place
    run(nbs[1]).backward()                                  # reclaim: copies (of zeros), then + g2
    torch.cuda.synchronize()
    for p in m.parameters():
        p2 = p.grad
        assert not _in_flat(eng, p2)
    for p in m.parameters():
        p.grad = None
    run(nbs[1]).backward()
    torch.cuda.synchronize()
    # (the pass above lent again: same values as the in-place loop produced)
    assert all(_in_flat(eng, p.grad) for p in m.parameters())
