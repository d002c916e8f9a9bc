"""Fused-trainer state hazards (round-1 advisor findings): parameter re-binding under an attached trainer, fp32 conv GEMM
images after the fused optimizer step, exceptions raised inside the backward stage hook, and the one-forward-per-backward
contract of the autograd path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import b2s_oracle as O                     # checker only
from oracle import synth, TINY, TINY96
from gpu_util import DEV
from test_gpu_model import build, dev_batch


def _batch(cfg):
    nb = synth.synthetic_batch(cfg, B=3, S=11, T=23, seed=7, in_lens=[11, 7, 4], tgt_lens=[23, 15, 9])
    return nb, dev_batch(nb)


@pytest.mark.parametrize("compute_dtype", ["fp32", "bf16"])
def test_rebind_under_trainer_updates_live_parameters(compute_dtype):
    """Parameters replaced after HipTrainer was built (here: every .data swapped for a fresh allocation, what
    load_state_dict(assign=True) / .to() do) must keep being updated by the fused Adam step: the run equals an undisturbed
    one, and the abandoned storage is never written again."""
    from b2s_hip.trainer import HipTrainer
    ma, cfg, _, hp = build(TINY96, compute_dtype=compute_dtype)
    nb, b = _batch(cfg)
    ma.train()
    ta = HipTrainer(ma, hp)
    ta.eng._seed = 99; ta.eng._calls = 0
    for _ in range(3):
        ta.train_step(b)
    ref = {k: v.detach().clone() for k, v in ma.state_dict().items()}
    mb, _, _, hp = build(TINY96, compute_dtype=compute_dtype)
    mb.train()
    tb = HipTrainer(mb, hp)
    tb.eng._seed = 99; tb.eng._calls = 0
    tb.train_step(b)
    torch.cuda.synchronize()
    old = {n: p.data for n, p in mb.named_parameters()}
    snap = {n: t.clone() for n, t in old.items()}
    with torch.no_grad():
        for n, p in mb.named_parameters():
            p.data = p.data.clone()                     # new storage, same values: data_ptr changes -> the engine re-binds
    for _ in range(2):
        tb.train_step(b)
    torch.cuda.synchronize()
    for n, t in old.items():
        assert torch.equal(t, snap[n]), "stale storage of %s was written after the re-bind" % n
    tol = 1e-4 if compute_dtype == "fp32" else 2e-2     # (split-K atomics reorder sums: runs differ by ~2e-5 after 3 Adam steps)
    for k, v in mb.state_dict().items():
        d = float((v.double() - ref[k].double()).abs().max())
        assert d <= tol * (1.0 + float(ref[k].double().abs().max())), (k, d)
    moved = max(float((p.data - snap[n]).abs().max()) for n, p in mb.named_parameters())
    assert moved > 1e-4, "live parameters did not move after the re-bind"


def test_fp32_step_then_eval_forward_uses_updated_conv_weights():
    """fp32 (parity) mode: after one fused step an eval-mode forward (postnet conv GEMM images included) must see the weights
    the step produced -- compared with the oracle evaluated on the oracle's own updated parameters."""
    from b2s_hip.trainer import HipTrainer
    m, cfg, st, hp = build(TINY96)
    nb, b = _batch(cfg)
    m.train()
    tr = HipTrainer(m, hp)
    P = O.to_torch_state(st, requires_grad=True)
    ob = O.to_torch_batch(nb)
    opt = {}
    for step in range(2):
        tr.train_step(b)
        O.train_step(P, cfg, ob, opt, step, train=True)
    m.eval()
    with torch.no_grad():
        o = m(**b)
        ref = O.tacotron_forward({k: v.detach() for k, v in P.items()}, cfg, ob, train=False)
    for k in ("mel_bef", "mel_aft"):
        d = float((o[k].cpu() - ref[k]).abs().max())
        assert d < 2e-4, (k, d)
    # the postnet on its own (the segment eval_batch calls) agrees too
    with torch.no_grad():
        res = m.postnet(b["mel_targets"], b["target_lengths"]).cpu()
        rres = O.postnet_forward({k: v.detach() for k, v in P.items()}, cfg, ob["mel_targets"], ob["target_lengths"], train=False)
    rres = rres[0] if isinstance(rres, tuple) else rres
    assert float((res - rres).abs().max()) < 2e-4


@pytest.mark.parametrize("with_split", [False, True])
def test_stage_hook_exception_is_raised_before_the_optimizer_step(with_split):
    """A stage hook that raises (a failed collective) abandons the step: no optimizer launch at all -- also with the data-parallel tail
    schedule armed (GradBucketer.split set), whose decoder / postnet update would otherwise be issued before the error is looked at."""
    from b2s_hip.trainer import HipTrainer
    import ctypes as C
    from b2s_hip import lib as L
    m, cfg, _, hp = build(TINY)
    _, b = _batch(cfg)
    m.train()
    tr = HipTrainer(m, hp)

    class Boom(object):
        launched = []
        def begin_step(self): pass
        def stage_done(self, stage): raise ValueError("boom at stage %d" % stage)
        def finish(self, expect_all=True): raise AssertionError("finish must not run after a failed hook")
        def abort(self): self.aborted = True
        def wait_prefix(self, limit): raise AssertionError("the tail update must not be attempted after a failed hook")
        def covers_all(self, expect_all=True): raise AssertionError("the tail update must not be attempted after a failed hook")
    if with_split:
        Boom.split = 1
    tr.bucketer = Boom()
    L.check(tr.lib.b2s_model_set_stage_hook(tr.eng.handle, C.cast(tr._hook, L.P), None, None))
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    with pytest.raises(RuntimeError, match="optimizer step was NOT applied") as ei:
        tr.train_step(b)
    assert isinstance(ei.value.__cause__, ValueError) and tr.bucketer.aborted
    torch.cuda.synchronize()
    assert tr.global_step == 0
    for n, p in m.named_parameters():
        assert torch.equal(p.detach(), before[n]), n


def test_autograd_path_refuses_two_forwards_in_one_backward():
    """(lossA + lossB).backward() over two forward passes would return 2 x (A + B) through the shared flat gradient buffer:
    the engine raises instead; one forward per backward keeps working and matches the oracle gradient."""
    from transformer.tacotron import compute_loss
    from b2s_hip import lib as L
    m, cfg, st, hp = build(TINY)
    nb, b = _batch(cfg)
    m.train()
    la = compute_loss(m, b["mel_targets"], b["target_lengths"], m(**b), hp)["loss"]
    lb = compute_loss(m, b["mel_targets"], b["target_lengths"], m(**b), hp)["loss"]
    with pytest.raises((L.B2SError, RuntimeError), match="one forward per backward"):
        (la + lb).backward()
    for p in m.parameters():
        p.grad = None
    lc = compute_loss(m, b["mel_targets"], b["target_lengths"], m(**b), hp)["loss"]
    lc.backward()
    P = O.to_torch_state(st, requires_grad=True)
    ob = O.to_torch_batch(nb)
    rl = O.compute_loss(P, cfg, ob["mel_targets"], ob["target_lengths"], O.tacotron_forward(P, cfg, ob, train=True))["loss"]
    rl.backward()
    for n, p in m.named_parameters():
        ref = P[n].grad
        assert abs(float(p.grad.double().norm()) - float(ref.double().norm())) <= 2e-3 * float(ref.double().norm()) + 1e-6, n


@pytest.mark.parametrize("extra", ["", ",n_decoder_layer=3"])
def test_tail_optimizer_step_matches_single_launch(extra):
    """Default schedule (HipTrainer(tail_adam=True)): the decoder / postnet update runs on the trainer's stream behind b2s_model_mark_grads_ready,
    beside the encoder backward on its own stream, the encoder group after it.  Same arithmetic as the single launch after the whole
    backward pass: fp32 mode (no atomics in the compared path beyond the column sums), identical losses and parameters after 3 steps."""
    from b2s_hip.trainer import HipTrainer
    res = []
    for tail in (False, True):
        m, cfg, _, hp = build(TINY96 + extra, compute_dtype="bf16")
        w0 = m.decoder.prenet.dense0.weight.detach().clone()
        _, b = _batch(cfg)
        m.train()
        tr = HipTrainer(m, hp, tail_adam=tail)
        assert tr.tail_adam == tail and tr.overlap_encoder
        for _ in range(3):
            v = tr.train_step(b)
        torch.cuda.synchronize()
        assert tr.last_step_tail_update == tail
        res.append(({k: t.detach().clone() for k, t in m.state_dict().items()}, v.cpu().numpy()))
        # (an odd number of decoder stages leaves the prenet stage's weight-gradient group to the drain: the prenet must still train)
        assert float((m.decoder.prenet.dense0.weight.detach() - w0).abs().max()) > 1e-4
    for k, t in res[0][0].items():
        d = float((t.double() - res[1][0][k].double()).abs().max())
        assert d <= 2e-2 * (1.0 + float(t.double().abs().max())), (k, d)           # bf16 + atomics: runs are not bit-reproducible
    assert abs(res[0][1][0] - res[1][1][0]) < 2e-2 * abs(res[0][1][0])


def test_tail_update_protocol_is_checked():
    """b2s_adam_step_groups(behind_mark = 1) is only valid behind b2s_model_mark_grads_ready, never for the encoder group, and a group is
    stepped once per step: each misuse is an error code with a message, not a silent partial update.  b2s_add3(c) == two b2s_add3(NULL) calls."""
    from b2s_hip.trainer import HipTrainer
    from b2s_hip import lib as L
    m, cfg, _, hp = build(TINY96, compute_dtype="bf16")
    _, b = _batch(cfg)
    m.train()
    tr = HipTrainer(m, hp)
    tr.train_step(b)
    torch.cuda.synchronize()
    lib, h = tr.lib, tr.eng.handle
    adam = (1e-3, 2, 0.9, 0.999, 1e-6, 0.0, 1.0)
    with pytest.raises(L.B2SError, match="mark_grads_ready"):
        L.check(lib.b2s_adam_step_groups(h, *adam, 2 | 4, 1, L.stream()))           # no mark
    L.check(lib.b2s_model_mark_grads_ready(h))
    with pytest.raises(L.B2SError, match="complete at the mark"):
        L.check(lib.b2s_adam_step_groups(h, *adam, 1 | 2, 1, L.stream()))           # encoder group behind a decoder mark
    with pytest.raises(L.B2SError, match="placement"):
        L.check(lib.b2s_adam_step_groups(h, *adam, 2, 3, L.stream()))
    before = {k: v.detach().clone() for k, v in m.state_dict().items() if v.is_floating_point()}
    torch.cuda.synchronize()
    for k, v in m.state_dict().items():
        if k in before:
            assert torch.equal(v, before[k]), k                                         # nothing was applied by the refused calls
    lib.b2s_model_backward_abort(h, L.stream())                                         # (drops the mark)
    with pytest.raises(L.B2SError, match="mark_grads_ready"):
        L.check(lib.b2s_adam_step_groups(h, *adam, 2 | 4, 1, L.stream()))
    x, y, z = (torch.randn(1000, 80, device=DEV) for _ in range(3))
    assert torch.equal(tr.eng.add3(x, y, z), tr.eng.add(tr.eng.add(x, y), z))          # (eng.add = b2s_add3 with c = NULL)
    tr.train_step(b)                                                                    # the trainer is still usable
    torch.cuda.synchronize()
    assert tr.global_step == 2


def test_failure_after_the_partial_update_consumes_the_step():
    """Data-parallel tail schedule: the decoder / postnet update is issued before the remaining collectives are waited for.  If one of THOSE
    fails (finish() raises), the step is half applied -- the trainer must say so, advance its step counter (a retry under the same number
    would be refused by b2s_adam_step_groups: the trainer used to be wedged) and keep working.  What can be checked without waiting (a
    hook error, a stage that did not report) is checked BEFORE the partial update: such a step is refused as a whole, nothing applied."""
    from b2s_hip.trainer import HipTrainer
    from test_gpu_dp_race import FakeDist
    m, cfg, _, hp = build(TINY96, compute_dtype="bf16")
    _, b = _batch(cfg)
    m.train()
    tr = HipTrainer(m, hp, dist=FakeDist(spin_cycles=0), grad_payload="bf16")
    tr.close()                                                                           # (world 2 selected the data-parallel GEMM tile policy process-wide)
    assert tr.bucketer is not None and tr.bucketer.split is not None
    tr.train_step(b)
    torch.cuda.synchronize()
    assert tr.global_step == 1 and tr.last_step_tail_update
    enc0 = m.encoder.encoder.ffn_layers[0].input_layer.weight.detach().clone()
    dec0 = m.decoder.decoder.ffn_layers[0].input_layer.weight.detach().clone()
    real_finish = tr.bucketer.finish

    def failing_finish(expect_all=True):
        real_finish(expect_all)
        raise RuntimeError("collective failed")
    tr.bucketer.finish = failing_finish
    with pytest.raises(RuntimeError, match="PARTIAL update") as ei:
        tr.train_step(b)
    assert "collective failed" in str(ei.value.__cause__)
    torch.cuda.synchronize()
    assert tr.global_step == 2                                                           # consumed
    assert not torch.equal(m.decoder.decoder.ffn_layers[0].input_layer.weight.detach(), dec0)      # decoder group was updated ...
    assert torch.equal(m.encoder.encoder.ffn_layers[0].input_layer.weight.detach(), enc0)          # ... the encoder group was not
    tr.bucketer.finish = real_finish
    tr.train_step(b)                                                                     # not wedged: the next step is step 3 on every group
    torch.cuda.synchronize()
    assert tr.global_step == 3 and tr.last_step_tail_update
    assert not torch.equal(m.encoder.encoder.ffn_layers[0].input_layer.weight.detach(), enc0)
    # a stage that never reports: noticed before any update is issued -> the single-update path refuses the whole step
    snap = {k: v.detach().clone() for k, v in m.state_dict().items() if v.is_floating_point()}
    real_stage_done = tr.bucketer.stage_done
    last = tr.eng.n_stages() - 1
    tr.bucketer.stage_done = lambda st: None if st == last - 1 else real_stage_done(st)
    with pytest.raises((RuntimeError, AssertionError)) as ei2:
        tr.train_step(b)
    assert "PARTIAL" not in str(ei2.value)
    torch.cuda.synchronize()
    assert tr.global_step == 3
    for k, v in m.state_dict().items():
        if k in snap and "running" not in k:
            assert torch.equal(v, snap[k]), k
    tr.bucketer.stage_done = real_stage_done
    tr.train_step(b)
    torch.cuda.synchronize()
    assert tr.global_step == 4


@pytest.mark.parametrize("where", ["decoder_backward", "encoder_backward"])
def test_failed_backward_is_abandoned_cleanly(where):
    """An exception between the backward entry points (which hand queued weight-gradient work to each other: deferred joins) must not
    leave that work queued against freed contexts: the trainer calls b2s_model_backward_abort, the step is not applied, and the NEXT
    step equals the first step of an undisturbed trainer."""
    from b2s_hip.trainer import HipTrainer
    ma, cfg, _, hp = build(TINY96, compute_dtype="fp32")
    nb, b = _batch(cfg)
    ma.train()
    ta = HipTrainer(ma, hp)
    ta.eng._seed = 5; ta.eng._calls = 0
    before = {k: v.detach().clone() for k, v in ma.state_dict().items() if v.is_floating_point()}
    orig = getattr(ta.eng, where)
    def boom(*a, **k):
        raise RuntimeError("injected failure in " + where)
    setattr(ta.eng, where, boom)
    with pytest.raises(RuntimeError, match="injected"):
        ta.train_step(b)
    setattr(ta.eng, where, orig)
    torch.cuda.synchronize()
    assert ta.global_step == 0
    for k, v in ma.state_dict().items():
        if k in before and "running_" not in k and "num_batches" not in k:
            assert torch.equal(v, before[k]), "parameter %s changed in a step that failed" % k
    ta.eng._seed = 5; ta.eng._calls = 0                 # same dropout stream as the undisturbed run below
    la = ta.train_step(b)
    mb, _, _, hp = build(TINY96, compute_dtype="fp32")
    mb.train()
    tb = HipTrainer(mb, hp)
    tb.eng._seed = 5; tb.eng._calls = 0
    lb = tb.train_step(b)
    torch.cuda.synchronize()
    assert torch.allclose(la, lb, rtol=1e-5, atol=1e-6)
    sb = mb.state_dict()
    for k, v in ma.state_dict().items():
        if "running_" in k or "num_batches" in k:      # (BatchNorm statistics saw the failed step's forward pass as well)
            continue
        d = float((v.double() - sb[k].double()).abs().max())
        assert d <= 1e-4 * (1.0 + float(sb[k].double().abs().max())), (k, d)      # (atomics reorder the gradient sums from run to run)
