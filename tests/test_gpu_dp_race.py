"""Single-GPU race detector for the data-parallel exchange (train.py:33-41,122-125: DDP's gradient all-reduce overlapped with backward).

The 2-rank tests use gloo, which synchronises the device, and a 1-rank RCCL group launches no collective kernel -- neither can fail on a
stream-ordering bug.  Here torch.distributed is replaced by a stand-in whose all_reduce is an ASYNCHRONOUS DEVICE op on a communication
stream of its own, ordered exactly like RCCL's (the communication stream waits for the caller's current stream at launch; work.wait()
makes the caller's stream wait for the collective): in place x2, i.e. "world = 2 with identical ranks", so with grad_scale = 1/2 the step
must equal the step without a process group.  A spin kernel in front of every bucket's op delays the collective (the optimizer must really
wait for it); without the spin the collective runs as early as its ordering allows (a gradient written after its bucket was reduced shows
up un-doubled).  Checked per element on the gradients the optimizer consumes, and on the Adam moments / parameters after the step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import DEV  # noqa: E402
from oracle import synth, make_config  # noqa: E402


class _Work(object):
    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)
        return True


class FakeDist(object):
    """world = 2, both ranks identical: all_reduce(sum) == x2, asynchronously on a private stream."""

    def __init__(self, spin_cycles):
        self.comm = torch.cuda.Stream()
        self.spin = int(spin_cycles)
        self.calls = 0

    def get_world_size(self, group=None):
        return 2

    def get_rank(self, group=None):
        return 1            # (not rank 0: no start-up message)

    def broadcast(self, t, src, group=None):
        return None

    def all_reduce(self, buf, group=None, async_op=False):
        self.calls += 1
        self.comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.comm):
            if self.spin:
                torch.cuda._sleep(self.spin)
            buf.mul_(2)
            ev = torch.cuda.Event()
            ev.record(self.comm)
        w = _Work(ev)
        if not async_op:
            w.wait()
        return w


def _build(seed):
    import hyperparams
    from hyperparams import hparams as hp
    from transformer.tacotron import Tacotron
    hp.override_from_dict(hyperparams.DEFAULTS)
    hp.parse("compute_dtype=bf16")
    cfg = make_config("")
    st = synth.synthetic_state(cfg, seed)
    m = Tacotron(hp)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in st.items()})
    return m.to(DEV).train(), hp, cfg


@pytest.mark.parametrize("spin", [0, 400000])
def test_exchange_ordering_bf16_wire_is_exact(spin):
    """bf16 payload, optimizer fed from the wire buffer: the fp32 buffer keeps this rank's gradients, so IN ONE RUN
    wire == 2 * bf16(local gradient) must hold bit for bit on every element (a bucket packed before its gradients were complete, or a gradient
    written after the pack, breaks it), and after the step exp_avg == beta1 * exp_avg + (1 - beta1) * (wire / 2 [+ l2 * p]) (an optimizer
    that did not wait for the delayed collective sees the un-doubled wire)."""
    from b2s_hip.trainer import HipTrainer
    from b2s_hip import lib as L
    torch.manual_seed(1234)
    m, hp, cfg = _build(3)
    fd = FakeDist(spin)
    tr = HipTrainer(m, hp, bucket_mb=8.0, grad_payload="bf16", dist=fd)
    try:
        assert tr.overlap_encoder and tr.world == 2 and tr.bucketer.consume_wire     # encoder on its own stream, tail policy on: the schedules that could race
        seen = []
        tr.grad_probe = lambda flat, wire: seen.append((flat.detach().clone(), wire.detach().clone()))
        names = [n for n, _ in m.named_parameters()]
        for i in range(8):
            nb = synth.synthetic_batch(cfg, 6, 60, 200, seed=40 + i, n_spk=1, n_lang=1)
            m_before = tr.exp_avg.clone()
            tr.train_step({k: (torch.from_numpy(np.asarray(v)).to(DEV) if not isinstance(v, list) else v) for k, v in nb.items()})
            torch.cuda.synchronize()
            assert tr.last_step_tail_update and tr.bucketer.launched[0][0] == 0 and any(hi == tr.bucketer.split for _, hi in tr.bucketer.launched)     # decoder / postnet update beside the encoder backward, behind ITS buckets only
            flat, wire = seen[-1]
            assert fd.calls >= 4 * (i + 1)
            want = (flat.to(torch.bfloat16).float() * 2).to(torch.bfloat16)
            bad = int((wire.view(torch.int16) != want.view(torch.int16)).sum())
            assert bad == 0, "step %d: %d wire elements differ from 2 x bf16(local gradient)" % (i, bad)
            # non-L2 parameters (biases, LayerNorm / BatchNorm, embeddings): exp_avg follows the wire exactly
            for n in names:
                if ("layer_norm" in n or n.endswith(".bias") or "batchnorm" in n) and n in tr.eng.param_offsets:
                    off, cnt = tr.eng.param_offsets[n]
                    ref = 0.9 * m_before[off:off + cnt] + 0.1 * 0.5 * wire[off:off + cnt].float()
                    d = float((tr.exp_avg[off:off + cnt] - ref).abs().max())
                    assert d <= 1e-6 * float(ref.abs().max()) + 1e-12, (i, n, d)
    finally:
        tr.close()                                                         # (the world-2 trainer selected the data-parallel tile policy process-wide)
        assert not tr._set_tile_policy


@pytest.mark.parametrize("spin", [0, 400000])
def test_exchange_ordering_fp32_payload_vs_no_exchange(spin):
    """fp32 payload (in-place all-reduce of the gradient buffer): compared across two runs from the same seeds -- per tensor, the exchanged
    gradient is 2 x the gradient of the run without a process group up to the bf16 step's run-to-run noise (atomics, tile shapes); a range
    that missed its collective, or was reduced before it was complete, is off by a factor of two."""
    from b2s_hip.trainer import HipTrainer
    from b2s_hip import lib as L
    steps = 4
    grads, moms, offs = {}, {}, None
    try:
        for arm in ("plain", "dp"):
            torch.manual_seed(1234)
            m, hp, cfg = _build(3)
            L.check(L.load().b2s_gemm_set_tile_policy(4))                 # the same tiles in both arms
            fd = FakeDist(spin) if arm == "dp" else None
            tr = HipTrainer(m, hp, bucket_mb=8.0, grad_payload="fp32", dist=fd)
            seen = []
            tr.grad_probe = lambda flat, wire, seen=seen: seen.append(flat.detach().clone())
            for i in range(steps):
                nb = synth.synthetic_batch(cfg, 6, 60, 200, seed=40 + i, n_spk=1, n_lang=1)
                tr.train_step({k: (torch.from_numpy(np.asarray(v)).to(DEV) if not isinstance(v, list) else v) for k, v in nb.items()})
            torch.cuda.synchronize()
            grads[arm], moms[arm], offs = seen, tr.exp_avg.clone(), dict(tr.eng.param_offsets)
            del tr, m
    finally:
        L.check(L.load().b2s_gemm_set_tile_policy(0))
    worst = (0.0, None)
    for n, (off, cnt) in offs.items():
        a, b = grads["dp"][0][off:off + cnt] * 0.5, grads["plain"][0][off:off + cnt]
        e = float((a - b).norm() / (b.norm() + 1e-20))
        worst = max(worst, (e, n))
        assert e < 0.05, (n, e)                                           # (a missed / early collective: 0.5 .. 1.0)
    rel = float((moms["dp"] - moms["plain"]).norm() / (moms["plain"].norm() + 1e-30))
    print("fp32 payload, spin %d: worst per-tensor relative error of the exchanged gradient %.2e (%s); exp_avg after %d steps %.2e" % (spin, worst[0], worst[1], steps, rel))
    assert rel < 0.05, rel


def test_bf16_wire_tracks_fp32_wire_over_200_steps():
    """Convergence-level evidence for the bf16 gradient wire (HipTrainer(grad_payload="bf16"); the default is the reference's fp32 mean,
    train.py:125): 200 optimizer steps of the tiny model on 8 recurring batches, dropout live, same seeds, through the exchange path with
    the fp32 wire and with the bf16 wire (2 identical ranks: the wire carries 2 x bf16(g), the optimizer halves it).  The two loss curves
    must coincide within the step-to-step noise of the bf16 compute mode: mean relative difference of the last 50 losses < 2 %, both
    trained (loss fell by > 30 %)."""
    import hyperparams
    from hyperparams import hparams as hp
    from transformer.tacotron import Tacotron
    from b2s_hip.trainer import HipTrainer
    from oracle import TINY96
    over = TINY96.replace("transformer_dropout_rate=0.0,decoder_dropout_rate=0.0", "transformer_dropout_rate=0.1,decoder_dropout_rate=0.5")
    curves = {}
    for wire in ("fp32", "bf16"):
        torch.manual_seed(77)
        hp.override_from_dict(hyperparams.DEFAULTS)
        hp.parse(over)
        hp.parse("compute_dtype=bf16")
        cfg = make_config(over)
        st = synth.synthetic_state(cfg, 5)
        m = Tacotron(hp)
        m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in st.items()})
        m = m.to(DEV).train()
        tr = HipTrainer(m, hp, bucket_mb=0.25, grad_payload=wire, dist=FakeDist(0))
        try:
            batches = []
            for i in range(8):
                nb = synth.synthetic_batch(cfg, 6, 24, 60, seed=100 + i)
                batches.append({k: (torch.from_numpy(np.asarray(v)).to(DEV) if not isinstance(v, list) else v) for k, v in nb.items()})
            ls = []
            for step in range(200):
                ls.append(tr.train_step(batches[step % 8])[0])
            torch.cuda.synchronize()
            curves[wire] = np.array([float(x) for x in ls])
            assert (tr.bucketer.wire is not None) == (wire == "bf16")
        finally:
            tr.close()
        del tr, m
    a, b = curves["fp32"], curves["bf16"]
    print("bf16 wire vs fp32 wire, 200 steps: loss %.4f -> %.4f (fp32 wire), %.4f -> %.4f (bf16 wire); mean |rel diff| of the last 50: %.4f"
          % (a[:8].mean(), a[-8:].mean(), b[:8].mean(), b[-8:].mean(), float(np.mean(np.abs(a[-50:] - b[-50:]) / a[-50:]))))
    assert a[-8:].mean() < 0.7 * a[:8].mean() and b[-8:].mean() < 0.7 * b[:8].mean()
    assert float(np.mean(np.abs(a[-50:] - b[-50:]) / a[-50:])) < 0.02


def test_dataparallel_over_several_devices_is_refused():
    """nn.DataParallel replicas (train.py:126-127 on a multi-GPU box) are refused with a message instead of running the engine on tensors
    of another device.  (A one-GPU box cannot create real replicas: the flag torch's replicate() sets is set by hand.)"""
    from b2s_hip.lib import B2SError
    m, hp, cfg = _build(5)
    m._is_replica = True
    nb = synth.synthetic_batch(cfg, 2, 20, 40, seed=1, n_spk=1, n_lang=1)
    b = {k: (torch.from_numpy(np.asarray(v)).to(DEV) if not isinstance(v, list) else v) for k, v in nb.items()}
    with pytest.raises(B2SError, match="one process per GPU"):
        m(**b)
