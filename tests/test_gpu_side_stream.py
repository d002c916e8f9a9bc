"""Side stream of the decoder backward (include/b2s_hip.h: b2s_model_set_side_stream): the dK / dV kernel of every encoder-decoder attention
(transformer/attention.py:72-92 under autograd) runs on a stream of the caller that is idle during the call.  Same kernels, same reduction
orders: d(memory) and the memory-side kv weight gradients are BIT-IDENTICAL with and without it, and the trainer step with the switch
on / off stays inside the run-to-run bars of one schedule."""
import ctypes as C
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import synth, make_config, TINY96
from test_gpu_model import build, dev_batch
from test_gpu_dropout_parity import with_dropout, worst_direction


@pytest.mark.parametrize("ragged", [False, True])
def test_side_stream_results_are_bit_identical(ragged):
    over = with_dropout(TINY96)
    cfg0 = make_config(over)
    st = synth.synthetic_state(cfg0, 1234)
    nb = synth.synthetic_batch(cfg0, B=4, S=19, T=150, seed=5, in_lens=[19, 12, 19, 3], tgt_lens=[150, 70, 128, 2])
    m, cfg, _, hp = build(over, compute_dtype="bf16", state_edit=lambda s: s.update(st))
    m.train()
    eng = m.engine()
    b = dev_batch(nb)
    in32, tgt32 = b["input_lengths"].int(), b["target_lengths"].int()
    mem, c_enc = eng.encoder_forward(b["inputs"], in32, b.get("input_spk_ids"), b.get("input_language_vecs"), True, 11, False)
    side = torch.cuda.Stream()
    host = [int(x) for x in nb["target_lengths"]] if ragged else None
    kv_names = [n for n in eng.param_offsets if "encdec_attentions" in n and "kv_transform.weight" in n]
    assert len(kv_names) == cfg.n_decoder_layer
    outs = []
    for use_side in (False, True, True):
        from b2s_hip import lib as L
        L.check(eng.lib.b2s_model_set_side_stream(eng.handle, C.c_void_p(side.cuda_stream) if use_side else None))
        mels, stop, c = eng.decoder_forward(mem, in32, b["mel_targets"], tgt32, True, 13, True, padded_unobserved=True, target_lengths_host=host)
        g = torch.Generator(device=mels.device).manual_seed(1)
        dm = torch.randn(mels.shape, generator=g, device=mels.device)
        ds = torch.randn(stop.shape, generator=g, device=mels.device)
        valid = (torch.arange(mels.shape[1], device=mels.device)[None, :] < b["target_lengths"][:, None])
        dm = dm * valid[..., None]; ds = ds * valid
        eng.begin_backward()
        dmem = eng.decoder_backward(c, dm, ds, mem.shape)
        torch.cuda.synchronize()
        flat = eng._gflat
        kv = torch.cat([flat[eng.param_offsets[n][0]:eng.param_offsets[n][0] + eng.param_offsets[n][1]] for n in kv_names]).clone()
        outs.append((dmem.clone(), kv))
        c.free()
    L.check(eng.lib.b2s_model_set_side_stream(eng.handle, None))
    (d0, k0), (d1, k1), (d2, k2) = outs
    assert float(d0.abs().max()) > 0 and float(k0.abs().max()) > 0
    assert torch.equal(d0, d1) and torch.equal(d1, d2), "d(memory) is bit-identical with the dK / dV kernels on the side stream"
    assert torch.equal(k0, k1) and torch.equal(k1, k2), "kv weight gradients are bit-identical"


def test_side_stream_trainer_step_within_run_to_run_bars(monkeypatch):
    """Fused trainer, dropout on, bf16: two steps with the side stream (the default) against two without."""
    from b2s_hip.trainer import HipTrainer
    over = with_dropout(TINY96)
    cfg0 = make_config(over)
    st = synth.synthetic_state(cfg0, 1234)
    nb = synth.synthetic_batch(cfg0, B=4, S=19, T=150, seed=5, in_lens=[19, 12, 19, 3], tgt_lens=[150, 70, 128, 2])
    res = []
    for use_side in (True, False):
        m, cfg, _, hp = build(over, compute_dtype="bf16", state_edit=lambda s: s.update(st))
        m.train()
        tr = HipTrainer(m, hp, dist=False, side_stream=use_side)
        assert tr.side_stream == use_side
        bb = dev_batch(nb)
        bb["target_lengths_host"] = [int(x) for x in nb["target_lengths"]]
        grads, vals = [], []
        tr.grad_probe = lambda flat, wire: grads.append(flat.detach().clone())
        for _ in range(2):
            vals.append(tr.train_step(bb).detach().clone())
        torch.cuda.synchronize()
        g0 = {n: grads[0][tr.eng.param_offsets[n][0]:tr.eng.param_offsets[n][0] + tr.eng.param_offsets[n][1]].cpu() for n in tr.eng.param_offsets}
        res.append((vals, g0))
    (va, ga), (vb, gb) = res
    for i in range(va[0].numel()):
        assert abs(float(va[0][i]) - float(vb[0][i])) <= 2e-5 * abs(float(va[0][i])) + 1e-9, (i, va[0], vb[0])
    e, n = worst_direction(ga, gb)
    print("worst per-tensor gradient difference side stream on vs off %.2e (%s)" % (e, n))
    assert e < 1e-2, (e, n)


def test_streams_are_shared_by_every_trainer_of_a_process():
    """HIP assigns a stream its hardware queue when the stream is created; a fresh second stream per model and a fresh encoder stream per trainer put the
    third trainer of one process on the main stream's queue (13.3 instead of 6.5 ms per step, tools/leg_order_lab.py -- bench.py --gpus N builds one
    trainer per leg).  Both are process-wide per device now: every model / trainer sees the streams of the first."""
    from b2s_hip.trainer import HipTrainer
    over = TINY96
    cfg0 = make_config(over)
    st = synth.synthetic_state(cfg0, 1234)
    nb = synth.synthetic_batch(cfg0, B=3, S=11, T=23, seed=7, in_lens=[11, 7, 4], tgt_lens=[23, 15, 9])
    seen = []
    for _ in range(3):
        m, cfg, _, hp = build(over, compute_dtype="bf16", state_edit=lambda s: s.update(st))
        m.train()
        tr = HipTrainer(m, hp, dist=False)
        v = tr.train_step(dev_batch(nb))
        torch.cuda.synchronize()
        assert np.isfinite(float(v[0]))
        seen.append((tr._enc_stream.cuda_stream, tr.lib.b2s_model_second_stream(tr.eng.handle)))
        tr.close()
    assert seen[0][0] and seen[0][1] and seen[0] == seen[1] == seen[2], seen
