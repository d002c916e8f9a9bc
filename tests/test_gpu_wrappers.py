"""The reference's own wrappers around the model (train.py:122-127): `DistributedDataParallel(m, device_ids=[local_rank])` under --ddp,
`nn.DataParallel(m)` otherwise.  The literal loop of train.py:171-174,188-189 runs through both and is compared with the CPU oracle:
DDP -> mean of the per-rank gradients applied by torch.optim.Adam on every rank (2 gloo ranks on device 0), DataParallel (one visible
device: the wrapper calls the module directly) -> the plain single-process step."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import b2s_oracle as O                     # checker only
from oracle import synth, make_config, TINY
from gpu_util import DEV
from test_gpu_model import build, dev_batch

STEPS = 2


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _reference_loop(m, hp, batches):
    """train.py:130-131,171-174,188-189 verbatim (m is the wrapped model)."""
    from functools import partial
    from transformer import tacotron
    optim = torch.optim.Adam(m.parameters(), lr=hp.max_lr, eps=hp.adam_eps)
    sched = torch.optim.lr_scheduler.LambdaLR(optim, lr_lambda=partial(tacotron.learning_rate_schedule, hp=hp))
    losses = None
    for batch in batches:
        outputs = m(**batch)
        losses = tacotron.compute_loss(m, batch['mel_targets'], batch['target_lengths'], outputs, hp)
        optim.zero_grad()
        losses['loss'].backward()
        optim.step()
        sched.step()
    return losses


def _oracle_steps(cfg, seed, world):
    P = O.to_torch_state(synth.synthetic_state(cfg, seed), requires_grad=True)
    opt = {}
    names = [n for n in P if O.is_parameter(n)]
    for step in range(STEPS):
        gsum = None
        for rank in range(world):
            ob = O.to_torch_batch(synth.synthetic_batch(cfg, B=2, S=9, T=14, seed=100 + 10 * step + rank))
            o = O.tacotron_forward(P, cfg, ob, train=True)
            loss = O.compute_loss(P, cfg, ob["mel_targets"], ob["target_lengths"], o)["loss"]
            g = torch.autograd.grad(loss, [P[n] for n in names], allow_unused=True)
            g = [x if x is not None else torch.zeros_like(P[n]) for x, n in zip(g, names)]
            gsum = g if gsum is None else [a + b for a, b in zip(gsum, g)]
        with torch.no_grad():
            O.adam_step(P, {n: x / world for n, x in zip(names, gsum)}, opt, step, cfg)
    return {n: P[n].detach().numpy() for n in names}


def _check(got, ref):
    for n, r in ref.items():
        assert abs(np.linalg.norm(got[n]) - np.linalg.norm(r)) <= 2e-4 * np.linalg.norm(r) + 1e-5, n
        assert np.abs(got[n] - r).max() < 4.5e-3, n               # <= STEPS x lr on ill-conditioned (|g| ~ 0) elements


def _ddp_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    m, cfg, st, hp = build(TINY, seed=1234 if rank == 0 else 999)      # rank 1 starts different: DDP broadcasts rank 0's at construction
    m = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], output_device=0)
    m.train()
    batches = [dev_batch(synth.synthetic_batch(cfg, B=2, S=9, T=14, seed=100 + 10 * step + rank)) for step in range(STEPS)]
    losses = _reference_loop(m, hp, batches)
    assert np.isfinite(float(losses['loss']))
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **{k: v.detach().cpu().numpy() for k, v in m.module.state_dict().items()})
    dist.destroy_process_group()


def test_reference_loop_under_distributed_data_parallel(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_ddp_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = dict(np.load(os.path.join(str(tmp_path), "rank0.npz")))
    r1 = dict(np.load(os.path.join(str(tmp_path), "rank1.npz")))
    cfg = make_config(TINY)
    ref = _oracle_steps(cfg, 1234, 2)
    for n in ref:
        assert np.array_equal(r0[n], r1[n]), n                      # replicas stay bit-identical
    _check(r0, ref)
    # DDP(broadcast_buffers=True): rank 1's BatchNorm buffers entered the last forward as rank 0's of the step before
    for k in r0:
        if "batchnorm" in k and "num_batches" in k:
            assert int(r0[k]) == int(r1[k]), k


def test_reference_loop_under_data_parallel():
    """`nn.DataParallel(m)` -- the reference's default when --ddp is absent.  On a one-GPU box the wrapper forwards to the module
    itself (no replication); with several visible devices it would replicate modules whose parameters are bound to one engine, which
    the wrapper cannot express -- restrict it with device_ids=[0] there."""
    m, cfg, st, hp = build(TINY, seed=1234)
    m = torch.nn.DataParallel(m, device_ids=[0])
    m.train()
    batches = [dev_batch(synth.synthetic_batch(cfg, B=2, S=9, T=14, seed=100 + 10 * step)) for step in range(STEPS)]
    for b in batches:
        b["names"] = ["utt%d" % i for i in range(2)]              # the dataloader's extra key travels through **batch (tacotron.py:126)
    losses = _reference_loop(m, hp, batches)
    assert np.isfinite(float(losses['loss']))
    torch.cuda.synchronize()
    got = {k: v.detach().cpu().numpy() for k, v in m.module.state_dict().items()}
    _check(got, _oracle_steps(make_config(TINY), 1234, 1))
