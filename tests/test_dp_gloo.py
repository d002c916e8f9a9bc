"""Data-parallel path on CPU: 2 gloo ranks.  The bucketed, stage-driven all-reduce must (a) reduce every element of
the flat gradient buffer exactly once and (b) give mean-of-per-rank gradients, i.e. the reference's DDP semantics
(per-rank loss normalisation, train.py:125 + common.py:86), checked with the CPU oracle's gradients."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "few-shot-transformer-tts_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import b2s_oracle as O          # noqa: E402
from oracle import synth, make_config, TINY  # noqa: E402


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _stage_layout(names, sizes):
    """Flat layout exactly as HipEngine lays gradients out: parameters sorted by backward stage."""
    import hyperparams
    from b2s_hip.engine import HipEngine
    from transformer.tacotron import Tacotron
    hp = hyperparams.hparams
    hp.override_from_dict(hyperparams.DEFAULTS)
    hp.parse(TINY)
    eng = HipEngine(Tacotron(hp), hp)
    order = sorted(range(len(names)), key=lambda i: (eng.stage_of(names[i]), i))
    off, ranges, offsets = 0, {}, {}
    for i in order:
        st = eng.stage_of(names[i])
        lo, hi = ranges.get(st, (off, off))
        ranges[st] = (min(lo, off), off + sizes[i])
        offsets[names[i]] = (off, sizes[i])
        off += sizes[i]
    return ranges, offsets, off, eng.n_stages()


def _worker(rank, world, port, out, payload="fp32"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from b2s_hip.dp import GradBucketer
    torch.set_num_threads(2)
    cfg = make_config(TINY)
    P = O.to_torch_state(synth.synthetic_state(cfg, 1234), requires_grad=True)
    batch = O.to_torch_batch(synth.synthetic_batch(cfg, B=2, S=9, T=14, seed=100 + rank))
    o = O.tacotron_forward(P, cfg, batch, train=True)
    loss = O.compute_loss(P, cfg, batch["mel_targets"], batch["target_lengths"], o)["loss"]
    names = [n for n in P if O.is_parameter(n)]
    grads = torch.autograd.grad(loss, [P[n] for n in names], allow_unused=True)
    grads = [g if g is not None else torch.zeros_like(P[n]) for g, n in zip(grads, names)]
    ranges, offsets, total, n_stages = _stage_layout(names, [g.numel() for g in grads])
    flat = torch.zeros(total)
    for n, g in zip(names, grads):
        off, sz = offsets[n]
        flat[off:off + sz] = g.flatten()
    local = flat.clone()
    b = GradBucketer(flat, ranges, n_stages, bucket_elems=60000, dist=dist, payload=payload)
    b.begin_step()
    for st in range(n_stages):                 # the engine reports stages in backward order
        b.stage_done(st)
    b.finish()
    covered = sorted(b.launched)
    assert covered[0][0] == 0 and covered[-1][1] == total
    assert all(a[1] == c[0] for a, c in zip(covered, covered[1:])), "buckets must tile the buffer exactly once"
    assert 1 < len(covered) < n_stages, "small stages are merged into buckets"
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    ref = sum(gathered) / world
    if payload == "fp32":
        assert torch.allclose(flat / world, ref, atol=1e-7)
    else:       # bf16 wire format: every rank's contribution and the sum are rounded to 8 mantissa bits
        wire = sum(g.to(torch.bfloat16) for g in gathered).to(torch.float32) / world
        assert torch.allclose(flat / world, wire, atol=1e-7)
        assert float((flat / world - ref).norm() / ref.norm()) < 6e-3
    if rank == 0:
        np.save(out, (flat / world).numpy())
    dist.destroy_process_group()


def test_bucketed_allreduce_mean_of_rank_gradients(tmp_path):
    out = str(tmp_path / "mean.npy")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    mean = np.load(out)
    assert np.isfinite(mean).all() and np.abs(mean).max() > 0


def test_bucketed_allreduce_bf16_payload(tmp_path):
    """grad_payload="bf16": same buckets, gradients travel as bf16 (half the bytes per all-reduce), result = bf16 sum of the
    bf16-rounded rank gradients, within 0.6 % of the exact mean in norm."""
    out = str(tmp_path / "mean16.npy")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out, "bf16"), nprocs=2, join=True)
    mean = np.load(out)
    assert np.isfinite(mean).all() and np.abs(mean).max() > 0


def _bn_worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from b2s_hip.dp import broadcast_buffers
    torch.manual_seed(rank)
    bufs = []
    for c in (512, 512, 80):
        bufs += [torch.randn(c), torch.rand(c) + 0.5, torch.tensor(7 + rank, dtype=torch.long)]     # running_mean, running_var, num_batches_tracked
    mine = [b.clone() for b in bufs]
    broadcast_buffers(bufs, dist, 0)
    torch.manual_seed(0)
    for c, i in ((512, 0), (512, 3), (80, 6)):
        rm, rv = torch.randn(c), torch.rand(c) + 0.5
        assert torch.equal(bufs[i], rm) and torch.equal(bufs[i + 1], rv) and int(bufs[i + 2]) == 7
        assert bufs[i + 2].dtype == torch.long and bufs[i + 2].dim() == 0
        if rank == 0:
            assert torch.equal(bufs[i], mine[i])
    dist.destroy_process_group()


def test_buffer_broadcast_every_step_semantics():
    """DDP(broadcast_buffers=True), train.py:125: every rank's BatchNorm buffers become rank 0's (one flat float message + one int64
    message; shapes, dtypes and 0-dim counters preserved)."""
    mp.spawn(_bn_worker, args=(2, _free_port()), nprocs=2, join=True)


def _worker_rs_ag(rank, world, port, payload):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from b2s_hip.dp import GradBucketer
    torch.set_num_threads(2)
    # a flat layout as HipEngine makes it: every stage a run of 64-element slots
    g = torch.Generator().manual_seed(5)
    n_stages, off, ranges = 9, 0, {}
    for st in range(n_stages):
        n = 64 * int(torch.randint(1, 40, (1,), generator=g))
        ranges[st] = (off, off + n)
        off += n
    total = off
    local = torch.randn(total, generator=torch.Generator().manual_seed(100 + rank))
    flat = local.clone()
    b = GradBucketer(flat, ranges, n_stages, bucket_elems=3000, dist=dist, payload=payload, mode="rs_ag")
    plan = b.plan()
    assert plan[0][0] == 0 and plan[-1][1] == total and all(a[1] == c[0] for a, c in zip(plan, plan[1:])) and 1 < len(plan) < n_stages
    b.begin_step()
    for st in range(n_stages):
        b.stage_done(st)
    b.finish()
    assert b.launched == plan, "the buckets a step launches are the planned ones (the optimizer's shard was bound to the plan)"
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    ref = sum(gathered)
    own = b.owned_ranges()
    # every element is owned by exactly one rank ...
    cover = torch.zeros(total)
    for r in range(world):
        for lo, hi in b.owned_ranges(r):
            cover[lo:hi] += 1
    assert bool((cover == 1).all())
    # ... and the owner holds the sum over ranks there (fp32 wire: exactly; bf16 wire: each contribution and the sum rounded to bf16)
    for lo, hi in own:
        if payload == "fp32":
            assert torch.allclose(flat[lo:hi], ref[lo:hi], atol=1e-6)
        else:
            rb = sum(x.to(torch.bfloat16).float() for x in gathered)
            assert torch.allclose(flat[lo:hi], rb[lo:hi], atol=2e-2, rtol=2e-2)
    # parameter all-gather: every rank writes its owned slices of the wire, afterwards every rank holds every owner's values
    wire = torch.full((total,), -1.0)
    for lo, hi in own:
        wire[lo:hi] = float(rank + 1)
    b.all_gather_params(wire)
    want = torch.zeros(total)
    for r in range(world):
        for lo, hi in b.owned_ranges(r):
            want[lo:hi] = float(r + 1)
    assert torch.equal(wire, want)
    dist.destroy_process_group()


def test_reduce_scatter_all_gather_mode_two_ranks():
    """dp_mode = "rs_ag" on 2 gloo ranks: the planned buckets are the launched ones, every element of the flat buffer is owned by exactly one
    rank, the owner's slice holds the sum over ranks after the in-place reduce-scatter (fp32 and bf16 wire), and the in-place parameter
    all-gather leaves every rank with every owner's values (train.py:125,188-189: all replicas step to the same parameters)."""
    for payload in ("fp32", "bf16"):
        mp.spawn(_worker_rs_ag, args=(2, _free_port(), payload), nprocs=2, join=True)


def _worker_rs_ag_moments(rank, world, port):
    """HipTrainer.gather_optimizer_state / the state_dict guard, on the trainer's own methods with a CPU stand-in for its device state."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from types import SimpleNamespace
    from b2s_hip.dp import GradBucketer
    from b2s_hip.trainer import HipTrainer
    from b2s_hip import lib as L
    n_stages, off, ranges = 7, 0, {}
    g = torch.Generator().manual_seed(11)
    for st in range(n_stages):
        n = 64 * int(torch.randint(1, 30, (1,), generator=g))
        ranges[st] = (off, off + n)
        off += n
    total = off
    b = GradBucketer(torch.zeros(total), ranges, n_stages, bucket_elems=2000, dist=dist, mode="rs_ag")
    # what the sharded optimizer leaves behind: true moments on the owned slices, zeros (never updated) everywhere else
    true_m = torch.randn(total, generator=torch.Generator().manual_seed(1))
    true_v = torch.rand(total, generator=torch.Generator().manual_seed(2))
    m, v = torch.zeros(total), torch.zeros(total)
    for lo, hi in b.owned_ranges():
        m[lo:hi], v[lo:hi] = true_m[lo:hi], true_v[lo:hi]
    t = SimpleNamespace(bucketer=b, exp_avg=m, exp_avg_sq=v, dist=dist, global_step=3, sync=lambda: None)
    try:                                     # a rank-0-only save without the gather would write zero moments for the other rank's half
        HipTrainer.state_dict(t)
        raise AssertionError("state_dict under rs_ag must refuse before gather_optimizer_state")
    except L.B2SError as e:
        assert "gather_optimizer_state" in str(e)
    HipTrainer.gather_optimizer_state(t)
    assert torch.equal(t.exp_avg, true_m) and torch.equal(t.exp_avg_sq, true_v), "every rank holds the whole Adam state after the gather"
    assert t._moments_gathered_at == 3
    t.global_step = 4                        # one more step: the gathered copy is stale again
    try:
        HipTrainer.state_dict(t)
        raise AssertionError("a later step invalidates the gathered moments")
    except L.B2SError:
        pass
    dist.destroy_process_group()


def test_sharded_optimizer_state_is_gathered_before_a_checkpoint():
    """dp_mode = "rs_ag": Adam moments live on the owning rank only; HipTrainer.gather_optimizer_state (a collective) makes every rank's copy whole,
    and state_dict refuses to serialise a sharded state (utils/checkpoint.py:27 is called on rank 0 only: train.py:183,225)."""
    mp.spawn(_worker_rs_ag_moments, args=(2, _free_port()), nprocs=2, join=True)
