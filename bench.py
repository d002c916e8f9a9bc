#!/usr/bin/env python3
"""Benchmark of the Transformer-TTS hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

A "step" is one full training step (forward + loss + backward + gradient all-reduce + Adam) of the default
83.5 M-parameter model on one packed batch per GPU, bf16 MFMA operands with fp32 accumulate / residual /
optimizer, dropout ON at the reference rates, synthetic data, TF-style random init.  Metric: padded mel
frames per second over the whole job.  Workloads:
  lj  BASELINE configs[1]: the LJSpeech-shaped batch B=14, S=114, T=582 (the mean batch the reference's packer produces),
      one speaker / language -- the default at --gpus 1;
  c3  BASELINE configs[2]: the 38-language byte2speech batch per rank, B=14, S=256 (multi-byte scripts), T=582, 572 speakers,
      identical shapes on every rank, different seeds -- the default at --gpus N > 1 (and a `c3` leg of the N = 1 line,
      through the data-parallel exchange path on a 1-rank RCCL group).
--mode decode benchmarks the autoregressive loop instead.  One JSON line is printed by rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  With RCCL's streams in the process the engine's
# second stream landed on the main stream's hardware queue and the two serialised (single-rank RCCL run: 10.9 ms per step
# against 9.98 with 8 queues; no effect without RCCL).  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# The launcher's choice (the package itself never touches NCCL_*), data-parallel runs: cap RCCL's channel count -- every channel holds a CU
# while a collective runs (profiles/r03_cu_loss.txt) and the exchange needs 22-45 GB/s per GPU.  Must precede the process group.
if os.environ.get("WORLD_SIZE", "1").strip().isdigit() and int(os.environ.get("WORLD_SIZE", "1")) > 1:
    os.environ.setdefault("NCCL_MAX_NCHANNELS", "16")
ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "few-shot-transformer-tts_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_FP32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def fwd_flops(B, S, T, De=512, Fe=2048, Dd=768, Fd=3072, Le=6, Ld=6):
    """BASELINE.md section 3 (validated against torch.utils.flop_counter): forward FLOPs of one batch."""
    return (Le * (B * S * (8 * De * De + 4 * De * Fe) + 4 * B * S * S * De)
            + Ld * (B * T * (12 * Dd * Dd + 4 * Dd * Fd) + 4 * B * S * Dd * Dd + 4 * B * T * T * Dd + 4 * B * T * S * Dd)
            + 2 * B * T * (80 * 256 + 256 * 256 + 256 * Dd) + 2 * B * T * Dd * 81
            + 10 * B * T * (80 * 512 + 3 * 512 * 512 + 512 * 80) + 2 * B * (2 * 128 * 128 + 100 * 128))


def make_batch(cfg, B, S, T, seed, device, n_spk=1, n_lang=1):
    from benchdata import synthetic_batch
    nb = synthetic_batch(cfg, B, S, T, seed=seed, n_spk=n_spk, n_lang=n_lang)      # (1, 1) = LJSpeech: one speaker, en-us; (572, 38) = C3
    out = {k: (torch.from_numpy(np.asarray(v)).to(device) if not isinstance(v, list) else v) for k, v in nb.items()}
    # the dataloader's host copy of the lengths travels with the batch (b2s_hip/batching.py: DeviceStager does the same): the fused trainer keeps
    # the decoder's rows ragged from it -- sum(target_lengths) rows per row-wise kernel instead of B x T -- without a device-to-host read
    out["target_lengths_host"] = [int(x) for x in np.asarray(nb["target_lengths"])]
    for k in ("input_lengths", "target_lengths"):          # (and int32 device copies of the lengths, as DeviceStager.put)
        out[k + "_i32"] = torch.from_numpy(np.asarray(nb[k]).astype(np.int32)).to(device)
    return out


def cpu_baseline_train(budget_s=25.0):
    """The CPU oracle (fp32 PyTorch-CPU restatement of the reference's op sequence, live dropout, dense masks) timed
    on this host's cores at the reference's CPU config (B=4, S=100, T=600).  Reported baseline, not the target."""
    from oracle import b2s_oracle as O, synth, make_config
    cores = min(os.cpu_count() or 1, 16)          # more threads than this only adds contention for these shapes
    torch.set_num_threads(cores)
    cfg = make_config("")
    P = O.to_torch_state(synth.synthetic_state(cfg, 1), requires_grad=True)
    b = O.to_torch_batch(synth.synthetic_batch(cfg, 4, 100, 600, seed=0, in_lens=[100, 90, 80, 70],
                                               tgt_lens=[600, 550, 500, 450], n_spk=1, n_lang=1))
    opt = {}
    best, t_all, n = None, time.time(), 0
    for i in range(4):
        t0 = time.time()
        O.train_step(P, cfg, b, opt, i, train=True)
        dt = time.time() - t0
        if i > 0 or dt > budget_s:                     # a very slow host: keep the (cold) first step rather than run on
            best = dt if best is None else min(best, dt)
            n += 1
        if time.time() - t_all > budget_s and n >= 1:
            break
    return {"value": round(4 * 600 / best, 1), "unit": "padded mel-frames/s", "cores": cores, "kind": "port",
            "sample": "oracle train step fwd+loss+bwd+Adam, B=4 S=100 T=600, dropout on, 1 warm-up + best of %d" % n,
            "seconds_per_step": round(best, 3)}


def load_kernel_stats(suffix="train_bf16_kernel_stats.csv"):
    """Newest committed rocprofv3 --kernel-trace --stats summary of this command (profiles/rNN_<suffix>, tools/gpu_round_profiles.sh) ->
    (file name, steps in the profiled run, {kernel name: (launches per step, average ns)}).  The step count is read off a kernel that runs
    exactly once per step (k_shift_pe_fwd)."""
    import csv
    pdir = os.path.join(ROOT, "profiles")
    names = sorted(n for n in (os.listdir(pdir) if os.path.isdir(pdir) else []) if n.endswith(suffix) and n[0] == "r" and n[1:3].isdigit())
    if not names:
        return None, 0, {}
    rows = list(csv.DictReader(open(os.path.join(pdir, names[-1]))))
    steps = [int(r["Calls"]) for r in rows if "k_shift_pe_fwd" in r["Name"]]
    if not steps:
        return names[-1], 0, {}
    return names[-1], steps[0], {r["Name"]: (int(r["Calls"]) / steps[0], float(r["AverageNs"])) for r in rows}


def _sub_bench(extra, env=None):
    """Run another bench mode in a child process and return the fields of its JSON line that matter on the parent's line."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", "1"] + extra, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, **env) if env else None)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": (r.stderr or r.stdout)[-400:]}
    d = json.loads(lines[-1])
    keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "final_loss", "config", "roofline", "roofline_step", "rccl_ranks",
            "cpu_baseline", "value_incl_host_copy", "ms_per_step_incl_host_copy", "valid_frames_per_s")
    d = {k: d[k] for k in keep if k in d}
    if isinstance(d.get("roofline"), dict):
        d["roofline"] = {k: v for k, v in d["roofline"].items() if k not in ("variants", "isolated", "note")}
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", default="train", choices=["train", "decode", "finetune", "dropin"],
                    help="train: BASELINE configs[1-2]; decode: configs[3]; finetune: configs[4] (frozen encoder, guided "
                         "attention on, batches of B drawn from a 30-utterance pool); dropin: the reference's own loop (train.py:171-174,"
                         "188-190: m(**batch), compute_loss, zero_grad, backward, torch.optim.Adam.step, sched.step) on the drop-in modules")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--workload", default=None, choices=["lj", "c3"],
                    help="lj: BASELINE configs[1] (B=14, S=114, T=582, one speaker / language; default at --gpus 1); c3: configs[2], the "
                         "per-rank 38-language batch (B=14, S=256, T=582, 572 speakers; default at --gpus N > 1)")
    ap.add_argument("--batch", type=int, default=14)
    ap.add_argument("--S", type=int, default=None)
    ap.add_argument("--T", type=int, default=582)
    ap.add_argument("--hparams", default="", help="extra hparams overrides (experiments), e.g. transformer_dropout_rate=0.0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline-pass", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="train mode, 1 GPU: skip the decode / finetune / fp32_mode sub-benchmarks added to the JSON line")
    args = ap.parse_args()
    if args.workload is None:
        args.workload = "c3" if (args.gpus > 1 and args.mode == "train") else "lj"
    if args.S is None:
        args.S = 256 if args.workload == "c3" else 114
    n_spk, n_lang = (572, 38) if args.workload == "c3" else (1, 1)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start N ranks ourselves (one process per GPU, RCCL rendezvous on
        # 127.0.0.1) -- the same command line the driver uses
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the hot path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    force_dp = bool(os.environ.get("B2S_FORCE_DP"))     # 1-rank RCCL group: exercises the whole exchange path on one GPU
    if world > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.distributed.init_process_group("nccl", init_method="env://", device_id=device)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)

    if args.mode == "decode":
        from bench_decode import run_decode
        return run_decode(args, rank, world, device)

    from hyperparams import hparams as hp
    from transformer.tacotron import Tacotron, initialize_variables
    from b2s_hip.trainer import HipTrainer
    from b2s_hip import lib as L
    hp.parse("compute_dtype=%s" % args.dtype)
    if args.hparams:
        hp.parse(args.hparams)
    finetune = args.mode == "finetune"
    if finetune:
        hp.parse("freeze_encoder=true,guided_attention_weight=1.0")
    torch.manual_seed(0)                               # identical init on every rank (train.py:33)
    model = Tacotron(hp)
    initialize_variables(model)
    model = model.to(device).train()
    # gradient wire of the data-parallel exchange: the bf16 performance lines send bf16 (167 instead of 334 MB per step over the
    # point-to-point xGMI links; HipTrainer's own default is the reference's fp32 mean) -- stated in config.grad_payload
    payload = os.environ.get("B2S_GRAD_PAYLOAD") or ("bf16" if args.dtype == "bf16" else "fp32")
    dropin = args.mode == "dropin"
    init_state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()} if (world > 1 or force_dp) else None
    if dropin:
        # the literal reference loop (train.py:130-131,171-174,188-190) around the drop-in modules: autograd functions over the engine's
        # segments, torch.optim.Adam (foreach), LambdaLR -- what a user gets WITHOUT editing train.py
        from functools import partial as _partial
        from transformer.tacotron import compute_loss, learning_rate_schedule
        optim = torch.optim.Adam(model.parameters(), lr=hp.max_lr, eps=hp.adam_eps)
        sched = torch.optim.lr_scheduler.LambdaLR(optim, lr_lambda=_partial(learning_rate_schedule, hp=hp))
        trainer = None

        def train_step(b):
            outputs = model(**b)
            losses = compute_loss(model, b["mel_targets"], b["target_lengths"], outputs, hp)
            optim.zero_grad()
            losses["loss"].backward()
            optim.step()
            sched.step()
            return [losses["loss"].detach()]
    else:
        trainer = HipTrainer(model, hp, grad_payload=payload)
        train_step = trainer.train_step
    cfg = hp                                           # (synthetic_batch reads vocab_size / num_mels / max_num_* only)
    B, S, T = args.batch, args.S, args.T
    batch = make_batch(cfg, B, S, T, seed=rank, device=device, n_spk=n_spk, n_lang=n_lang)     # same shape on every rank, different data
    batches = [batch]
    if finetune:        # 30-utterance adaptation pool; every step trains on B of them (same padded shape, lengths vary)
        pool = make_batch(cfg, 30, S, T, seed=1000 + rank, device=device, n_spk=n_spk, n_lang=n_lang)
        rng = np.random.default_rng(rank)
        batches = []
        for _ in range(8):
            idx = torch.from_numpy(np.sort(rng.choice(30, B, replace=False))).to(device)
            bb = {k: (v.index_select(0, idx) if torch.is_tensor(v) else v) for k, v in pool.items()}
            bb["target_lengths_host"] = [pool["target_lengths_host"][int(i)] for i in idx.cpu()]
            batches.append(bb)

    def sync():
        if world > 1 or force_dp:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed(bs, steps, warmup, step_fn=None):
        """warmup untimed steps, then exactly `steps` steps between barrier + synchronize on both sides; MAX over ranks."""
        step_fn = step_fn or train_step
        v = None
        for i in range(warmup):
            v = step_fn(bs[i % len(bs)])
        sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for i in range(steps):
            v = step_fn(bs[i % len(bs)])
        e1.record()
        host = time.perf_counter() - t0                # launch loop only: ~= elapsed means the step is host- (launch-) bound
        sync()
        el = time.perf_counter() - t0
        dev = e0.elapsed_time(e1)
        if world > 1 or force_dp:
            t = torch.tensor([el], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            el = float(t.item())
        return v, el, host, dev

    vals, elapsed, host_s, dev_ms = timed(batches, args.steps, args.warmup)
    loss = float(vals[0])
    assert np.isfinite(loss), "training diverged: loss = %r" % loss
    # the two extremes of the reference's packer (dataloader.py:401-410: <= 8000 frames and B (S^2 + T^2) <= 7e6 per batch) next to the
    # typical batch: same model, same step, a few steps each
    extremes = {}
    if world == 1 and args.mode == "train" and args.workload == "lj" and (B, S, T) == (14, 114, 582) and not args.no_extras:
        for (b_, s_, t_) in ((32, 50, 250), (9, 158, 808)):
            _, el, _, _ = timed([make_batch(cfg, b_, s_, t_, seed=7, device=device)], max(5, args.steps // 2), 2)
            n_ = max(5, args.steps // 2)
            fl = 3.0 * fwd_flops(b_, s_, t_)
            extremes["B%d_S%d_T%d" % (b_, s_, t_)] = {
                "ms_per_step": round(el / n_ * 1e3, 3), "value": round(b_ * t_ * n_ / el, 1), "unit": "mel-frames/s",
                "roofline_step_frac": round(fl / (el / n_) / 1e12 / (PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_FP32_TFLOPS), 4)}

    out = None
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        step_flops = 3.0 * fwd_flops(B, S, T)
        if finetune:    # no encoder backward, no d(memory): drop 2 x encoder forward FLOPs and the cross-KV dX GEMMs
            enc = 6 * (B * S * (8 * 512 * 512 + 4 * 512 * 2048) + 4 * B * S * S * 512) + 2 * B * (2 * 128 * 128 + 100 * 128)
            step_flops -= 2.0 * enc + 6 * 4 * B * S * 768 * 768
        out = {"metric": "padded mel-frames/sec, " + ("few-shot fine-tune step (frozen encoder, guided attention; fwd+loss+bwd+allreduce+Adam)"
                                                      if finetune else "the reference's own loop on the drop-in modules (train.py:171-174,188-190: m(**batch), compute_loss, "
                                                      "zero_grad, backward, torch.optim.Adam.step, LambdaLR.step)" if dropin
                                                      else "full training step (fwd+loss+bwd+allreduce+Adam)"),
               "value": round(world * B * T * args.steps / elapsed, 1), "unit": "mel-frames/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
               "data": "synthetic", "final_loss": round(loss, 5),
               "config": {"workload": "%s packed batch per GPU: B=%d S=%d T=%d, default hparams (83.5M params), "
                                      "dropout on, %s%s" % ("LJSpeech-shaped (BASELINE configs[1])" if args.workload == "lj" else
                                                            "38-language byte2speech (BASELINE configs[2])", B, S, T,
                                                            "single speaker/language" if args.workload == "lj" else "572 speakers / 38 languages, different seed per rank",
                                                            ", frozen encoder, guided-attention weight 1.0, batches drawn from a 30-utterance pool" if finetune else ""),
                          "global_batch": world * B, "seq_len": T, "parallelism": "dp%d" % world,
                          "grad_payload": (payload if (world > 1 or force_dp) else None)},
               "device_ms_per_step": round(dev_ms / args.steps, 3), "host_launch_ms_per_step": round(host_s / args.steps * 1e3, 3)}
        valid = float(np.mean([int(b["target_lengths"].sum()) for b in batches]))
        out["valid_frames_per_s"] = round(world * valid * args.steps / elapsed, 1)     # sum(target_lengths): the batch is ~10 % padding
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_FP32_TFLOPS
        ach = step_flops / (ms * 1e-3) / 1e12
        out["roofline_step"] = {"bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s",
                                "frac": round(ach / peak, 4), "flops_per_step": step_flops}
        if extremes:
            out["packer_extremes"] = extremes
    if rank == 0 and world == 1 and not args.no_roofline_pass and not dropin and trainer is not None:
        # dominant kernel (MFMA GEMM): per-launch HIP events on the launch stream, same steps, separate pass
        lib = L.load()
        lib.b2s_prof_enable(1)
        nprof = max(2, min(5, args.steps))
        for _ in range(nprof):
            train_step(batch)
        torch.cuda.synchronize()
        lib.b2s_prof_enable(0)
        res = (C.c_double * 51)()
        L.check(lib.b2s_prof_collect(res, 17))

        def kname(v):               # the kernel names rocprofv3 reports (profiles/*kernel_stats.csv)
            dt, ta, tb, ga = v >> 3, (v >> 2) & 1, (v >> 1) & 1, v & 1
            tf = lambda x: "true" if x else "false"
            if dt:                  # 256-row tile kernel, 128- or 96-column variant chosen per shape (NB = 4 | 3)
                # NB = 3 | 4 (96- / 128-column tiles), MW = 4 | 2 (256- / 128-row tiles), G = 1 | 2 (conv gather)
                # (plain K-contiguous A with a compute-dtype output: the persistent kernel of the same form, gemm_glds256.hip)
                base = "t256::gemm_glds256_kernel<%s, %s, %s, NB, MW>" % (tf(ta), tf(tb), "G" if ga else "0")
                return base if (ta or ga) else base + " + t256::gemm_glds256_persist_kernel<%s, NB>" % tf(tb)
            return "gemm_kernel<float, %s, %s>%s" % (tf(ta), tf(tb), " (conv gather)" if ga else "")
        names = {v: kname(v) for v in range(16)}
        names[16] = "t256::gemm_glds256_grouped_kernel<4>"     # one launch = the weight gradients of one backward stage
        variants = []
        tot_f = tot_ms = 0.0
        for v in range(17):
            f, msv, cnt = res[v * 3], res[v * 3 + 1], res[v * 3 + 2]
            if cnt:
                variants.append({"kernel": names[v], "launches_per_step": cnt / nprof, "avg_us": round(msv * 1e3 / cnt, 2),
                                 "ms_per_step": round(msv / nprof, 3), "tflops": round(f / (msv * 1e-3) / 1e12, 1)})
                tot_f += f
                tot_ms += msv
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_FP32_TFLOPS
        # The second clock: the committed rocprofv3 --kernel-trace --stats summary of this same command (kernel durations proper; the HIP events
        # above sit on the launch stream and include the kernel boundary and whatever shares the CUs).  The DOMINANT kernel is the GEMM family with
        # the most time per step in that summary, and roofline.frac is computed from ITS clock, so that a reader can recompute it from profiles/:
        #     frac = flops_per_step of the family (counted live, below) / (launches per step x average duration in the csv) / peak
        ks_file, ks_steps, ks = load_kernel_stats() if (args.dtype == "bf16" and args.mode == "train" and args.workload == "lj" and
                                                           (B, S, T) == (14, 114, 582)) else (None, 0, {})
        import re
        for d in variants:
            # the csv's names carry the numeric template arguments the variant names abbreviate: NB / MW -> digits, G -> 1 | 2 (conv gather)
            pats = [re.escape(k).replace("NB", r"\d").replace("MW", r"\d").replace(r",\ G,", r",\ [12],") for k in d["kernel"].split(" + ")]
            rows = [(n, v) for n, v in ks.items() if any(re.search(pt, n) for pt in pats)]
            if rows:
                d["rocprof_launches_per_step"] = round(sum(v[0] for _, v in rows), 2)
                d["rocprof_ms_per_step"] = round(sum(v[0] * v[1] for _, v in rows) * 1e-6, 4)
                d["rocprof_avg_us"] = round(d["rocprof_ms_per_step"] * 1e3 / max(d["rocprof_launches_per_step"], 1e-9), 2)
                d["rocprof_tflops"] = round(d["tflops"] * d["ms_per_step"] / d["rocprof_ms_per_step"], 1)      # same FLOPs, the profiler's clock
        have_ks = any("rocprof_ms_per_step" in d for d in variants)
        variants.sort(key=lambda d: -(d.get("rocprof_ms_per_step", 0.0) if have_ks else d["ms_per_step"]))
        dom = variants[0]
        # HBM-side traffic of that kernel from the committed PMC run (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
        # passes, FETCH_SIZE doubled per MI355X_MICROARCH.md; tools/gpu_pmc.sh) -- counters cannot be read from inside this process
        traffic = mfma_util = traffic_source = None
        from bench_decode import load_pmc
        pmc, pmc_commit, pmc_rows = load_pmc("train_bf16_pmc_hbm_traffic_mfma.json")
        if pmc and args.dtype == "bf16" and args.mode == "train" and args.workload == "lj" and (B, S, T) == (14, 114, 582):
            pres = [k.split(", NB")[0].replace(", G", ", ") for k in dom["kernel"].split(" + ")]
            rows = [r for r in pmc_rows if any(r["kernel"].startswith(pre) for pre in pres)]
            traffic_source = "profiles/%s @ commit %s (committed rocprofv3 --pmc run, not measured in this process)" % (pmc, pmc_commit)
            n = sum(r["launches"] for r in rows)
            if n:
                traffic = round(sum(r["launches"] * (r["fetch_MB_per_launch_corrected_x2"] + (r["WRITE_SIZE_KB_per_launch"] or 0) / 1024)
                                    for r in rows) / n * 1e6)
                mu = [r for r in rows if r.get("mfma_util") is not None]
                if mu:          # SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE), same committed PMC run
                    mfma_util = round(sum(r["launches"] * r["mfma_util"] for r in mu) / sum(r["launches"] for r in mu), 4)
        # the same MFMA kernels with the GPU to themselves: the step's forward / dX shapes launched back-to-back through the
        # C-ABI op (no second stream, operands cache-resident after the first launch) -- what the kernel does when it is
        # not sharing CUs and not paying a fused residual epilogue; context for the in-step figure above, not a replacement
        isolated = None
        if args.dtype == "bf16":
            Mt, Me = B * T, B * S
            shapes = [("fwd qkv", Mt, 2304, 768, 0), ("fwd attn-out / q", Mt, 768, 768, 0), ("fwd ffn-in", Mt, 3072, 768, 0),
                      ("fwd ffn-out", Mt, 768, 3072, 0), ("dX qkv", Mt, 768, 2304, 1), ("dX ffn-in", Mt, 768, 3072, 1),
                      ("dX ffn-out", Mt, 3072, 768, 1), ("enc fwd ffn-out", Me, 512, 2048, 0)]
            bufA = torch.randn(Mt * 3072, device=device).to(torch.bfloat16)
            bufB = torch.randn(3072 * 3072, device=device).to(torch.bfloat16)
            bufC = torch.empty(Mt * 3072, device=device, dtype=torch.bfloat16)
            isolated, fl_sum, us_sum = [], 0.0, 0.0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for what, m_, n_, k_, tb in shapes:
                d = L.GemmDesc()
                d.dtype, d.trans_a, d.trans_b, d.M, d.N, d.K = 1, 0, tb, m_, n_, k_
                d.lda, d.ldb, d.ldc, d.c_fp32, d.batch, d.batch_inner, d.alpha = k_, (n_ if tb else k_), n_, 0, 1, 1, 1.0
                call = lambda: L.check(lib.b2s_gemm(C.byref(d), bufA.data_ptr(), bufB.data_ptr(), bufC.data_ptr(), None, None, None,
                                                    None, L.stream()))
                for _ in range(3):
                    call()
                e0.record()
                for _ in range(30):
                    call()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / 30
                fl = 2.0 * m_ * n_ * k_
                isolated.append({"shape": "%s %dx%dx%d" % (what, m_, n_, k_), "us": round(us, 1), "tflops": round(fl / us / 1e6, 1)})
                fl_sum += fl
                us_sum += us
            isolated.append({"shape": "all of the above", "us": round(us_sum, 1), "tflops": round(fl_sum / us_sum / 1e6, 1),
                             "frac_of_peak": round(fl_sum / us_sum / 1e6 / peak, 4)})
            del bufA, bufB, bufC
        ach = dom.get("rocprof_tflops", dom["tflops"])
        out["roofline"] = {"bound": "mfma", "kernel": dom["kernel"], "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                           "frac": round(ach / peak, 4), "traffic": traffic, "traffic_unit": "bytes per launch (HBM side, PMC)", "traffic_source": traffic_source,
                           "mfma_util": mfma_util,
                           "clock": ("rocprofv3 kernel durations, profiles/%s (%d profiled steps)" % (ks_file, ks_steps)) if "rocprof_tflops" in dom
                                    else "HIP events on the launch stream (no committed rocprofv3 summary for this workload)",
                           "flops_per_step": round(dom["tflops"] * 1e12 * dom["ms_per_step"] * 1e-3),
                           "avg_launch_us_rocprof": dom.get("rocprof_avg_us"), "avg_launch_us_events": dom["avg_us"],
                           "achieved_events": dom["tflops"], "frac_events": round(dom["tflops"] / peak, 4),
                           "launches_per_step": dom["launches_per_step"],
                           "all_gemm": {"tflops": round(tot_f / (tot_ms * 1e-3) / 1e12, 1), "ms_per_step": round(tot_ms / nprof, 3)},
                           "variants": variants, "isolated": isolated,
                           "note": "dominant kernel = the GEMM family with the most time per step in the committed rocprofv3 summary; achieved / frac = "
                                   "algorithmic FLOPs of that family per step (2MNK per launch, counted live) / its duration per step on the profiler's "
                                   "clock (launches x average ns of the csv); *_events = the same FLOPs / HIP-event durations on the kernel's launch stream, separate "
                                   "instrumented pass of the same steps; the weight-gradient GEMMs (grouped kernel, <true, true, *>) run on a "
                                   "second stream concurrently with the rest of the backward pass, so their event durations -- and those "
                                   "of the main-stream kernels they overlap -- include time spent sharing the CUs; <true, true, *> "
                                   "durations also include the split-K slab reduction that belongs to the launch"}
    if (world > 1 or force_dp) and args.mode == "train" and not args.no_extras:
        # One invocation decides the data-parallel defaults: after the headline leg (bucketed all-reduce, the wire named in config.grad_payload) the same
        # step is timed with the other gradient wire, with the sharded optimizer (reduce-scatter + all-gather of the updated parameters) on both
        # wires, with the whole exchange path on but the wire removed (every collective completes at once: what the rank pays for hooks, packs,
        # waits and the tile policy), and with no process group at all (the single-GPU step on this node's GPUs).  Every leg: a fresh model from
        # the same initial state, the same warm-up and step counts, barrier + synchronize on both sides, MAX over ranks.
        class _Done(object):
            def wait(self, *a, **k):
                return True

        class _NoWire(object):
            def __init__(self, world_, rank_):
                self.w, self.r = world_, rank_

            def get_world_size(self, group=None):
                return self.w

            def get_rank(self, group=None):
                return self.r

            def broadcast(self, *a, **k):
                return _Done()

            all_reduce = reduce_scatter_tensor = all_gather_into_tensor = broadcast

        head_ms = elapsed / args.steps * 1e3
        legs = [{"leg": "headline", "dp_mode": "allreduce", "wire": payload, "ms_per_step": round(head_ms, 3)}]
        trainer.close()
        del trainer, train_step
        torch.cuda.empty_cache()
        for leg, mode, wire in (("allreduce / fp32 wire", "allreduce", "fp32"), ("allreduce / bf16 wire", "allreduce", "bf16"), ("rs_ag / fp32 wire", "rs_ag", "fp32"),
                                ("rs_ag / bf16 wire", "rs_ag", "bf16"), ("exchange path on, wire removed", "allreduce", payload), ("no process group (single-GPU step)", None, None)):
            if (mode, wire) == ("allreduce", payload) and "wire removed" not in leg:
                continue                                   # = the headline leg
            m2 = t2 = None
            try:
                m2 = Tacotron(hp)
                m2.load_state_dict(init_state)
                m2 = m2.to(device).train()
                if mode is None:
                    t2 = HipTrainer(m2, hp, dist=False)
                elif "wire removed" in leg:
                    t2 = HipTrainer(m2, hp, grad_payload=wire, dp_mode=mode, dist=_NoWire(world, rank))
                else:
                    t2 = HipTrainer(m2, hp, grad_payload=wire, dp_mode=mode)
                v2, el2, _, _ = timed(batches, args.steps, args.warmup, step_fn=t2.train_step)
                legs.append({"leg": leg, "dp_mode": mode, "wire": wire, "ms_per_step": round(el2 / args.steps * 1e3, 3), "final_loss": round(float(v2[0]), 5)})
            except Exception as e:                          # a leg that cannot run (e.g. rs_ag at a world size its buckets do not divide) is reported, not fatal
                legs.append({"leg": leg, "dp_mode": mode, "wire": wire, "error": "%s: %s" % (type(e).__name__, str(e)[:200])})
                sync()
            if t2 is not None:
                t2.close()
            del t2, m2
            torch.cuda.empty_cache()
        if rank == 0:
            by = {l["leg"]: l.get("ms_per_step") for l in legs}
            local_ms, nowire_ms = by.get("no process group (single-GPU step)"), by.get("exchange path on, wire removed")
            out["dp_legs"] = legs
            out["local_ms_per_step"] = local_ms
            # weak scaling: value(N) / (N x value(1)) = t(1) / t(N) with t(1) = the no-process-group leg of this same run
            out["scaling_efficiency"] = round(local_ms / head_ms, 4) if local_ms else None
            out["exposed_comm_ms"] = round(head_ms - nowire_ms, 3) if nowire_ms else None
            out["exchange_path_overhead_ms"] = round(nowire_ms - local_ms, 3) if (nowire_ms and local_ms) else None
            best = min((l for l in legs if l.get("ms_per_step") and l["dp_mode"] and "wire removed" not in l["leg"]), key=lambda l: l["ms_per_step"])
            out["fastest_dp_leg"] = {"leg": best["leg"], "ms_per_step": best["ms_per_step"]}
        trainer = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_train()
    if world > 1 or force_dp:
        out_n = torch.distributed.get_world_size()
        if rank == 0:
            out["rccl_ranks"] = out_n                  # what the RCCL group itself reports (== n_gpus)
    if rank == 0 and world == 1 and not force_dp and args.mode == "train" and args.dtype == "bf16" and args.workload == "lj" and not args.no_extras:
        # the other BASELINE.json configs on the same JSON line (each in its own process: the hparams object is global):
        #   decode     configs[3]  64 utterances x 1000 frames, hipGraph-captured KV-cached loop
        #   finetune   configs[4]  frozen encoder + guided-attention loss, batches from a 30-utterance pool
        #   fp32_mode  the parity-proven arithmetic (exact-fp32 MFMA) on the headline batch
        del trainer, model
        torch.cuda.empty_cache()
        out["decode"] = _sub_bench(["--mode", "decode", "--steps", "10", "--warmup", "1"] + (["--no-cpu-baseline"] if args.no_cpu_baseline else []))
        out["finetune"] = _sub_bench(["--mode", "finetune", "--steps", str(args.steps), "--warmup", str(args.warmup), "--no-cpu-baseline",
                                      "--no-roofline-pass"])
        out["fp32_mode"] = _sub_bench(["--dtype", "fp32", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-roofline-pass"])
        # the headline step with the data-parallel exchange path switched on (1-rank RCCL group: stage hooks, buckets, bf16 pack, all-reduce
        # calls, waits, Adam reading the wire buffer -- everything but the wire itself): what one rank of an N-GPU run pays on top
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            dp_port = sk.getsockname()[1]
        dp_env = {"B2S_FORCE_DP": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(dp_port), "RANK": "0", "WORLD_SIZE": "1"}
        dp = _sub_bench(["--steps", str(args.steps), "--warmup", str(args.warmup), "--no-cpu-baseline", "--no-roofline-pass", "--no-extras"], env=dp_env)
        out["dp_path_ms_per_step"] = dp.get("ms_per_step", dp.get("error"))
        # BASELINE configs[2] on one GPU: the per-rank workload of the N-GPU run (what `--gpus N` times on every rank), exchange path on
        out["c3"] = _sub_bench(["--workload", "c3", "--steps", str(args.steps), "--warmup", str(args.warmup), "--no-cpu-baseline", "--no-roofline-pass",
                                "--no-extras"], env=dp_env)
        # the reference's own loop (train.py:171-174,188-190) on the drop-in modules, same batch, same timing discipline: what train.py gets UNEDITED
        out["dropin_loop"] = _sub_bench(["--mode", "dropin", "--steps", str(args.steps), "--warmup", str(args.warmup), "--no-cpu-baseline", "--no-roofline-pass",
                                         "--no-extras"])
    if world > 1 or force_dp:
        torch.distributed.destroy_process_group()
    if rank == 0:
        C.CDLL(None).fflush(None)          # RCCL prints its library path through C stdio: keep the JSON line last
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
