#!/usr/bin/env python3
"""Convergence-level evidence for the benchmarked arithmetic: the full-size model trained for N steps at the LJSpeech shape (B = 14, S = 114,
T = 582, dropout on at the reference rates, TF-style init) by the bf16 engine and by the fp32 engine IN LOCKSTEP -- same initial parameters,
same batches (8 synthetic ones, cycled), same dropout seeds (so the same masks: the engine's masks are a function of seed, op and index only) --
plus a third arm, the bf16 engine with the fp32 residual gradient (B2S_DX_BF16=0), in a child process (the switch is read once per process).

Recorded: every arm's loss curve, the mean loss over the last 50 steps, and every 100 steps the distance between the bf16 and the fp32 parameters,
relative to how far the fp32 run has moved from the initial point (|p_bf16 - p_fp32| / |p_fp32 - p_0|) and relative to the parameters themselves,
overall / per segment / worst tensor.  tests/test_gpu_trajectory.py gates the curves (2 % over the last 50 steps) and commits nothing; the
numbers of record are profiles/rNN_bf16_trajectory.json, written by `python tools/trajectory.py --out profiles/r06_bf16_trajectory.json`.

Reference: the loop of train.py:165-191 (the fused trainer computes the same step, tests/test_gpu_trainer*.py).
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "few-shot-transformer-tts_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def make_batches(cfg, device, n=8, B=14, S=114, T=582):
    from benchdata import synthetic_batch
    out = []
    for seed in range(n):
        nb = synthetic_batch(cfg, B, S, T, seed=seed, n_spk=1, n_lang=1)
        d = {k: (torch.from_numpy(np.asarray(v)).to(device) if not isinstance(v, list) else v) for k, v in nb.items()}
        d["target_lengths_host"] = [int(x) for x in np.asarray(nb["target_lengths"])]        # (ragged decoder rows, as bench.py)
        out.append(d)
    return out


def build(dtype, init_state, device):
    """A full-size model + fused trainer computing in `dtype`; the engine is created before the global hparams object is touched again."""
    import hyperparams
    from hyperparams import hparams as hp
    from transformer.tacotron import Tacotron, initialize_variables
    from b2s_hip.trainer import HipTrainer
    hp.override_from_dict(hyperparams.DEFAULTS)
    hp.parse("compute_dtype=%s" % dtype)
    m = Tacotron(hp)
    if init_state is None:
        torch.manual_seed(0)
        initialize_variables(m)
        init_state = {k: v.detach().clone() for k, v in m.state_dict().items()}
    else:
        m.load_state_dict(init_state)
    m = m.to(device).train()
    return m, HipTrainer(m, hp, dist=False), init_state, hp


def divergence(ma, mb, init_state):
    """bf16 run (ma) against fp32 run (mb): distances over all parameters, per segment and the worst tensor (by movement-relative distance)."""
    acc = {}
    worst = (0.0, None)
    for (n, a), (_, b) in zip(ma.named_parameters(), mb.named_parameters()):
        a, b = a.detach().double(), b.detach().double()
        p0 = init_state[n].to(b.device).double()
        d2, mv2, n2 = float(((a - b) ** 2).sum()), float(((b - p0) ** 2).sum()), float((b ** 2).sum())
        for key in ("all", n.split(".")[0]):
            t = acc.setdefault(key, [0.0, 0.0, 0.0])
            t[0] += d2; t[1] += mv2; t[2] += n2
        if b.numel() >= 1024 and mv2 > 0 and (d2 / mv2) ** 0.5 > worst[0]:
            worst = ((d2 / mv2) ** 0.5, n)
    out = {k: {"dist_over_movement": round((v[0] / max(v[1], 1e-30)) ** 0.5, 5), "dist_over_norm": round((v[0] / max(v[2], 1e-30)) ** 0.5, 6)} for k, v in acc.items()}
    out["worst_tensor"] = {"name": worst[1], "dist_over_movement": round(worst[0], 5)}
    return out


def run_single(dtype, steps, device):
    m, tr, _, hp = build(dtype, None, device)
    batches = make_batches(hp, device)
    losses = []
    for i in range(steps):
        losses.append(float(tr.train_step(batches[i % len(batches)])[0]))
    return losses


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--every", type=int, default=100)
    ap.add_argument("--out", default=None)
    ap.add_argument("--single", default=None, help="(child mode) run ONE arm of this dtype and print its loss curve as JSON")
    ap.add_argument("--no-dx32-arm", action="store_true")
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    if args.single:
        print(json.dumps({"losses": run_single(args.single, args.steps, device)}))
        return
    ma, ta, init_state, hp = build("bf16", None, device)
    mb, tb, _, _ = build("fp32", init_state, device)
    batches = make_batches(hp, device)
    la, lb, div = [], [], {}
    for i in range(args.steps):
        b = batches[i % len(batches)]
        la.append(float(ta.train_step(b)[0]))
        lb.append(float(tb.train_step(b)[0]))
        if (i + 1) % args.every == 0 or i + 1 == args.steps:
            torch.cuda.synchronize()
            div[str(i + 1)] = divergence(ma, mb, init_state)
    tail = min(50, args.steps)
    res = {"steps": args.steps, "shape": {"B": 14, "S": 114, "T": 582}, "batches_cycled": len(batches), "dropout": "reference rates, identical masks in every arm",
           "arms": {"bf16": {"losses": [round(x, 5) for x in la], "mean_last_%d" % tail: round(float(np.mean(la[-tail:])), 5)},
                    "fp32": {"losses": [round(x, 5) for x in lb], "mean_last_%d" % tail: round(float(np.mean(lb[-tail:])), 5)}},
           "bf16_vs_fp32_parameters": div}
    res["bf16_vs_fp32_loss_rel_last_%d" % tail] = round(abs(np.mean(la[-tail:]) - np.mean(lb[-tail:])) / abs(np.mean(lb[-tail:])), 5)
    res["bf16_vs_fp32_loss_rel_max_pointwise"] = round(float(np.max(np.abs(np.array(la) - np.array(lb)) / np.abs(np.array(lb)))), 5)
    if not args.no_dx32_arm:
        del ma, mb, ta, tb
        torch.cuda.empty_cache()
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--single", "bf16", "--steps", str(args.steps)], capture_output=True, text=True,
                           env=dict(os.environ, B2S_DX_BF16="0"), timeout=1800)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode == 0 and lines:
            lc = json.loads(lines[-1])["losses"]
            res["arms"]["bf16_dx_fp32 (B2S_DX_BF16=0)"] = {"losses": [round(x, 5) for x in lc], "mean_last_%d" % tail: round(float(np.mean(lc[-tail:])), 5)}
            res["bf16_dx_fp32_vs_fp32_loss_rel_last_%d" % tail] = round(abs(np.mean(lc[-tail:]) - np.mean(lb[-tail:])) / abs(np.mean(lb[-tail:])), 5)
        else:
            res["arms"]["bf16_dx_fp32 (B2S_DX_BF16=0)"] = {"error": (r.stderr or r.stdout)[-400:]}
    txt = json.dumps(res)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            f.write(txt + "\n")
    brief = {k: v for k, v in res.items() if k != "arms"}
    brief["arms"] = {k: {kk: vv for kk, vv in v.items() if kk != "losses"} for k, v in res["arms"].items()}
    brief["loss_first_last"] = {k: [v["losses"][0], v["losses"][-1]] for k, v in res["arms"].items() if "losses" in v}
    print(json.dumps(brief, indent=1))


if __name__ == "__main__":
    main()
