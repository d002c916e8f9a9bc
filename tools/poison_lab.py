#!/usr/bin/env python3
"""Lab: does any kernel of the training step read memory it was never given?  The caching allocator's free blocks are filled with NaN bit patterns before
every step (workspaces come from torch.empty), then the fused trainer runs: a NaN in any loss, gradient or parameter is a stale read.
usage: python tools/poison_lab.py [ragged|padded] [fp32|bf16]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "few-shot-transformer-tts_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from oracle import synth, make_config, TINY96
from test_gpu_model import build, dev_batch
from test_gpu_dropout_parity import with_dropout

mode = sys.argv[1] if len(sys.argv) > 1 else "ragged"
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp32"
from b2s_hip.trainer import HipTrainer
over = with_dropout(TINY96)
cfg0 = make_config(over)
st = synth.synthetic_state(cfg0, 1234)
for case, (B, S, T, il, tl) in enumerate(((3, 11, 23, [11, 7, 4], [23, 15, 9]), (5, 37, 200, [37, 30, 11, 37, 5], [200, 129, 64, 63, 1]))):
    nb = synth.synthetic_batch(cfg0, B=B, S=S, T=T, seed=7, in_lens=il, tgt_lens=tl)
    ref = None
    for poison in (0, 1, 1):
        m, cfg, _, hp = build(over, compute_dtype=dtype, state_edit=lambda s: s.update(st))
        m.train()
        tr = HipTrainer(m, hp, dist=False)
        b = dev_batch(nb)
        if mode == "ragged":
            b["target_lengths_host"] = [int(x) for x in nb["target_lengths"]]
        if poison:
            junk = [torch.full((64 << 20,), float("nan"), device="cuda") for _ in range(8)]      # 2 GB of NaN, returned to the allocator's free list
            del junk
        grads = []
        tr.grad_probe = lambda flat, wire: grads.append(flat.detach().clone())
        v = tr.train_step(b)
        torch.cuda.synchronize()
        g = grads[0]
        bad = [n for n, (o, c) in tr.eng.param_offsets.items() if not torch.isfinite(g[o:o + c]).all()]
        print("case %d %s %s poison=%d: losses finite %s, non-finite gradient tensors %d %s" % (case, mode, dtype, poison, bool(torch.isfinite(v).all()), len(bad), bad[:4]))
        if ref is None:
            ref = g
        else:
            worst = max(((float((g[o:o + c] - ref[o:o + c]).norm()) / max(float(ref[o:o + c].norm()), 1e-12), n) for n, (o, c) in tr.eng.param_offsets.items()))
            print("        worst gradient difference to the unpoisoned run: %.2e (%s)" % worst)
