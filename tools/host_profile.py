#!/usr/bin/env python3
"""Where does the host spend its time in one training step?  cProfile over the launch loop (no device sync inside)."""
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "few-shot-transformer-tts_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hyperparams import hparams as hp
from transformer.tacotron import Tacotron, initialize_variables
from b2s_hip.trainer import HipTrainer
import numpy as np
from oracle import synth, make_config


def make_batch(cfg, B, S, T, seed, device):
    nb = synth.synthetic_batch(cfg, B, S, T, seed=seed, n_spk=1, n_lang=1)
    return {k: (torch.from_numpy(np.asarray(v)).to(device) if not isinstance(v, list) else v) for k, v in nb.items()}


hp.parse("compute_dtype=bf16")
torch.manual_seed(0)
model = Tacotron(hp)
initialize_variables(model)
model = model.to("cuda").train()
tr = HipTrainer(model, hp)
batch = make_batch(make_config(""), 14, 114, 582, seed=0, device="cuda")
for _ in range(5):
    tr.train_step(batch)
torch.cuda.synchronize()
# one step at a time with an empty queue: pure host cost of enqueueing a step (no back-pressure from the GPU)
single = []
for _ in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_step(batch)
    single.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
print("host time to enqueue one step on an idle queue (ms):", " ".join("%.2f" % x for x in single))
n = 20
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(n):
    tr.train_step(batch)
pr.disable()
host = time.perf_counter() - t0
torch.cuda.synchronize()
print("host launch loop: %.3f ms/step, with sync %.3f ms/step" % (host / n * 1e3, (time.perf_counter() - t0) / n * 1e3))
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
