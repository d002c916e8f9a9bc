#!/usr/bin/env python3
"""How far ahead of the GPU does the host run inside a training step?  A sleep of d ms is inserted on the host before the encoder
backward (near the end of the step): the step gets slower by (d - lead).  usage: python tools/host_lead_lab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "few-shot-transformer-tts_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from hyperparams import hparams as hp
from transformer.tacotron import Tacotron, initialize_variables
from b2s_hip.trainer import HipTrainer
from oracle import synth, make_config

hp.parse("compute_dtype=bf16")
torch.manual_seed(0)
m = Tacotron(hp); initialize_variables(m); m = m.to("cuda").train()
tr = HipTrainer(m, hp)
nb = synth.synthetic_batch(make_config(""), 14, 114, 582, seed=0, n_spk=1, n_lang=1)
batch = {k: (torch.from_numpy(np.asarray(v)).to("cuda") if not isinstance(v, list) else v) for k, v in nb.items()}
eng = tr.eng
orig = eng.encoder_backward
delay = [0.0]
def slow(*a, **k):
    if delay[0] > 0:
        t = time.perf_counter()
        while time.perf_counter() - t < delay[0]:
            pass
    return orig(*a, **k)
eng.encoder_backward = slow
for d in (0.0, 0.0005, 0.001, 0.002, 0.004):
    delay[0] = d
    for _ in range(5): tr.train_step(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): tr.train_step(batch)
    torch.cuda.synchronize()
    print("host stall %.1f ms before the encoder backward: %.3f ms per step" % (d * 1e3, (time.perf_counter() - t0) / 30 * 1e3))
