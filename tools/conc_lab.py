#!/usr/bin/env python3
"""Experiment: do two independent half-batch training chains on two HIP streams of ONE process overlap each other's per-kernel
fixed costs?  Two models / trainers (B=7 each), one Python thread per chain (ctypes releases the GIL in the C calls); compared
with one trainer at B=14 and one at B=7.  usage: python tools/conc_lab.py"""
import os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "few-shot-transformer-tts_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from hyperparams import hparams as hp
from transformer.tacotron import Tacotron, initialize_variables
from b2s_hip.trainer import HipTrainer
from oracle import synth, make_config

dev = torch.device("cuda", 0)
hp.parse("compute_dtype=bf16")
cfg = make_config("")


def mk(B, seed):
    torch.manual_seed(0)
    m = Tacotron(hp); initialize_variables(m); m = m.to(dev).train()
    nb = synth.synthetic_batch(cfg, B, 114, 582, seed=seed, n_spk=1, n_lang=1)
    b = {k: (torch.from_numpy(np.asarray(v)).to(dev) if not isinstance(v, list) else v) for k, v in nb.items()}
    return HipTrainer(m, hp), b


def timed(chains, steps=40, warm=8):
    streams = [torch.cuda.Stream() for _ in chains]
    def work(i, n):
        tr, b = chains[i]
        with torch.cuda.stream(streams[i]):
            for _ in range(n):
                tr.train_step(b)
    def run(n):
        th = [threading.Thread(target=work, args=(i, n)) for i in range(len(chains))]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
    run(warm)
    t0 = time.perf_counter(); run(steps); dt = time.perf_counter() - t0
    return dt / steps * 1e3


one14 = [mk(14, 0)]
print("one chain  B=14: %.3f ms/step" % timed(one14)); del one14
one7 = [mk(7, 0)]
print("one chain  B=7 : %.3f ms/step" % timed(one7)); del one7
two7 = [mk(7, 0), mk(7, 1)]
print("two chains B=7 : %.3f ms per step pair (14 utterances)" % timed(two7))
