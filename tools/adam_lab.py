"""Optimizer update alone (default hparams, bf16): microseconds per b2s_adam_step call."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "few-shot-transformer-tts_amd"))
from hyperparams import hparams as hp
from transformer.tacotron import Tacotron, initialize_variables
from b2s_hip.trainer import HipTrainer
from b2s_hip import lib as L
hp.parse("compute_dtype=bf16")
torch.manual_seed(0)
m = Tacotron(hp); initialize_variables(m); m = m.to("cuda").train()
tr = HipTrainer(m, hp)
eng = tr.eng
eng._gflat.normal_(0, 1e-3)
args = (1e-3, 1, 0.9, 0.999, hp.adam_eps, hp.reg_weight, 1.0)
def step(i):
    L.check(tr.lib.b2s_adam_step(eng.handle, args[0], i + 1, *args[2:], L.stream()))
for i in range(5): step(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 50
e0.record()
for i in range(N): step(5 + i)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / N * 1e3
n = eng._gflat.numel()
print("adam %s: %.1f us per step, %d params, %.2f TB/s (30 B/param)" % ("v1" if os.environ.get("B2S_ADAM_V1") else "v2", us, n, n * 30 / us / 1e6))
