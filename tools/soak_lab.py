"""Soak: N training steps over 4 alternating batches; prints the loss every 50 steps and a checksum of the parameters at the end.  Run with the lab library and
B2S_LAB_GEMM_PERSIST = 0 / 1: the persistent GEMM kernel is bit-identical to the plain one, so the two runs must print the same numbers (the ticket
counters wrap their 1024-set pool several times in 400 steps)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "few-shot-transformer-tts_amd"))
from hyperparams import hparams as hp
from transformer.tacotron import Tacotron, initialize_variables
from b2s_hip.trainer import HipTrainer
from benchdata import synthetic_batch
hp.parse("compute_dtype=bf16")
torch.manual_seed(0)
m = Tacotron(hp); initialize_variables(m); m = m.to("cuda").train()
tr = HipTrainer(m, hp)
bs = []
for i, (B, S, T) in enumerate(((14, 114, 582), (9, 158, 808), (32, 50, 250), (14, 114, 582))):
    nb = synthetic_batch(hp, B, S, T, seed=i, n_spk=1, n_lang=1)
    bs.append({k: (torch.from_numpy(np.asarray(v)).cuda() if not isinstance(v, list) else v) for k, v in nb.items()})
    bs[-1]["target_lengths_host"] = [int(x) for x in np.asarray(nb["target_lengths"])]      # ragged decoder rows, as bench.py
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for i in range(N):
    v = tr.train_step(bs[i % 4])
    if (i + 1) % 50 == 0:
        print("step %4d loss %.6f" % (i + 1, float(v[0])), flush=True)
torch.cuda.synchronize()
cs = 0.0
for n_, p in m.named_parameters():
    cs += float(p.detach().double().abs().sum())
print("parameter checksum %.9e" % cs)
