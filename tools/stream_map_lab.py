"""How sensitive is the step to WHICH stream object the encoder stream is?  K dummy torch streams are taken from torch's pool before the trainer takes its
encoder stream (a DeviceStager, a user's copy stream, ... do the same in a real run); LAB_HIPSTREAMS=n also creates n raw HIP streams first (shifts the engine's second stream)."""
import os, sys, time, gc, ctypes
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
torch.cuda.init(); torch.zeros(1, device="cuda")
K = int(os.environ.get("LAB_K", "0")); NH = int(os.environ.get("LAB_HIPSTREAMS", "0"))
hip = ctypes.CDLL("libamdhip64.so")
raw = []
for _ in range(NH):
    s = ctypes.c_void_p(); assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0; raw.append(s)
dummies = [torch.cuda.Stream() for _ in range(K)]
nb = synthetic_batch(hp, 14, 114, 582, seed=0, n_spk=1, n_lang=1)
batch = {k: (torch.from_numpy(np.asarray(v)).cuda() if not isinstance(v, list) else v) for k, v in nb.items()}
batch["target_lengths_host"] = [int(x) for x in np.asarray(nb["target_lengths"])]
m = Tacotron(hp); initialize_variables(m); m = m.to("cuda").train()
tr = HipTrainer(m, hp, dist=False)
for _ in range(8): tr.train_step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): tr.train_step(batch)
torch.cuda.synchronize()
print("K=%d dummy torch streams, %d raw HIP streams first: %.3f ms per step" % (K, NH, (time.perf_counter() - t0) / 30 * 1e3), flush=True)
