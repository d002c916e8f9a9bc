"""Lab: does the step get shorter when the trainer's stream starts its (memory-independent) decoder work LATER, leaving the first part of the encoder
forward the chip to itself?  A one-workgroup sleep kernel (tools/cu_hold.hip) in front of the decoder forward, 0 .. 300 us."""
import ctypes, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "few-shot-transformer-tts_amd"))
from hyperparams import hparams as hp
from transformer.tacotron import Tacotron, initialize_variables
from b2s_hip.trainer import HipTrainer
from benchdata import synthetic_batch
hold = ctypes.CDLL(os.path.join(ROOT, "tools", "bin", "libcuhold.so"))
hold.cu_hold.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
hp.parse("compute_dtype=bf16")
torch.manual_seed(0)
m = Tacotron(hp); initialize_variables(m); m = m.to("cuda").train()
tr = HipTrainer(m, hp)
nb = synthetic_batch(hp, 14, 114, 582, seed=0, n_spk=1, n_lang=1)
batch = {k: (torch.from_numpy(np.asarray(v)).cuda() if not isinstance(v, list) else v) for k, v in nb.items()}
delay = [0.0]
orig = tr.eng.decoder_forward
def dec_f(*a, **k):
    if delay[0] > 0:
        assert hold.cu_hold(1, delay[0], torch.cuda.current_stream().cuda_stream) == 0
    return orig(*a, **k)
tr.eng.decoder_forward = dec_f
for _ in range(8): tr.train_step(batch)
torch.cuda.synchronize()
for rep in range(2):
    for d in (0.0, 0.05, 0.1, 0.15, 0.2, 0.25, 0.3):
        delay[0] = d
        for _ in range(3): tr.train_step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30): tr.train_step(batch)
        torch.cuda.synchronize()
        print("delay %3.0f us: %.3f ms per step" % (d * 1e3, (time.perf_counter() - t0) / 30 * 1e3), flush=True)
