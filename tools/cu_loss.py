"""Training step with n CUs held by another kernel (tools/cu_hold.hip): what the step pays for every CU a communication library's
channel kernels occupy while it runs.  Prints one row per n.
usage: cu_loss.py [step|bwd]     step: the CUs are held for the whole step; bwd (default): from the start of the backward pass for 4.5 ms
(where a data-parallel run's bucketed all-reduces are in flight)"""
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
batch["target_lengths_host"] = [int(x) for x in np.asarray(nb["target_lengths"])]      # ragged decoder rows, as bench.py
for _ in range(8): tr.train_step(batch)
torch.cuda.synchronize()
side = torch.cuda.Stream()
STEPS = 20
base = None
mode = sys.argv[1] if len(sys.argv) > 1 else "bwd"
held = [0]
orig_loss_backward = tr.eng.loss_backward
def loss_backward(*a, **k):                                         # first engine call of the backward pass
    if mode == "bwd" and held[0] > 0:
        ev = torch.cuda.Event(); ev.record()
        side.wait_event(ev)
        assert hold.cu_hold(held[0], 4.5, side.cuda_stream) == 0
    return orig_loss_backward(*a, **k)
tr.eng.loss_backward = loss_backward
print("mode: %s, B2S_GEMM256_NB=%s" % (mode, os.environ.get("B2S_GEMM256_NB", "auto")))
for n in [0, 8, 16, 32, 64, 0]:
    torch.cuda.synchronize()
    held[0] = n
    if mode == "step":
        assert hold.cu_hold(n, 400.0, side.cuda_stream) == 0        # 400 ms: outlives the 20 timed steps
        time.sleep(0.02)                                            # let its workgroups settle before the step's kernels arrive
    t0 = time.perf_counter()
    for _ in range(STEPS): tr.train_step(batch)
    torch.cuda.current_stream().synchronize()
    ms = (time.perf_counter() - t0) / STEPS * 1e3
    torch.cuda.synchronize()
    base = base or ms
    print("held CUs %3d: %.3f ms per step (%+.1f %%; CUs lost %.1f %%)" % (n, ms, (ms / base - 1) * 100, n / 256 * 100), flush=True)
