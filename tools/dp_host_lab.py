"""Where does the HOST spend its time in a data-parallel step (1-rank RCCL group)?  perf_counter around the trainer's calls, no device syncs."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "few-shot-transformer-tts_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
from hyperparams import hparams as hp
from transformer.tacotron import Tacotron, initialize_variables
from b2s_hip.trainer import HipTrainer
from benchdata import synthetic_batch
hp.parse("compute_dtype=bf16")
if os.environ.get("B2S_FORCE_DP"):
    torch.distributed.init_process_group("nccl", init_method="env://", device_id=torch.device("cuda", 0))
torch.manual_seed(0)
m = Tacotron(hp); initialize_variables(m); m = m.to("cuda").train()
tr = HipTrainer(m, hp)
nb = synthetic_batch(hp, 14, 114, 582, seed=0, n_spk=1, n_lang=1)
batch = {k: (torch.from_numpy(np.asarray(v)).cuda() if not isinstance(v, list) else v) for k, v in nb.items()}
acc = {}
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc[label or name] = acc.get(label or name, 0.0) + time.perf_counter() - t0; return r
    setattr(obj, name, g)
for n in ("encoder_forward", "decoder_forward", "postnet_forward", "loss_forward", "loss_backward", "postnet_backward", "decoder_backward", "encoder_backward"):
    wrap(tr.eng, n)
if tr.bucketer is not None:
    wrap(tr.bucketer, "finish"); wrap(tr.bucketer, "stage_done"); wrap(tr.bucketer, "_launch")
for _ in range(8): tr.train_step(batch)
torch.cuda.synchronize(); acc.clear()
N = 20
t0 = time.perf_counter()
for _ in range(N): tr.train_step(batch)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host enqueue %.2f ms per step, wall %.2f ms per step" % (t_host / N * 1e3, t_all / N * 1e3))
print("  ".join("%s %.2f" % (k, v / N * 1e3) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])))
