"""Where the main stream is at the trainer's call boundaries (unprofiled HIP events, 20 steps): forward / backward phases of the step."""
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
S = int(os.environ.get("LAB_S", "114")); nsl = (572, 38) if S > 128 else (1, 1)
nb = synthetic_batch(hp, 14, S, 582, seed=0, n_spk=nsl[0], n_lang=nsl[1])
batch = {k: (torch.from_numpy(np.asarray(v)).cuda() if not isinstance(v, list) else v) for k, v in nb.items()}
batch["target_lengths_host"] = [int(x) for x in np.asarray(nb["target_lengths"])]      # ragged decoder rows, as bench.py
eng = tr.eng
ev = {}
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(torch.cuda.current_stream()); ev.setdefault(name, []).append(e)
def wrap(name):
    f = getattr(eng, name)
    def g(*a, **k):
        mark(name + ":in"); r = f(*a, **k); mark(name + ":out"); return r
    setattr(eng, name, g)
names = ["encoder_forward", "decoder_forward", "postnet_forward", "loss_backward", "postnet_backward", "decoder_backward", "encoder_backward"]
for n in names: wrap(n)
_adam = tr.lib.b2s_adam_step_groups
def adam_groups(*a):
    k = "adam_groups_%d" % a[8]
    mark(k + ":in"); r = _adam(*a); mark(k + ":out"); return r
class _Lib(object):
    def __init__(self, l): self._l = l
    def __getattr__(self, n): return adam_groups if n == "b2s_adam_step_groups" else getattr(self._l, n)
tr.lib = _Lib(tr.lib)
names += ["adam_groups_6", "adam_groups_1"]
for _ in range(8): tr.train_step(batch)
torch.cuda.synchronize(); ev.clear()
N = 20
main = torch.cuda.current_stream()
for _ in range(N):
    e = torch.cuda.Event(enable_timing=True); e.record(main); ev.setdefault("step:in", []).append(e)
    tr.train_step(batch)
    e = torch.cuda.Event(enable_timing=True); e.record(main); ev.setdefault("step:out", []).append(e)
torch.cuda.synchronize()
def avg(a, b): return sum(x.elapsed_time(y) for x, y in zip(ev[a], ev[b])) / N * 1e3
print("step %.0f us" % avg("step:in", "step:out"))
order = ["step:in"] + [n + s for n in names for s in (":in", ":out")] + ["step:out"]
prev = "step:in"
for k in order[1:]:
    if k not in ev: continue
    print("  %-26s +%7.0f us   (at %7.0f)" % (k, avg(prev, k), avg("step:in", k)))
    prev = k
