"""Timing lab (lab build only: B2S_LIB_PATH=tools/bin/libb2s_hip_lab.so): what would the step gain if the decoder / postnet optimizer update did not
wait for the held weight-gradient groups of the decoder's first layer (engine.hip: dw_hold_from) -- the mark recorded in FRONT of them
(B2S_LAB_EARLY_MARK=1) and the update issued right behind the decoder backward, before the encoder backward is enqueued (LAB_EARLY=1).
The held layer's update then races with its own gradients: the numbers are an upper bound of a correct split update, the results are not used.
usage: [B2S_LAB_EARLY_MARK=1] LAB_EARLY=0|1 python tools/early_adam_lab.py"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "few-shot-transformer-tts_amd"))
from hyperparams import hparams as hp
from transformer.tacotron import Tacotron, initialize_variables
from b2s_hip.trainer import HipTrainer
from b2s_hip import lib as L
from benchdata import synthetic_batch
hp.parse("compute_dtype=bf16")
torch.manual_seed(0)
m = Tacotron(hp); initialize_variables(m); m = m.to("cuda").train()
tr = HipTrainer(m, hp)
nb = synthetic_batch(hp, 14, 114, 582, seed=0, n_spk=1, n_lang=1)
batch = {k: (torch.from_numpy(np.asarray(v)).cuda() if not isinstance(v, list) else v) for k, v in nb.items()}
batch["target_lengths_host"] = [int(x) for x in np.asarray(nb["target_lengths"])]
early = os.environ.get("LAB_EARLY", "0") == "1"
_lib = tr.lib
class _Lib(object):
    def __getattr__(self, n):
        f = getattr(_lib, n)
        if not early: return f
        if n == "b2s_model_mark_grads_ready":
            def g(h):
                r = f(h)
                if r == 0: r = _lib.b2s_adam_step_groups(h, *tr._last_adam, 2 | 4, 1, L.stream())
                return r
            return g
        if n == "b2s_adam_step_groups":
            def g(h, *a):
                return 0 if a[7] == (2 | 4) else f(h, *a)
            return g
        return f
tr.lib = _Lib()
for _ in range(8): tr.train_step(batch)
torch.cuda.synchronize()
for rnd in range(3):
    t0 = time.perf_counter()
    for _ in range(40): tr.train_step(batch)
    torch.cuda.synchronize()
    print("LAB_EARLY=%d B2S_LAB_EARLY_MARK=%s: %.3f ms per step" % (early, os.environ.get("B2S_LAB_EARLY_MARK", "-"), (time.perf_counter() - t0) / 40 * 1e3), flush=True)
