"""Does a second / third HipTrainer built in the SAME process run the step as fast as the first (bench.py --gpus N times its legs that way)?
Builds model + trainer, times 30 steps, closes, repeats.  LAB_REUSE=1: every trainer is handed the first one's encoder stream."""
import os, sys, time, gc
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
nb = synthetic_batch(hp, 14, 114, 582, seed=0, n_spk=1, n_lang=1)
batch = {k: (torch.from_numpy(np.asarray(v)).cuda() if not isinstance(v, list) else v) for k, v in nb.items()}
batch["target_lengths_host"] = [int(x) for x in np.asarray(nb["target_lengths"])]
m0 = Tacotron(hp); initialize_variables(m0); init = {k: v.clone() for k, v in m0.state_dict().items()}
del m0
keep = None
for it in range(4):
    m = Tacotron(hp); m.load_state_dict(init); m = m.to("cuda").train()
    tr = HipTrainer(m, hp, dist=False)
    if os.environ.get("LAB_REUSE") == "1":
        if keep is None: keep = torch.cuda.Stream()
        tr._enc_stream = keep
        from b2s_hip import lib as L
        import ctypes as C
        L.check(tr.lib.b2s_model_set_side_stream(tr.eng.handle, C.c_void_p(keep.cuda_stream)))
    for _ in range(8): tr.train_step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30): tr.train_step(batch)
    torch.cuda.synchronize()
    print("trainer %d: %.3f ms per step (enc stream %s)" % (it, (time.perf_counter() - t0) / 30 * 1e3, hex(tr._enc_stream.cuda_stream) if tr._enc_stream is not None else None), flush=True)
    tr.close(); del tr, m; gc.collect(); torch.cuda.empty_cache()
