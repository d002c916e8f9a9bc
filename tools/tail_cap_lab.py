"""One default-size bf16 training step with the tail policy's capped persistent weight-gradient launches; prints a gradient checksum."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "few-shot-transformer-tts_amd"))
from hyperparams import hparams as hp
from transformer.tacotron import Tacotron, initialize_variables
from b2s_hip.trainer import HipTrainer
from benchdata import synthetic_batch
hp.parse("compute_dtype=bf16,transformer_dropout_rate=0.0,decoder_dropout_rate=0.0")
torch.manual_seed(0)
m = Tacotron(hp); initialize_variables(m); m = m.to("cuda").train()
tr = HipTrainer(m, hp)
nb = synthetic_batch(hp, 14, 114, 582, seed=0, n_spk=1, n_lang=1)
batch = {k: (torch.from_numpy(np.asarray(v)).cuda() if not isinstance(v, list) else v) for k, v in nb.items()}
v = tr.train_step(batch)
torch.cuda.synchronize()
g = tr.eng._gflat
print("loss %.6f grad abs-sum %.6e grad sq-sum %.6e" % (float(v[0]), float(g.abs().sum()), float((g.double() ** 2).sum())), flush=True)
