"""How long does the main stream wait for the encoder backward at the end of the step (unprofiled)?  Events around the trainer's calls."""
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
if os.environ.get("B2S_LAB_HP"):                     # e.g. "freeze_encoder=true,guided_attention_weight=1.0"
    hp.parse(os.environ["B2S_LAB_HP"])
if os.environ.get("B2S_FORCE_DP"):                 # the exchange path on a 1-rank RCCL group
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    torch.distributed.init_process_group("nccl", init_method="env://", device_id=torch.device("cuda", 0))
torch.manual_seed(0)
m = Tacotron(hp); initialize_variables(m); m = m.to("cuda").train()
tr = HipTrainer(m, hp)
nb = synthetic_batch(hp, 14, 114, 582, seed=0, n_spk=1, n_lang=1)
batch = {k: (torch.from_numpy(np.asarray(v)).cuda() if not isinstance(v, list) else v) for k, v in nb.items()}
batch["target_lengths_host"] = [int(x) for x in np.asarray(nb["target_lengths"])]      # ragged decoder rows, as bench.py
eng = tr.eng
ev = {}
def mark(name, stream=None):
    e = torch.cuda.Event(enable_timing=True); e.record(stream or torch.cuda.current_stream()); ev.setdefault(name, []).append(e)
o_dec, o_enc, o_lb, o_ef = eng.decoder_backward, eng.encoder_backward, eng.loss_backward, eng.encoder_forward
def dec_b(*a, **k):
    r = o_dec(*a, **k); mark("dec_bwd_end"); return r
def enc_b(*a, **k):
    mark("enc_bwd_start"); r = o_enc(*a, **k); mark("enc_bwd_end"); return r
def loss_b(*a, **k):
    mark("bwd_start"); return o_lb(*a, **k)
def enc_f(*a, **k):
    mark("enc_fwd_start"); r = o_ef(*a, **k); mark("enc_fwd_end"); return r
eng.decoder_backward, eng.encoder_backward, eng.loss_backward, eng.encoder_forward = dec_b, enc_b, loss_b, enc_f
for _ in range(8): tr.train_step(batch)
torch.cuda.synchronize(); ev.clear()
N = 20
for _ in range(N):
    mark("step_start"); tr.train_step(batch); mark("step_end")
torch.cuda.synchronize()
def avg(a, b):
    if a not in ev or b not in ev: return float("nan")
    return sum(x.elapsed_time(y) for x, y in zip(ev[a], ev[b])) / N * 1e3
print("step %.0f us | forward+loss %.0f | backward start -> decoder backward end (main stream) %.0f | decoder backward end -> encoder backward end %.0f | "
      "encoder backward %.0f us on its stream | encoder forward %.0f | encoder backward end -> step end (optimizer) %.0f" % (
      avg("step_start", "step_end"), avg("step_start", "bwd_start"), avg("bwd_start", "dec_bwd_end"), avg("dec_bwd_end", "enc_bwd_end"),
      avg("enc_bwd_start", "enc_bwd_end"), avg("enc_fwd_start", "enc_fwd_end"), avg("enc_bwd_end", "step_end")))
