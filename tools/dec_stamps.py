#!/usr/bin/env python3
"""Per-phase timing of the fused decode kernels from in-kernel wall-clock stamps (100 MHz).
usage: bash tools/build_stamped_decode.sh; B2S_LIB_PATH=$PWD/tools/bin/libb2s_hip_stamped.so python tools/dec_stamps.py [frames]
The stamps of the LAST launch of each kernel kind survive, so the numbers describe frame `frames - 1`."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "few-shot-transformer-tts_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from hyperparams import hparams as hp
from transformer.tacotron import Tacotron, initialize_variables
import synthesize
from oracle import synth, make_config
from b2s_hip import lib as L

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 500
dev = torch.device("cuda", 0)
B, S = 64, 160
hp.parse("compute_dtype=bf16,max_generation_frames=%d" % frames)
torch.manual_seed(0)
m = Tacotron(hp); initialize_variables(m)
with torch.no_grad():
    m.decoder.stop_net.bias.fill_(-100.0)
m = m.to(dev); m.eval(); m.decoder.train()
cfg = make_config("")
nb = synth.synthetic_batch(cfg, B, S, 4, seed=0, in_lens=[S] * B, n_spk=1, n_lang=1)
nb.pop("mel_targets"); nb.pop("target_lengths")
batch = {k: (torch.from_numpy(np.asarray(v)).to(dev) if not isinstance(v, list) else v) for k, v in nb.items()}
synthesize.eval_batch(m, batch, use_bar=False, bar_interval=-1, sync_interval=64, device_results=True)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["B2S_LIB_PATH"])
buf = np.zeros((3, 512, 16), dtype=np.uint64)
rc = lib.b2s_df_stamp_read(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
labels = json.load(open(os.path.join(ROOT, "tools/bin/decode_fused_stamped.json")))
names = {0: ("self-attention", [x for x in labels["attn"] if x != "load_x_ln"]), 1: ("cross-attention", [x for x in labels["attn"] if x != "load_x_ln"]),
         2: ("ffn", labels["ffn"][:labels["ffn"].index("store_partial") + 1])}
for kind in range(3):
    name, lab = names[kind]
    st = buf[kind].astype(np.int64)
    n_wg = int((st[:, 0] > 0).sum())
    nph = int((st[0] > 0).sum())
    st = st[:n_wg, :nph]
    t0 = st[:, 0].min()
    print("%s: %d workgroups, %d stamps; kernel span first start -> last end %.2f us; start skew %.2f us" %
          (name, n_wg, nph, (st[:, -1].max() - t0) / 100.0, (st[:, 0].max() - t0) / 100.0))
    for p in range(1, nph):
        d = (st[:, p] - st[:, p - 1]) / 100.0
        print("   %-14s median %6.2f  p90 %6.2f  max %6.2f us" % (lab[p] if p < len(lab) else "?", np.median(d), np.percentile(d, 90), d.max()))
    tot = (st[:, -1] - st[:, 0]) / 100.0
    print("   %-14s median %6.2f  p90 %6.2f  max %6.2f us" % ("workgroup", np.median(tot), np.percentile(tot, 90), tot.max()))
