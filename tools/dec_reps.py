import os, sys, time
ROOT="/root/repo"
for p in (ROOT, os.path.join(ROOT, "few-shot-transformer-tts_amd")): sys.path.insert(0, p)
import numpy as np, torch
from hyperparams import hparams as hp
from transformer.tacotron import Tacotron, initialize_variables
import synthesize
from oracle import synth, make_config
dev=torch.device("cuda",0); B,S,frames=64,160,1000
hp.parse("compute_dtype=bf16,max_generation_frames=%d"%frames)
torch.manual_seed(0); m=Tacotron(hp); initialize_variables(m)
with torch.no_grad(): m.decoder.stop_net.bias.fill_(-100.0)
m=m.to(dev); m.eval(); m.decoder.train()
cfg=make_config(""); nb=synth.synthetic_batch(cfg,B,S,4,seed=0,in_lens=[S]*B,n_spk=1,n_lang=1); nb.pop("mel_targets"); nb.pop("target_lengths")
batch={k:(torch.from_numpy(np.asarray(v)).to(dev) if not isinstance(v,list) else v) for k,v in nb.items()}
for i in range(6):
    torch.cuda.synchronize(); t0=time.perf_counter()
    r=synthesize.eval_batch(m,batch,use_bar=False,bar_interval=-1,sync_interval=64,device_results=True)
    torch.cuda.synchronize(); print("rep",i,"%.4f ms/frame"%((time.perf_counter()-t0)/frames*1e3), flush=True)
    del r
