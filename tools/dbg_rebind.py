import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "few-shot-transformer-tts_amd")
import torch
from oracle import synth, TINY96
from test_gpu_model import build, dev_batch
from b2s_hip.trainer import HipTrainer
def run(swap):
    m, cfg, _, hp = build(TINY96, compute_dtype="fp32")
    nb = synth.synthetic_batch(cfg, B=3, S=11, T=23, seed=7, in_lens=[11, 7, 4], tgt_lens=[23, 15, 9])
    b = dev_batch(nb)
    m.train()
    t = HipTrainer(m, hp)
    t.train_step(b); torch.cuda.synchronize()
    if swap:
        with torch.no_grad():
            for n, p in m.named_parameters():
                p.data = p.data.clone()
    for _ in range(2):
        v = t.train_step(b)
    torch.cuda.synchronize()
    print("swap", swap, "loss", v.cpu().numpy()[:3])
    return {k: v.detach().clone() for k, v in m.state_dict().items()}
a = run(False); b_ = run(False); c = run(True)
for name, x in (("a-b", b_), ("a-c", c)):
    worst = sorted(((float((x[k].double() - a[k].double()).abs().max()), k) for k in a), reverse=True)[:5]
    print(name, worst)
