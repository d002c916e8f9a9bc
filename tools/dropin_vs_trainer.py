"""Full-size model, LJ-shaped batches, dropout on: N steps of the reference's own loop on the drop-in modules (m(**batch), compute_loss, zero_grad, backward,
torch.optim.Adam.step, LambdaLR.step -- train.py:171-174,188-190) against N steps of HipTrainer from the same initial state, the same batches and the same
dropout seeds (padded decoder rows in both, so the masks coincide; LAB_RAGGED=1: the trainer on ragged rows, other masks).  Same arithmetic except for who applies Adam:
the loss curves must agree to the bf16 run-to-run level.  usage: python tools/dropin_vs_trainer.py [steps]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "few-shot-transformer-tts_amd"))
from hyperparams import hparams as hp
from transformer.tacotron import Tacotron, initialize_variables, compute_loss, learning_rate_schedule
from b2s_hip.trainer import HipTrainer
from benchdata import synthetic_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
hp.parse("compute_dtype=bf16")
torch.manual_seed(0)
m0 = Tacotron(hp); initialize_variables(m0)
init = {k: v.clone() for k, v in m0.state_dict().items()}
batches = []
for s in range(4):
    nb = synthetic_batch(hp, 14, 114, 582, seed=s, n_spk=1, n_lang=1)
    b = {k: (torch.from_numpy(np.asarray(v)).cuda() if not isinstance(v, list) else v) for k, v in nb.items()}
    batches.append((b, [int(x) for x in np.asarray(nb["target_lengths"])]))

def run(kind):
    torch.manual_seed(1)
    m = Tacotron(hp); m.load_state_dict(init); m = m.cuda().train()
    out = []
    if kind == "trainer":
        tr = HipTrainer(m, hp, dist=False)
        for i in range(N):
            b, lens = batches[i % 4]
            bb = dict(b)
            if os.environ.get("LAB_RAGGED") == "1": bb["target_lengths_host"] = lens      # (ragged rows index the decoder's dropout sites differently: other masks)
            out.append(float(tr.train_step(bb)[0]))
        tr.close()
    else:
        opt = torch.optim.Adam(m.parameters(), lr=hp.max_lr, betas=(0.9, 0.999), eps=hp.adam_eps)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda st: learning_rate_schedule(st, hp))
        for i in range(N):
            b, _ = batches[i % 4]
            o = m(**b)
            losses = compute_loss(m, b["mel_targets"], b["target_lengths"], o, hp)
            opt.zero_grad(); losses["loss"].backward(); opt.step(); sched.step()
            out.append(float(losses["loss"].detach()))
    torch.cuda.synchronize()
    return np.array(out), {k: v.detach().float().cpu() for k, v in m.state_dict().items() if v.dtype.is_floating_point}

a, pa = run("trainer")
b, pb = run("dropin")
print("step   trainer    drop-in")
for i in list(range(0, N, max(1, N // 10))) + [N - 1]:
    print("%4d  %9.5f  %9.5f" % (i, a[i], b[i]))
rel = np.abs(a - b) / np.abs(a)
num = sum(float(((pa[k] - pb[k]) ** 2).sum()) for k in pa); den = sum(float(((pa[k] - init[k].float()) ** 2).sum()) for k in pa)
print("max relative loss difference %.2e, last-10 mean %.5f / %.5f; parameter distance / movement %.3f" % (rel.max(), a[-10:].mean(), b[-10:].mean(), (num / max(den, 1e-30)) ** 0.5))
