#!/usr/bin/env python3
"""Experiment: the encoder forward (small kernels, 78..208 workgroups on 256 CUs) as 1 / 2 / 4 independent utterance-range chains on
separate HIP streams.  usage: python tools/enc_chain_lab.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "few-shot-transformer-tts_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from hyperparams import hparams as hp
from transformer.tacotron import Tacotron, initialize_variables
from b2s_hip.engine import _i32
from oracle import synth, make_config

dev = torch.device("cuda", 0)
hp.parse("compute_dtype=bf16")
cfg = make_config("")
torch.manual_seed(0)
m = Tacotron(hp); initialize_variables(m); m = m.to(dev).train()
eng = m.engine(); eng.ensure_bound()
nb = synth.synthetic_batch(cfg, 14, 114, 582, seed=0, n_spk=1, n_lang=1)
b = {k: (torch.from_numpy(np.asarray(v)).to(dev) if not isinstance(v, list) else v) for k, v in nb.items()}
in32 = _i32(b["input_lengths"])


def run(nchains, reps=30, bwd=False):
    B = 14
    bounds = [(i * B // nchains, (i + 1) * B // nchains) for i in range(nchains)]
    streams = [torch.cuda.Stream() for _ in range(nchains)]
    main = torch.cuda.current_stream()
    dmem = torch.randn(B, 114, 768, device=dev) * 1e-3
    def once():
        ctxs = []
        for (lo, hi), st in zip(bounds, streams):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                mem, c = eng.encoder_forward(b["inputs"][lo:hi], in32[lo:hi], b["input_spk_ids"][lo:hi], b["input_language_vecs"][lo:hi], True, eng.next_seed(), bwd)
                ctxs.append((mem, c))
        if bwd:
            for (lo, hi), st, (mem, c) in zip(bounds, streams, ctxs):
                with torch.cuda.stream(st):
                    eng.begin_backward()
                    eng.encoder_backward(c, dmem[lo:hi])
        for st in streams:
            main.wait_stream(st)
        for mem, c in ctxs:
            if c is not None: c.free()
    for _ in range(5): once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): once()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

if os.environ.get("ENC_CHAINS"):
    n = int(os.environ["ENC_CHAINS"])
    print("encoder fwd, %d chain(s): %.3f ms" % (n, run(n, reps=10)))
else:
    for bwd in (False, True):
        for n in (1, 2, 4, 7):
            print("encoder %s, %d chain(s): %.3f ms" % ("fwd+bwd" if bwd else "fwd", n, run(n, bwd=bwd)))
