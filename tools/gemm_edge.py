#!/usr/bin/env python3
"""Does the partial last row tile (8148 = 31 x 256 + 212 rows) gate the step's GEMMs?  Same N, K at M = 8148 and M = 8192 (GPU only);
bf16 output and the fp32 residual epilogue."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "few-shot-transformer-tts_amd"))
import torch
from b2s_hip import ops
dev = "cuda"
for N, K in [(768, 768), (2304, 768), (3072, 768), (768, 3072)]:
    for res in (False, True):
        for M in (8148, 8192, 8148, 8192):
            A = torch.randn(8192, K, device=dev).to(torch.bfloat16).view(torch.int16)
            B = torch.randn(N, K, device=dev).to(torch.bfloat16).view(torch.int16)
            R = torch.randn(8192, N, device=dev) if res else None
            out = torch.zeros(8192, N, device=dev, dtype=torch.float32 if res else torch.int16)
            kw = dict(out=out, c_fp32=res, residual=R)
            for _ in range(3): ops.gemm(1, A, B, M, N, K, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): ops.gemm(1, A, B, M, N, K, **kw)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 50
            print("N=%4d K=%4d %s M=%d: %6.1f us" % (N, K, "fp32+residual" if res else "bf16 out     ", M, us), flush=True)
