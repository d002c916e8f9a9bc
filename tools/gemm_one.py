#!/usr/bin/env python3
"""Run one GEMM shape a few times (for rocprofv3 --pmc)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "few-shot-transformer-tts_amd"))
import torch
from b2s_hip import ops
ta, tb, M, N, K = [int(x) for x in sys.argv[1:6]]
A = torch.randn((K, M) if ta else (M, K), device="cuda").to(torch.bfloat16).view(torch.int16)
B = torch.randn((K, N) if tb else (N, K), device="cuda").to(torch.bfloat16).view(torch.int16)
out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if ta else torch.int16)
for _ in range(5):
    ops.gemm(1, A, B, M, N, K, trans_a=bool(ta), trans_b=bool(tb), out=out, c_fp32=bool(ta), accumulate=bool(ta))
torch.cuda.synchronize()
