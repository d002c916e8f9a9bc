"""Persistent multi-round GEMM kernel against the plain one, per epilogue mode (lab build: B2S_LAB_GEMM_PERSIST=0 | 1 through B2S_LIB_PATH)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "few-shot-transformer-tts_amd")]
import torch
from b2s_hip import ops, lib as L

dev = torch.device("cuda", 0)
l = L.load()
M = 8148
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for what, N, K, tb in (("q / out", 768, 768, False), ("dX q / out", 768, 768, True), ("dX qkv", 768, 2304, True), ("qkv", 2304, 768, False), ("ffn-in", 3072, 768, False), ("dX ffn-out", 3072, 768, True)):
    A = ops.to_compute(torch.randn(M, K, device=dev), 1)
    B = ops.to_compute(torch.randn(K, N, device=dev) if tb else torch.randn(N, K, device=dev), 1)
    out = torch.empty(M, N, dtype=torch.int16, device=dev)
    for mode, kw in (("plain", {}), ("relu", dict(relu=True)), ("relu+drop", dict(relu=True, drop_p=0.1, seed=3))):
        for _ in range(3):
            ops.gemm(1, A, B, M, N, K, trans_b=tb, c_fp32=False, out=out, **kw)
        e0.record()
        for _ in range(30):
            ops.gemm(1, A, B, M, N, K, trans_b=tb, c_fp32=False, out=out, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 30
        print("%-12s %-10s %6.1f us  %6.1f TFLOP/s" % (what, mode, us, 2.0 * M * N * K / us / 1e6))
