#!/usr/bin/env python3
"""Context only (not part of the product): what the vendor library (torch.matmul -> hipBLASLt / rocBLAS) reaches on the
training step's GEMM shapes, bf16, same box -- a calibration point for the hand-written kernels' numbers in
profiles/README.md.  Plain GEMM only (no fused epilogues)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "few-shot-transformer-tts_amd"))

SHAPES = [("fwd attn out", 8148, 768, 768, "nt"), ("fwd qkv", 8148, 2304, 768, "nt"), ("fwd ffn in", 8148, 3072, 768, "nt"),
          ("fwd ffn out", 8148, 768, 3072, "nt"), ("enc ffn out", 1596, 512, 2048, "nt"), ("dX ffn in", 8148, 768, 3072, "nn"),
          ("dX ffn out", 8148, 3072, 768, "nn"), ("dW attn out", 768, 768, 8148, "tn"), ("dW ffn in", 3072, 768, 8148, "tn"),
          ("4096^3", 4096, 4096, 4096, "nt"), ("8192^3", 8192, 8192, 8192, "nt")]


def ours_us(M, N, K, form, dev="cuda"):
    """The repo's GEMM through the C ABI op (plain epilogue: bf16 output for NT / NN, fp32 overwrite for TN), same shapes, same process."""
    from b2s_hip import ops
    ta, tb = {"nt": (0, 0), "nn": (0, 1), "tn": (1, 1)}[form]
    A = torch.randn((K, M) if ta else (M, K), device=dev).to(torch.bfloat16).view(torch.int16)
    B = torch.randn((K, N) if tb else (N, K), device=dev).to(torch.bfloat16).view(torch.int16)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if ta else torch.int16)
    kw = dict(trans_a=bool(ta), trans_b=bool(tb), out=out, c_fp32=bool(ta), accumulate=False)
    for _ in range(5):
        ops.gemm(1, A, B, M, N, K, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.gemm(1, A, B, M, N, K, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 50


def main():
    dev = "cuda"
    tot_us = tot_fl = tot_ours = 0.0
    for name, M, N, K, form in SHAPES:
        if form == "nt":
            a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
            f = lambda: a @ b.t()
        elif form == "nn":
            a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
            f = lambda: a @ b
        else:
            a = torch.randn(K, M, device=dev, dtype=torch.bfloat16); b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
            f = lambda: a.t() @ b
        for _ in range(5):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        fl = 2.0 * M * N * K
        mine = ours_us(M, N, K, form)
        print("| %s %dx%dx%d %s | %.1f us = %.0f TFLOP/s | %.1f us = %.0f TFLOP/s |" % (name, M, N, K, form.upper(), us, fl / us / 1e6, mine, fl / mine / 1e6))
        if "^3" not in name:
            tot_us += us; tot_fl += fl; tot_ours += mine
    print("| the step's shape mix | %.0f TFLOP/s | %.0f TFLOP/s |" % (tot_fl / tot_us / 1e6, tot_fl / tot_ours / 1e6))


if __name__ == "__main__":
    main()
