#!/usr/bin/env python3
"""Context only (not part of the product): what the vendor library (torch.matmul -> hipBLASLt / rocBLAS) reaches on the
training step's GEMM shapes, bf16, same box -- a calibration point for the hand-written kernels' numbers in
profiles/README.md.  Plain GEMM only (no fused epilogues)."""
import torch

SHAPES = [("fwd attn out", 8148, 768, 768, "nt"), ("fwd qkv", 8148, 2304, 768, "nt"), ("fwd ffn in", 8148, 3072, 768, "nt"),
          ("fwd ffn out", 8148, 768, 3072, "nt"), ("enc ffn out", 1596, 512, 2048, "nt"), ("dX ffn in", 8148, 768, 3072, "nn"),
          ("dX ffn out", 8148, 3072, 768, "nn"), ("dW attn out", 768, 768, 8148, "tn"), ("dW ffn in", 3072, 768, 8148, "tn"),
          ("4096^3", 4096, 4096, 4096, "nt"), ("8192^3", 8192, 8192, 8192, "nt")]


def main():
    dev = "cuda"
    tot_us = tot_fl = 0.0
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
        print("%-14s M=%5d N=%5d K=%5d %s  %8.2f us  %7.1f TF" % (name, M, N, K, form, us, fl / us / 1e6))
        if "^3" not in name:
            tot_us += us; tot_fl += fl
    print("step shapes: %.1f TF/s aggregate" % (tot_fl / tot_us / 1e6))


if __name__ == "__main__":
    main()
