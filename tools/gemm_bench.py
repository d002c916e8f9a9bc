#!/usr/bin/env python3
"""Micro-benchmark of the MFMA GEMM kernel on the shapes of the LJSpeech-shaped training step (GPU only)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "few-shot-transformer-tts_amd"))
import torch  # noqa: E402
from b2s_hip import ops  # noqa: E402

SHAPES = [  # (ta, tb, M, N, K, batch, label)
    (0, 0, 8148, 3072, 768, 1, "ffn1 fwd"), (0, 0, 8148, 768, 3072, 1, "ffn2 fwd"), (0, 0, 8148, 2304, 768, 1, "qkv fwd"),
    (0, 0, 8148, 768, 768, 1, "proj fwd"), (0, 1, 8148, 768, 3072, 1, "ffn1 dx"), (0, 1, 8148, 3072, 768, 1, "ffn2 dx"),
    (1, 1, 3072, 768, 8148, 1, "ffn1 dw"), (1, 1, 768, 768, 8148, 1, "proj dw"), (0, 0, 4096, 4096, 4096, 1, "square 4k"),
]


def main():
    dev = "cuda"
    dtype = 1
    for ta, tb, M, N, K, batch, label in SHAPES:
        a_shape = (K, M) if ta else (M, K)
        b_shape = (K, N) if tb else (N, K)
        A = torch.randn(a_shape, device=dev).to(torch.bfloat16).view(torch.int16)
        B = torch.randn(b_shape, device=dev).to(torch.bfloat16).view(torch.int16)
        out = torch.zeros(M, N, device=dev, dtype=torch.float32 if ta else torch.int16)
        kw = dict(trans_a=bool(ta), trans_b=bool(tb), out=out, c_fp32=bool(ta), accumulate=bool(ta))
        for _ in range(3):
            ops.gemm(dtype, A, B, M, N, K, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            ops.gemm(dtype, A, B, M, N, K, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print("%-10s ta=%d tb=%d M=%5d N=%5d K=%5d  %8.1f us  %7.1f TFLOP/s" % (label, ta, tb, M, N, K, us, 2.0 * M * N * K / us / 1e6))


if __name__ == "__main__":
    main()
