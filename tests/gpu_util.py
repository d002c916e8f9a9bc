"""Shared helpers for the GPU parity tests (the HIP path is reached through the C ABI via ctypes)."""
import numpy as np
import torch

DEV = "cuda"


def bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


def to_dev_compute(x, dtype):
    """fp32 CPU tensor -> device tensor in compute dtype (bf16 bits as int16)."""
    x = x.contiguous().to(DEV)
    if dtype:
        return x.to(torch.bfloat16).view(torch.int16)
    return x


def from_dev_compute(x, dtype):
    if dtype:
        return x.view(torch.bfloat16).to(torch.float32).cpu()
    return x.cpu()


def relerr(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def report(name, a, b):
    a, b = a.double(), b.double()
    d = (a - b).abs()
    return "%s: max|d|=%.3e max|ref|=%.3e rel=%.3e" % (name, float(d.max()), float(b.abs().max()),
                                                        float(d.max() / (b.abs().max() + 1e-12)))
