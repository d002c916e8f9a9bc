"""Tensor helpers with the reference's names and semantics (transformer/common.py:1-124).

On the MI355X path none of these run per step: masks are derived from lengths inside the HIP kernels and
the sinusoid tables live on the device (csrc/engine.hip:fill_pe).  They are kept as the public helper
API of the package (and for initialize_variables, which runs once on the host).
"""
import numpy as np
import torch


def get_sinusoid_encoding_table(length, channels, min_timescale=1, max_timescale=1e4):
    """[length, channels] float32 table: sin | cos halves, computed in float64 (common.py:4-29)."""
    position = np.arange(length)
    n = channels // 2
    inc = np.log(float(max_timescale) / float(min_timescale)) / (n - 1)
    inv = min_timescale * np.exp(np.arange(n) * -inc)
    st = position[:, None] * inv[None, :]
    signal = np.concatenate([np.sin(st), np.cos(st)], axis=1)
    signal = np.pad(signal, [[0, 0], [0, channels % 2]])
    return torch.FloatTensor(signal)


class AttentionBias(torch.Tensor):
    """Dense additive bias tensor (as the reference builds it) that also remembers how it was made, so
    MultiheadAttention can use the length/causal mask inside the kernel instead of reading the tensor."""

    @staticmethod
    def make(dense, mode, lengths=None):
        b = dense.as_subclass(AttentionBias)
        b.b2s_mode = mode
        b.b2s_lengths = lengths
        return b


def attention_bias(inputs, mode, inf=-1e20):
    """common.py:32-48 -- 'causal': inputs = length -> [1,1,L,L]; 'masking': inputs = bool mask [B,L] -> [B,1,1,L]."""
    if mode == "causal":
        dense = (torch.triu(torch.ones([inputs, inputs]), diagonal=1) * inf).reshape([1, 1, inputs, inputs])
        return AttentionBias.make(dense, "causal") if inf == -1e20 else dense
    elif mode == "masking":
        dense = ((1.0 - inputs.float()) * inf).unsqueeze(1).unsqueeze(1)
        if inf == -1e20 and inputs.dim() == 2:
            m = inputs.bool()
            lengths = m.sum(-1).to(torch.int32)
            prefix = bool((m == (torch.arange(m.shape[1], device=m.device)[None, :] < lengths[:, None])).all())
            if prefix:
                return AttentionBias.make(dense, "masking", lengths)
        return dense
    raise ValueError("Unknown mode %s" % mode)


def impute(x, lengths, channels_last=True):
    """Zero every time step >= length (common.py:51-70)."""
    max_length = x.shape[1] if channels_last else x.shape[-1]
    mask = torch.arange(max_length, device=lengths.device)[None, :] < lengths[:, None]
    for _ in range(len(x.shape) - 2):
        mask = mask.unsqueeze(-1) if channels_last else mask.unsqueeze(1)
    return x * mask


def mask_reduce(loss, lengths, per_sample=False):
    """Length-masked mean (common.py:73-87)."""
    if per_sample:
        return impute(loss, lengths).sum(-1) / lengths
    return impute(loss, lengths).sum() / lengths.sum()


def truncated_normal(tensor, mean=0, std=0.5):
    """A new tensor shaped like `tensor`, drawn from N(mean, std) truncated to two standard deviations (the distribution
    the reference approximates with best-of-8 resampling, common.py:90-105; only the distribution is contractual)."""
    out = torch.empty_like(tensor, dtype=torch.float32)
    torch.nn.init.trunc_normal_(out, mean=float(mean), std=float(std), a=mean - 2.0 * std, b=mean + 2.0 * std)
    return out


def variance_scaling_initializer(tensor, factor=2.0):
    """TF-style variance scaling, FAN_AVG mode: truncated normal with std = sqrt(1.3 * factor / ((fan_in + fan_out) / 2)),
    fans counted over the receptive field for conv weights (common.py:108-124)."""
    fan_in, fan_out = torch.nn.init._calculate_fan_in_and_fan_out(tensor)
    return truncated_normal(tensor, std=(1.3 * factor * 2.0 / (fan_in + fan_out)) ** 0.5)
