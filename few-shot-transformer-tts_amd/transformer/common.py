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
    """Best-of-8 resampling inside (-2 std, 2 std) (common.py:90-105)."""
    with torch.no_grad():
        tmp = tensor.new_empty(tuple(tensor.shape) + (8,)).normal_(mean=mean, std=std)
        valid = (tmp < 2 * std) & (tmp > -2 * std)
        ind = valid.max(-1, keepdim=True)[1]
        return tmp.gather(-1, ind).squeeze(-1)


def variance_scaling_initializer(tensor, factor=2.0):
    """FAN_AVG truncated normal, std = sqrt(1.3 * factor / n) (common.py:108-124)."""
    fan_in, fan_out = tensor.shape[1], tensor.shape[0]
    for dim in tensor.shape[2:]:
        fan_in *= dim
        fan_out *= dim
    n = (fan_in + fan_out) / 2
    return truncated_normal(tensor, std=float(np.sqrt(1.3 * factor / n)))
