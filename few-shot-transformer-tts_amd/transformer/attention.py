"""Multi-head attention with the reference's module surface (transformer/attention.py:1-122), computed by
the HIP kernels: MFMA projections + fused-mask softmax attention core (csrc/gemm.hip, csrc/rowops.hip)."""
import torch
from torch import nn

from b2s_hip import ops
from b2s_hip.engine import DTYPES
from transformer.common import AttentionBias


def split_heads(x, num_heads):
    """[B, L, C] -> [B, H, L, C/H] (attention.py:6-15).  Layout helper only; the kernels index heads in place."""
    assert x.shape[-1] % num_heads == 0, str(x.shape)
    return x.reshape(x.shape[:-1] + (num_heads, x.shape[-1] // num_heads)).permute(0, 2, 1, 3)


def combine_heads(x):
    """[B, H, L, C] -> [B, L, H*C] (attention.py:18-26)."""
    x = x.permute([0, 2, 1, 3])
    return x.reshape(x.shape[:-2] + (x.shape[-1] * x.shape[-2],))


class HipLinear(nn.Module):
    """Parameter holder with nn.Linear's state_dict surface (weight [out,in], optional bias); forward = HIP GEMM."""

    def __init__(self, in_features, out_features, bias=True):
        super(HipLinear, self).__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.zeros(out_features)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        self.compute_dtype = "fp32"

    def forward(self, x, relu=False):
        return ops.linear(x, self.weight, self.bias, relu, DTYPES[self.compute_dtype])


class MultiheadAttention(nn.Module):
    def __init__(self, key_size, value_size, is_self_attention, num_heads, dropout_rate=0.1, compute_dtype="fp32"):
        super(MultiheadAttention, self).__init__()
        assert key_size % num_heads == 0, "key_size=%d, num_heads=%d" % (key_size, num_heads)
        assert value_size % num_heads == 0, "value_size=%d, num_heads=%d" % (value_size, num_heads)
        if is_self_attention:
            self.qkv_transform = HipLinear(key_size, key_size * 2 + value_size, bias=False)
        else:
            self.q_transform = HipLinear(key_size, key_size, bias=False)
            self.kv_transform = HipLinear(key_size, key_size + value_size, bias=False)
        self.output_transform = HipLinear(key_size, key_size, bias=False)
        self.attn_dropout = nn.Dropout(dropout_rate)        # rate holder (state-free); the mask is drawn in-kernel
        self.num_heads = num_heads
        self.key_size = key_size
        self.value_size = value_size
        self.compute_dtype = compute_dtype
        self._calls = 0

    def _set_dtype(self):
        for m in self.children():
            if isinstance(m, HipLinear):
                m.compute_dtype = self.compute_dtype

    def forward(self, queries, memories, bias):
        """queries [B,Lq,C], memories [B,Lk,C] or None, bias from attention_bias() (or any additive tensor
        broadcastable to [B,1,Lq,Lk]) -> {"outputs": [B,Lq,C], "align": [B,H,Lk,Lq]}  (attention.py:94-122)."""
        self._set_dtype()
        dt = DTYPES[self.compute_dtype]
        C = self.key_size
        if memories is None:
            qkv = self.qkv_transform(queries)
            q, k, v = qkv[..., :C].contiguous(), qkv[..., C:2 * C].contiguous(), qkv[..., 2 * C:].contiguous()
        else:
            q = self.q_transform(queries)
            kv = self.kv_transform(memories)
            k, v = kv[..., :C].contiguous(), kv[..., C:].contiguous()
        mask_mode, klen, dense = 0, None, None
        if isinstance(bias, AttentionBias) and getattr(bias, "b2s_mode", None) == "causal":
            mask_mode = 2
        elif isinstance(bias, AttentionBias) and getattr(bias, "b2s_mode", None) == "masking":
            mask_mode, klen = 1, bias.b2s_lengths.to(q.device).contiguous()
        elif bias is not None:
            dense = bias.to(q.device)
            if dense.dim() != 4 or dense.shape[1] != 1:
                raise ValueError("bias must be broadcastable as [B or 1, 1, Lq or 1, Lk]")
        p = self.attn_dropout.p if self.training else 0.0
        self._calls += 1
        seed = (int(torch.initial_seed()) * 1000003 + self._calls) & 0xFFFFFFFFFFFF
        ctx, probs = ops.attention_core(q, k, v, self.num_heads, mask_mode, klen, dense, p, seed, dt)
        x = self.output_transform(ctx)
        return {"outputs": x, "align": probs.permute(0, 1, 3, 2)}
