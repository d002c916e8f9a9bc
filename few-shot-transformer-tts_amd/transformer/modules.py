"""Transformer encoder/decoder stacks with the reference's module surface and state_dict layout
(transformer/modules.py:1-145).  Inside Tacotron the stacks execute as fused segments of libb2s_hip (csrc/engine.hip) and
these classes only hold the parameters under the reference's names.  Called on their own (`stack(inputs, lengths)`), they run
the reference's layer sequence through the op-level HIP kernels (LayerNorm, GEMM, fused attention) with torch doing the
element-wise glue between them -- a convenience path for inspection and unit use, not the benchmarked one."""
import torch
from torch import nn

from b2s_hip import ops
from b2s_hip.engine import DTYPES
from transformer.attention import MultiheadAttention, HipLinear
from transformer.common import *  # noqa: F401,F403  (the reference re-exports the helpers from here)


class HipLayerNorm(nn.Module):
    """nn.LayerNorm state_dict surface (weight, bias), eps=1e-6; forward = HIP wavefront-shuffle kernel."""

    def __init__(self, size, eps=1e-6):
        super(HipLayerNorm, self).__init__()
        self.weight = nn.Parameter(torch.ones(size))
        self.bias = nn.Parameter(torch.zeros(size))
        self.eps = eps
        self.compute_dtype = "fp32"

    def forward(self, x):
        return ops.layernorm(x, self.weight, self.bias, self.eps, DTYPES[self.compute_dtype])


class FFNLayer(nn.Module):
    def __init__(self, input_size, hidden_size, output_size, dropout_rate=0.1, compute_dtype="fp32"):
        super(FFNLayer, self).__init__()
        self.input_layer = HipLinear(input_size, hidden_size, bias=False)
        self.dropout = nn.Dropout(dropout_rate)
        self.output_layer = HipLinear(hidden_size, output_size, bias=False)
        self.compute_dtype = compute_dtype

    def forward(self, inputs):
        """Linear -> ReLU -> dropout -> Linear (modules.py:15-20).  Inside the model the fused engine path applies the
        dropout in the GEMM epilogue; stand-alone it is an element-wise mask from the same counter RNG."""
        self.input_layer.compute_dtype = self.output_layer.compute_dtype = self.compute_dtype
        h = self.input_layer(inputs, relu=True)
        return self.output_layer(_dropout(self, h, self.dropout.p))


def _dropout(module, x, p):
    """Inverted dropout with the engine's counter RNG (csrc/b2s_common.h: b2s_keep): a fresh mask per call, regenerated
    identically in backward by autograd through the saved mask (stand-alone modules only)."""
    if not module.training or p <= 0:
        return x
    module._drop_calls = getattr(module, "_drop_calls", 0) + 1
    seed = (int(torch.initial_seed()) * 1000003 + module._drop_calls) & 0xFFFFFFFFFFFF
    return x * ops.dropout_mask(x.shape, p, seed, 23, x.device)


class _Stack(nn.Module):
    """Parameter holder of a pre-LayerNorm Transformer stack.  The sub-layer lists are registered in `_LISTS` order, which
    (together with the attribute names) IS the state_dict layout the HIP engine and published checkpoints rely on
    (checked against the layout captured from the reference: tests/golden/state_layout_*.json)."""
    _LISTS = ()
    _NAME = ""

    def __init__(self, input_size, width, n_layers, hparams, cross):
        super(_Stack, self).__init__()
        dtype = getattr(hparams, "compute_dtype", "fp32")
        heads, p_drop = hparams.n_attention_head, hparams.transformer_dropout_rate
        for name in self._LISTS:
            setattr(self, name, nn.ModuleList())
        self.pe_scale = nn.Parameter(torch.tensor(1.0))           # learnable scale of the sinusoid table
        self.dropout = nn.Dropout(p_drop)
        for i in range(n_layers):
            w_in = input_size if i == 0 else width                 # layer 0 runs at the width it is fed (decoder: memory width)
            self.attn_layer_norms.append(HipLayerNorm(w_in, eps=1e-6))
            self.self_attentions.append(MultiheadAttention(w_in, w_in, True, heads, p_drop, dtype))
            if cross:
                self.encdec_layer_norms.append(HipLayerNorm(w_in, eps=1e-6))
                self.encdec_attentions.append(MultiheadAttention(width, width, False, heads, p_drop, dtype))
            self.ffn_layer_norms.append(HipLayerNorm(width, eps=1e-6))
            self.ffn_layers.append(FFNLayer(width, 4 * width, width, p_drop, dtype))
        self.output_layer_norm = HipLayerNorm(width, eps=1e-6)


class TransformerEncoder(_Stack):
    _LISTS = ("self_attentions", "attn_layer_norms", "ffn_layers", "ffn_layer_norms")

    def __init__(self, input_size, hparams):
        super(TransformerEncoder, self).__init__(input_size, hparams.encoder_hidden, hparams.n_encoder_layer, hparams, cross=False)

    def prepare_inputs(self, inputs, input_lengths):
        """Zero padded positions, add the scaled sinusoid table, dropout; key-padding bias (modules.py:50-57)."""
        mask = torch.arange(inputs.shape[1], device=input_lengths.device)[None, :] < input_lengths[:, None]
        bias = attention_bias(mask, "masking")
        pe = get_sinusoid_encoding_table(inputs.shape[1], inputs.shape[2]).to(inputs.device)
        x = inputs * mask.unsqueeze(-1).to(inputs.device) + pe * self.pe_scale
        return _dropout(self, x, self.dropout.p), bias

    def forward(self, inputs, input_lengths):
        """Pre-LayerNorm stack: x += drop(self_attn(LN(x))); x += drop(ffn(LN(x))); output LayerNorm (modules.py:59-70)."""
        x, bias = self.prepare_inputs(inputs, input_lengths)
        p = self.dropout.p
        for i in range(len(self.self_attentions)):
            y = self.self_attentions[i](self.attn_layer_norms[i](x), None, bias)["outputs"]
            x = x + _dropout(self, y, p)
            y = self.ffn_layers[i](self.ffn_layer_norms[i](x))
            x = x + _dropout(self, y, p)
        return self.output_layer_norm(x)


class TransformerDecoder(_Stack):
    _LISTS = ("self_attentions", "attn_layer_norms", "encdec_attentions", "encdec_layer_norms", "ffn_layers", "ffn_layer_norms")

    def __init__(self, input_size, hparams):
        super(TransformerDecoder, self).__init__(input_size, hparams.decoder_hidden, hparams.n_decoder_layer, hparams, cross=True)

    def prepare_inputs(self, inputs, targets, input_lengths, target_lengths):
        """Memory mask bias, causal bias, targets zeroed past their length, shifted right by one frame, plus the scaled
        sinusoid table, dropout (modules.py:107-121)."""
        mask = torch.arange(inputs.shape[1], device=input_lengths.device)[None, :] < input_lengths[:, None]
        enc_bias = attention_bias(mask, "masking")
        dec_bias = attention_bias(targets.shape[1], "causal")
        t = impute(targets, target_lengths.to(targets.device))
        t = torch.cat([torch.zeros([t.shape[0], 1, t.shape[2]], device=t.device, dtype=t.dtype), t], dim=1)[:, :-1]
        pe = get_sinusoid_encoding_table(t.shape[1], t.shape[2]).to(t.device)
        t = t + pe * self.pe_scale
        return inputs, _dropout(self, t, self.dropout.p), dec_bias, enc_bias

    def forward(self, inputs, targets, input_lengths, target_lengths):
        """-> (outputs zeroed past target_lengths, {'self': [...], 'encdec': [...]} alignments per layer)  (modules.py:123-145)."""
        memory, x, query_bias, memory_bias = self.prepare_inputs(inputs, targets, input_lengths, target_lengths)
        p = self.dropout.p
        attn_align, encdec_align = [], []
        for i in range(len(self.self_attentions)):
            y = self.self_attentions[i](self.attn_layer_norms[i](x), None, query_bias)
            attn_align.append(y["align"])
            x = x + _dropout(self, y["outputs"], p)
            y = self.encdec_attentions[i](self.encdec_layer_norms[i](x), memory, memory_bias)
            encdec_align.append(y["align"])
            x = x + _dropout(self, y["outputs"], p)
            y = self.ffn_layers[i](self.ffn_layer_norms[i](x))
            x = x + _dropout(self, y, p)
        outputs = impute(self.output_layer_norm(x), target_lengths.to(x.device))
        return outputs, {"self": attn_align, "encdec": encdec_align}
