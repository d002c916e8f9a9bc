"""Transformer encoder/decoder stacks with the reference's module surface and state_dict layout
(transformer/modules.py:1-145).  The stacks execute inside libb2s_hip (csrc/engine.hip); these classes
hold the parameters under the reference's names and dispatch to the engine of the owning Tacotron."""
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
        """Linear -> ReLU -> dropout -> Linear (modules.py:15-20).  Stand-alone use supports dropout rate 0 / eval only;
        inside the model the fused engine path applies the in-kernel dropout."""
        if self.training and self.dropout.p > 0:
            raise NotImplementedError("stand-alone FFNLayer with dropout: run it inside Tacotron (engine path)")
        self.input_layer.compute_dtype = self.output_layer.compute_dtype = self.compute_dtype
        return self.output_layer(self.input_layer(inputs, relu=True))


def _no_standalone(name):
    raise NotImplementedError(
        "%s runs inside libb2s_hip as part of Encoder/Decoder (csrc/engine.hip); call model.encoder / model.decoder" % name)


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

    def forward(self, inputs, input_lengths):
        _no_standalone("TransformerEncoder")


class TransformerDecoder(_Stack):
    _LISTS = ("self_attentions", "attn_layer_norms", "encdec_attentions", "encdec_layer_norms", "ffn_layers", "ffn_layer_norms")

    def __init__(self, input_size, hparams):
        super(TransformerDecoder, self).__init__(input_size, hparams.decoder_hidden, hparams.n_decoder_layer, hparams, cross=True)

    def forward(self, inputs, targets, input_lengths, target_lengths):
        _no_standalone("TransformerDecoder")
