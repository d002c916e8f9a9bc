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


class TransformerEncoder(nn.Module):
    def __init__(self, input_size, hparams):
        super(TransformerEncoder, self).__init__()
        cd = getattr(hparams, "compute_dtype", "fp32")
        self.self_attentions = nn.ModuleList()
        self.attn_layer_norms = nn.ModuleList()
        self.ffn_layers = nn.ModuleList()
        self.ffn_layer_norms = nn.ModuleList()
        self.pe_scale = nn.Parameter(torch.tensor(1.0))
        self.dropout = nn.Dropout(hparams.transformer_dropout_rate)
        hidden_size = hparams.encoder_hidden
        for layer in range(hparams.n_encoder_layer):
            in_size = input_size if layer == 0 else hidden_size
            self.attn_layer_norms.append(HipLayerNorm(in_size, eps=1e-6))
            self.self_attentions.append(MultiheadAttention(in_size, in_size, True, hparams.n_attention_head,
                                                           hparams.transformer_dropout_rate, cd))
            self.ffn_layer_norms.append(HipLayerNorm(hidden_size, eps=1e-6))
            self.ffn_layers.append(FFNLayer(hidden_size, hidden_size * 4, hidden_size, hparams.transformer_dropout_rate, cd))
        self.output_layer_norm = HipLayerNorm(hidden_size, eps=1e-6)

    def forward(self, inputs, input_lengths):
        _no_standalone("TransformerEncoder")


class TransformerDecoder(nn.Module):
    def __init__(self, input_size, hparams):
        super(TransformerDecoder, self).__init__()
        cd = getattr(hparams, "compute_dtype", "fp32")
        self.self_attentions = nn.ModuleList()
        self.attn_layer_norms = nn.ModuleList()
        self.encdec_attentions = nn.ModuleList()
        self.encdec_layer_norms = nn.ModuleList()
        self.ffn_layers = nn.ModuleList()
        self.ffn_layer_norms = nn.ModuleList()
        self.pe_scale = nn.Parameter(torch.tensor(1.0))
        self.dropout = nn.Dropout(hparams.transformer_dropout_rate)
        hidden_size = hparams.decoder_hidden
        for layer in range(hparams.n_decoder_layer):
            in_size = input_size if layer == 0 else hidden_size
            self.attn_layer_norms.append(HipLayerNorm(in_size, eps=1e-6))
            self.self_attentions.append(MultiheadAttention(in_size, in_size, True, hparams.n_attention_head,
                                                           hparams.transformer_dropout_rate, cd))
            self.encdec_layer_norms.append(HipLayerNorm(in_size, eps=1e-6))
            self.encdec_attentions.append(MultiheadAttention(hidden_size, hidden_size, False, hparams.n_attention_head,
                                                             hparams.transformer_dropout_rate, cd))
            self.ffn_layer_norms.append(HipLayerNorm(hidden_size, eps=1e-6))
            self.ffn_layers.append(FFNLayer(hidden_size, hidden_size * 4, hidden_size, hparams.transformer_dropout_rate, cd))
        self.output_layer_norm = HipLayerNorm(hidden_size, eps=1e-6)

    def forward(self, inputs, targets, input_lengths, target_lengths):
        _no_standalone("TransformerDecoder")
