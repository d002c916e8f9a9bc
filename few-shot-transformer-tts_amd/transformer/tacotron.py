"""Tacotron wrapper (encoder / decoder / postnet), loss, init and LR schedule with the reference's API
(transformer/tacotron.py:1-179).  forward/backward of every segment run in libb2s_hip (hand-written HIP for
gfx950); this file only holds parameters under the reference's state_dict names and wires autograd."""
import os
import weakref

import torch
from torch import nn

from b2s_hip.engine import HipEngine, EncoderFn, DecoderFn, EncDecFn, PostnetFn, LossFn, _i32, DTYPES
from b2s_hip import ops
from transformer.attention import HipLinear
from transformer.modules import TransformerEncoder, TransformerDecoder
from transformer.common import impute, mask_reduce, truncated_normal, variance_scaling_initializer  # noqa: F401


class _Segment(nn.Module):
    """A model segment executed by the engine of the Tacotron that owns it."""
    _prefix = ""

    def _engine(self):
        root = self.__dict__.get("_root_ref")
        root = root() if root is not None else None
        if root is None:
            raise RuntimeError("%s must be part of a Tacotron model (it executes inside libb2s_hip)" % type(self).__name__)
        return root.engine()

    def _params(self):
        names, params = [], []
        for n, p in self.named_parameters():
            names.append(n)
            params.append(p)
        return names, params


class Encoder(_Segment):
    _prefix = "encoder."

    def __init__(self, hparams):
        super(Encoder, self).__init__()
        self.hparams = hparams
        self.embed = nn.Embedding(hparams.vocab_size, hparams.embed_size)       # parameter holder (gather runs in HIP)
        if hparams.multi_speaker:
            self.speaker_embed = nn.Embedding(hparams.max_num_speaker, hparams.speaker_embedding_size)
            self.speaker_layer = HipLinear(hparams.speaker_embedding_size, hparams.speaker_embedding_size)
        if hparams.multi_lingual:
            self.language_embed = HipLinear(hparams.max_num_language, hparams.language_embedding_size, bias=False)
            self.language_layer = HipLinear(hparams.language_embedding_size, hparams.language_embedding_size)
        self.encoder = TransformerEncoder(hparams.embed_size, hparams)

    def forward(self, inputs, input_lengths, input_spk_ids=None, input_language_vecs=None):
        """-> [B, S, encoder_hidden (+speaker) (+language)]  (tacotron.py:33-44)."""
        eng = self._engine()
        names, params = self._params()
        return EncoderFn.apply(eng, self._prefix, names, inputs, _i32(input_lengths), input_spk_ids, input_language_vecs,
                               self.training, *params)


class DecoderPrenet(nn.Module):
    def __init__(self, in_size, hidden_size, out_size, dropout_rate, compute_dtype="fp32"):
        super(DecoderPrenet, self).__init__()
        self.dense0 = HipLinear(in_size, hidden_size)
        self.dense1 = HipLinear(hidden_size, hidden_size)
        self.dense_final = HipLinear(hidden_size, out_size, bias=False)
        self.dropout = nn.Dropout(dropout_rate)
        self.compute_dtype = compute_dtype

    def forward(self, x):
        """Stand-alone prenet: two Linear + ReLU + dropout layers, then a bias-free Linear (tacotron.py:55-65).  Inside Decoder the
        dropout sits in the GEMM epilogues; here it is an element-wise mask from the same counter RNG."""
        from transformer.modules import _dropout
        for m in (self.dense0, self.dense1, self.dense_final):
            m.compute_dtype = self.compute_dtype
        p = self.dropout.p
        h = _dropout(self, self.dense0(x, relu=True), p)
        h = _dropout(self, self.dense1(h, relu=True), p)
        return self.dense_final(h)


class Postnet(_Segment):
    _prefix = "postnet."

    def __init__(self, hparams):
        super(Postnet, self).__init__()
        self.conv_layers = nn.ModuleList()
        self.batchnorm_layers = nn.ModuleList()
        self.dropout = nn.Dropout(hparams.decoder_dropout_rate)
        hidden = hparams.postnet_hidden
        for i in range(hparams.n_postnet_layer):
            in_size = hparams.num_mels if i == 0 else hidden
            out_size = hparams.num_mels if i == hparams.n_postnet_layer - 1 else hidden
            self.conv_layers.append(nn.Conv1d(in_size, out_size, 5, stride=1, padding=2, bias=False))   # holders
            self.batchnorm_layers.append(nn.BatchNorm1d(out_size))

    def forward(self, inputs, input_lengths, _fuse_add=False):
        """[B,T,M] -> residual [B,T,M] (tacotron.py:81-90): 5x {mask, conv k5 (implicit GEMM), BatchNorm, tanh, dropout}."""
        eng = self._engine()
        names, params = self._params()
        return PostnetFn.apply(eng, self._prefix, names, inputs, _i32(input_lengths), _fuse_add, self.training, *params)


class LazyAlignments(dict):
    """{'self': [L tensors], 'encdec': [L tensors]} of shape [B,H,Lk,Lq] (modules.py:145), materialised from the
    softmax weights held by the decoder context only when a key is read (training never reads them)."""

    def __init__(self, eng, ctx, n_layers, B, H, T, S):
        super(LazyAlignments, self).__init__()
        self._a = (eng, ctx, n_layers, B, H, T, S)

    def _fill(self, key):
        if not dict.__contains__(self, key):
            eng, ctx, n, B, H, T, S = self._a
            which, lk = (0, T) if key == "self" else (1, S)
            dict.__setitem__(self, key, [eng.decoder_alignment(ctx, which, i, B, H, lk, T) for i in range(n)])

    def __getitem__(self, key):
        if key in ("self", "encdec"):
            self._fill(key)
        return dict.__getitem__(self, key)

    def __contains__(self, key):
        return key in ("self", "encdec")

    def keys(self):
        return ["self", "encdec"]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return 2

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]


class Decoder(_Segment):
    _prefix = "decoder."

    def __init__(self, hparams):
        super(Decoder, self).__init__()
        in_size = hparams.encoder_hidden
        if hparams.multi_speaker:
            in_size += hparams.speaker_embedding_size
        if hparams.multi_lingual:
            in_size += hparams.language_embedding_size
        cd = getattr(hparams, "compute_dtype", "fp32")
        self.prenet = DecoderPrenet(hparams.num_mels, hparams.prenet_hidden, hparams.decoder_hidden,
                                    hparams.decoder_dropout_rate, cd)
        self.decoder = TransformerDecoder(in_size, hparams)
        self.mel_net = HipLinear(hparams.decoder_hidden, hparams.num_mels, bias=False)
        self.stop_net = HipLinear(hparams.decoder_hidden, 1)
        self._n_layers = hparams.n_decoder_layer
        self._heads = hparams.n_attention_head

    def forward(self, encoder_outputs, input_lengths, targets, target_lengths, leave_one=False):
        """-> (mels [B,T,M], stop_logits [B,T], {'self','encdec'} alignments)  (tacotron.py:107-116).
        leave_one zeroes the last prenet row, which the shift-right then drops (modules.py:115-116): a no-op."""
        eng = self._engine()
        names, params = self._params()
        holder = []
        mels, stop, guided = DecoderFn.apply(eng, self._prefix, names, encoder_outputs, _i32(input_lengths), targets,
                                             _i32(target_lengths), self.training, holder, *params)
        B, T = targets.shape[0], targets.shape[1]
        align = LazyAlignments(eng, holder[0], self._n_layers, B, self._heads, T, encoder_outputs.shape[1])
        align.guided_loss = guided           # extension: weight * guided-attention loss (None when the weight is 0)
        return mels, stop, align


class Tacotron(nn.Module):
    def __init__(self, hparams):
        super(Tacotron, self).__init__()
        self.encoder = Encoder(hparams)
        self.decoder = Decoder(hparams)
        self.postnet = Postnet(hparams)
        self.__dict__["_hp"] = hparams
        self.__dict__["_engine"] = None
        if getattr(hparams, "freeze_encoder", False):       # extension: few-shot fine-tuning with a frozen encoder
            for p in self.encoder.parameters():
                p.requires_grad_(False)
        for seg in (self.encoder, self.decoder, self.postnet):
            seg.__dict__["_root_ref"] = weakref.ref(self)

    def engine(self):
        """The HIP engine bound to this module tree (created on first use; raises if libb2s_hip.so is missing)."""
        if self.__dict__["_engine"] is None:
            self.__dict__["_engine"] = HipEngine(self, self.__dict__["_hp"])
        return self.__dict__["_engine"]

    def forward(self, inputs, input_lengths, mel_targets, target_lengths, input_spk_ids, input_language_vecs, **kwargs):
        """tacotron.py:126-133: kwargs are the dataloader batch dict (unknown keys such as `names` are ignored)."""
        if getattr(self, "_is_replica", False):
            # nn.DataParallel over several devices (train.py:126-127 without --ddp) replicates the module per device and per step; the HIP
            # engine is bound to ONE device's parameters, gradient buffer and streams.  One process per GPU is the supported layout.
            from b2s_hip.lib import B2SError
            raise B2SError("nn.DataParallel over more than one device is not supported by the HIP engine (it is bound to one device); run one "
                           "process per GPU -- torch.distributed.run + DistributedDataParallel (train.py --ddp) or b2s_hip.trainer.HipTrainer -- "
                           "or restrict the wrapper: nn.DataParallel(m, device_ids=[0])")
        if inputs.is_cuda and os.environ.get("B2S_DROPIN_OVERLAP", "1") != "0":
            # encoder + decoder as ONE autograd node: the encoder runs beside the decoder's first kernels (forward) and beside the end of the decoder
            # backward, as under HipTrainer; both streams are joined inside the node (b2s_hip/engine.py: EncDecFn).  B2S_DROPIN_OVERLAP=0: two nodes
            eng = self.engine()
            en, ep = self.encoder._params()
            dn, dp = self.decoder._params()
            holder = []
            mel_bef, stop_logits, guided = EncDecFn.apply(eng, len(ep), en, dn, inputs, _i32(input_lengths), input_spk_ids, input_language_vecs,
                                                          mel_targets, _i32(target_lengths), self.training, holder, *(ep + dp))
            alignments = LazyAlignments(eng, holder[0], self.decoder._n_layers, mel_targets.shape[0], self.decoder._heads, mel_targets.shape[1], holder[1][1])
            alignments.guided_loss = guided
        else:
            enc_outputs = self.encoder(inputs, input_lengths, input_spk_ids, input_language_vecs)
            mel_bef, stop_logits, alignments = self.decoder(enc_outputs, input_lengths, mel_targets, target_lengths)
        mel_aft = self.postnet(mel_bef, target_lengths, _fuse_add=True)          # mel_bef + postnet(mel_bef), fused
        outputs = {'mel_bef': mel_bef, 'mel_aft': mel_aft, 'stop_logits': stop_logits, 'alignments': alignments}
        if getattr(alignments, "guided_loss", None) is not None:
            outputs['guided_attention_loss'] = alignments.guided_loss
        return outputs


def compute_loss(model, mel_targets, target_lengths, outputs, hparams):
    """tacotron.py:136-158: masked MSE before/after the postnet, stop-token BCE (pos_weight 5), L2 over the
    name-filtered weights; one fused HIP reduction + fused gradient kernels."""
    root = model.module if hasattr(model, "module") else model
    eng = root.engine()
    vals, per = LossFn.apply(eng, outputs['mel_bef'], outputs['mel_aft'], outputs['stop_logits'], mel_targets,
                             _i32(target_lengths))
    res = {'loss': vals[0], 'bef_loss': vals[1], 'aft_loss': vals[2], 'aft_losses': per,
           'mse_loss': vals[3], 'l2': vals[4], 'stop_loss': vals[5]}
    if outputs.get('guided_attention_loss') is not None:     # extension (hparams.guided_attention_weight > 0)
        res['ga_loss'] = outputs['guided_attention_loss']
        res['loss'] = res['loss'] + res['ga_loss']
    return res


# name rule -> initialiser, first match wins (tacotron.py:161-173): byte embedding ~ N(0, 1); speaker / language tables ~
# truncated N(0, 0.5); every other weight matrix / conv kernel variance-scaled; biases zero; LayerNorm / BatchNorm affine
# parameters keep torch's defaults
_INIT_RULES = (
    (lambda n: n == "encoder.embed.weight", lambda t: torch.randn(t.shape)),
    (lambda n: n in ("encoder.speaker_embed.weight", "encoder.language_embed.weight"), lambda t: truncated_normal(t, 0, 0.5)),
    (lambda n: "weight" in n and "layer_norm" not in n and "batchnorm" not in n, variance_scaling_initializer),
    (lambda n: "bias" in n, torch.zeros_like),
)


def initialize_variables(model):
    """TF-style initialisation of a freshly constructed Tacotron, in place, once, on the host."""
    with torch.no_grad():
        for name, tensor in model.state_dict().items():
            for applies, make in _INIT_RULES:
                if applies(name):
                    tensor.copy_(make(tensor).to(tensor.device, tensor.dtype))
                    break


def learning_rate_schedule(global_step, hp):
    """LambdaLR multiplier (tacotron.py:176-179): 1 through the first warmup_steps, then an exponential decay that
    reaches lr_decay_rate after lr_decay_step further steps, floored at min_lr / max_lr."""
    past_warmup = max(global_step - hp.warmup_steps, 0)
    return max(hp.lr_decay_rate ** (past_warmup / hp.lr_decay_step), hp.min_lr / hp.max_lr)
