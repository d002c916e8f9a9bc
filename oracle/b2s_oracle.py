"""CPU oracle: functional fp32 restatement of the reference's Transformer-TTS hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Written from the math of the
reference, as pure functions over a {name: tensor} parameter dict `P` whose names equal
the reference's state_dict keys.  Each function cites the reference lines it follows.
Gradients come from torch autograd over these functions.

Parity status: PINNED against reference-generated goldens (tests/golden/*.npz, made by
tests/golden/make_goldens.py) for everything except dropout-on behaviour (the reference draws
from torch's global generator; its streams cannot be reproduced) and initialize_variables
(distributional only).  Dropout ON is checked the other way round: under `device_masks(...)`
every F.dropout call below takes the mask the HIP engine's counter RNG draws at that site
(oracle/rng.py: DeviceMasks), so forward, losses and every gradient of the dropout-on step
the benchmark times can be compared element for element (tests/test_gpu_dropout_parity.py).
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

NEG_INF = -1e20  # common.py:32


# --------------------------------------------------------------------------- helpers
def sinusoid_table(length, channels, min_timescale=1.0, max_timescale=1.0e4):
    """common.py:4-29 -- float64 NumPy table [sin | cos], zero column if channels is odd."""
    pos = np.arange(length, dtype=np.float64)
    nts = channels // 2
    inc = math.log(max_timescale / min_timescale) / (nts - 1)
    inv = min_timescale * np.exp(np.arange(nts, dtype=np.float64) * -inc)
    st = pos[:, None] * inv[None, :]
    sig = np.concatenate([np.sin(st), np.cos(st)], axis=1)
    if channels % 2:
        sig = np.pad(sig, [[0, 0], [0, 1]])
    return torch.from_numpy(sig.astype(np.float32))


def length_mask(lengths, max_len):
    """[B, L] bool, position < length (common.py:64, modules.py:50,109)."""
    return torch.arange(max_len)[None, :] < lengths[:, None]


def attention_bias_masking(mask):
    """common.py:44-46 -- (1 - mask) * -1e20, shape [B,1,1,L]."""
    return ((1.0 - mask.float()) * NEG_INF)[:, None, None, :]


def attention_bias_causal(n):
    """common.py:41-43 -- strict upper triangle * -1e20, shape [1,1,n,n]."""
    return (torch.triu(torch.ones(n, n), diagonal=1) * NEG_INF)[None, None]


def impute(x, lengths, channels_last=True):
    """common.py:51-70 -- zero every time step >= length."""
    L = x.shape[1] if channels_last else x.shape[-1]
    m = length_mask(lengths, L)
    while m.dim() < x.dim():
        m = m.unsqueeze(-1) if channels_last else m.unsqueeze(1)
    return x * m


def mask_reduce(loss, lengths, per_sample=False):
    """common.py:73-87."""
    if per_sample:
        return impute(loss, lengths).sum(-1) / lengths
    return impute(loss, lengths).sum() / lengths.sum()


_MASKS = None      # oracle/rng.py: DeviceMasks while a `device_masks` block is active, else None (torch's own generator)


class device_masks:
    """with device_masks(src): every dropout call below uses src.keep(site, layer, shape, p, frame_offset) -- the mask the HIP
    engine applies at that site (same seed) -- instead of drawing from torch's generator."""

    def __init__(self, src):
        self.src = src

    def __enter__(self):
        global _MASKS
        self.prev, _MASKS = _MASKS, self.src
        return self.src

    def __exit__(self, *a):
        global _MASKS
        _MASKS = self.prev


def _drop(x, p, train, site=None, layer=0, channels_first=False, frame_offset=0):
    """F.dropout(x, p) (modules.py:18 etc.).  site / layer name the reference call (csrc/drop_sites.h) for device_masks;
    channels_first: x is [B, C, T] where the engine holds [B, T, C]; frame_offset: decode loop -- row r of x was produced by the
    engine in frame r + frame_offset."""
    if not (train and p > 0.0):
        return x
    if _MASKS is None:
        return F.dropout(x, p, training=True)
    shape = tuple(x.shape)
    if channels_first:
        shape = (shape[0], shape[2], shape[1])
    keep = torch.from_numpy(_MASKS.keep(site, layer, shape, p, frame_offset))
    if channels_first:
        keep = keep.permute(0, 2, 1)
    return x * (keep.to(x.dtype) * float(np.float32(1.0) / (np.float32(1.0) - np.float32(p))))


def _ln(x, P, prefix):
    return F.layer_norm(x, (x.shape[-1],), P[prefix + ".weight"], P[prefix + ".bias"], eps=1e-6)


# --------------------------------------------------------------------------- attention
def multihead_attention(P, prefix, queries, memories, bias, num_heads, p_drop=0.0, train=False, site=None, layer=0):
    """attention.py:94-122 (A1-A4).  Returns (outputs [B,Lq,C], align [B,H,Lk,Lq])."""
    C = queries.shape[-1]
    if memories is None:                      # attention.py:62-64: fused q|k|v
        qkv = queries @ P[prefix + ".qkv_transform.weight"].t()
        q, k, v = qkv.split([C, C, C], dim=-1)
    else:                                     # attention.py:66-68: q ; k|v from memory
        q = queries @ P[prefix + ".q_transform.weight"].t()
        kv = memories @ P[prefix + ".kv_transform.weight"].t()
        k, v = kv.split([C, C], dim=-1)
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    dh = C // num_heads

    def heads(t, L):                          # attention.py:6-15
        return t.reshape(B, L, num_heads, dh).permute(0, 2, 1, 3)
    q, k, v = heads(q, Lq), heads(k, Lk), heads(v, Lk)
    q = q * dh ** -0.5                        # attention.py:113-114
    logits = q @ k.transpose(2, 3)            # attention.py:83
    if bias is not None:
        logits = logits + bias
    w = F.softmax(logits, dim=-1)
    align = w.permute(0, 1, 3, 2)             # attention.py:88 (pre-dropout)
    w = _drop(w, p_drop, train, site, layer)
    ctx = (w @ v).permute(0, 2, 1, 3).reshape(B, Lq, C)   # attention.py:18-26
    return ctx @ P[prefix + ".output_transform.weight"].t(), align


def ffn(P, prefix, x, p_drop=0.0, train=False, site=None, layer=0):
    """modules.py:15-20 (A5)."""
    h = F.relu(x @ P[prefix + ".input_layer.weight"].t())
    h = _drop(h, p_drop, train, site, layer)
    return h @ P[prefix + ".output_layer.weight"].t()


# --------------------------------------------------------------------------- encoder
def transformer_encoder(P, cfg, x, input_lengths, train=False, prefix="encoder.encoder"):
    """modules.py:49-69 (A6)."""
    p = cfg.transformer_dropout_rate
    mask = length_mask(input_lengths, x.shape[1])
    x = x * mask.unsqueeze(-1)
    bias = attention_bias_masking(mask)
    x = x + sinusoid_table(x.shape[1], x.shape[2]) * P[prefix + ".pe_scale"]
    x = _drop(x, p, train, "encoder.embed")
    for i in range(cfg.n_encoder_layer):
        y, _ = multihead_attention(P, "%s.self_attentions.%d" % (prefix, i),
                                   _ln(x, P, "%s.attn_layer_norms.%d" % (prefix, i)), None, bias,
                                   cfg.n_attention_head, p, train, "encoder.attn", i)
        x = x + _drop(y, p, train, "encoder.attn_res", i)
        y = ffn(P, "%s.ffn_layers.%d" % (prefix, i), _ln(x, P, "%s.ffn_layer_norms.%d" % (prefix, i)), p, train,
                "encoder.ffn_hidden", i)
        x = x + _drop(y, p, train, "encoder.ffn_res", i)
    return _ln(x, P, prefix + ".output_layer_norm")


def encoder_forward(P, cfg, inputs, input_lengths, input_spk_ids=None, input_language_vecs=None, train=False):
    """tacotron.py:33-44 (A9).  -> [B,S,encoder_hidden(+spk)(+lang)]."""
    x = F.embedding(inputs, P["encoder.embed.weight"])
    out = transformer_encoder(P, cfg, x, input_lengths, train)
    S = inputs.shape[1]
    if cfg.multi_speaker:                     # tacotron.py:27-31
        e = F.embedding(input_spk_ids, P["encoder.speaker_embed.weight"])
        e = F.softsign(e @ P["encoder.speaker_layer.weight"].t() + P["encoder.speaker_layer.bias"])
        out = torch.cat([out, e.unsqueeze(1).repeat(1, S, 1)], dim=-1)
    if cfg.multi_lingual:                     # tacotron.py:21-25
        e = input_language_vecs @ P["encoder.language_embed.weight"].t()
        e = F.softsign(e @ P["encoder.language_layer.weight"].t() + P["encoder.language_layer.bias"])
        out = torch.cat([out, e.unsqueeze(1).repeat(1, S, 1)], dim=-1)
    return out


# --------------------------------------------------------------------------- decoder
def prenet(P, cfg, x, train=False, prefix="decoder.prenet"):
    """tacotron.py:55-65 (A10)."""
    p = cfg.decoder_dropout_rate
    # (decode loop: row r of the loop's decoder input is the previous frame, the engine computes its prenet in frame r + 1)
    x = _drop(F.relu(x @ P[prefix + ".dense0.weight"].t() + P[prefix + ".dense0.bias"]), p, train, "decoder.prenet0", frame_offset=1)
    x = _drop(F.relu(x @ P[prefix + ".dense1.weight"].t() + P[prefix + ".dense1.bias"]), p, train, "decoder.prenet1", frame_offset=1)
    return x @ P[prefix + ".dense_final.weight"].t()


def transformer_decoder(P, cfg, memory, targets, input_lengths, target_lengths, train=False,
                        prefix="decoder.decoder"):
    """modules.py:108-145 (A7).  -> (outputs [B,T,Dd], {'self':[L], 'encdec':[L]})."""
    p = cfg.transformer_dropout_rate
    enc_bias = attention_bias_masking(length_mask(input_lengths, memory.shape[1]))
    dec_bias = attention_bias_causal(targets.shape[1])
    x = impute(targets, target_lengths)
    x = torch.cat([torch.zeros_like(x[:, :1]), x], dim=1)[:, :-1]       # shift right (modules.py:115-116)
    x = x + sinusoid_table(x.shape[1], x.shape[2]) * P[prefix + ".pe_scale"]
    x = _drop(x, p, train, "decoder.embed")
    a_self, a_cross = [], []
    for i in range(cfg.n_decoder_layer):
        y, al = multihead_attention(P, "%s.self_attentions.%d" % (prefix, i),
                                    _ln(x, P, "%s.attn_layer_norms.%d" % (prefix, i)), None, dec_bias,
                                    cfg.n_attention_head, p, train, "decoder.self_attn", i)
        a_self.append(al)
        x = x + _drop(y, p, train, "decoder.self_res", i)
        y, al = multihead_attention(P, "%s.encdec_attentions.%d" % (prefix, i),
                                    _ln(x, P, "%s.encdec_layer_norms.%d" % (prefix, i)), memory, enc_bias,
                                    cfg.n_attention_head, p, train, "decoder.cross_attn", i)
        a_cross.append(al)
        x = x + _drop(y, p, train, "decoder.cross_res", i)
        y = ffn(P, "%s.ffn_layers.%d" % (prefix, i), _ln(x, P, "%s.ffn_layer_norms.%d" % (prefix, i)), p, train,
                "decoder.ffn_hidden", i)
        x = x + _drop(y, p, train, "decoder.ffn_res", i)
    out = impute(_ln(x, P, prefix + ".output_layer_norm"), target_lengths)
    return out, {"self": a_self, "encdec": a_cross}


def decoder_forward(P, cfg, encoder_outputs, input_lengths, targets, target_lengths, leave_one=False,
                    train=False):
    """tacotron.py:107-116 (A12).  -> (mels [B,T,M], stop_logits [B,T], align dict)."""
    d = prenet(P, cfg, targets, train)
    if leave_one:                              # tacotron.py:109-110
        d = torch.cat([d[:, :-1], d[:, -1:] * 0], dim=1)
    out, align = transformer_decoder(P, cfg, encoder_outputs, d, input_lengths, target_lengths, train)
    mels = impute(out @ P["decoder.mel_net.weight"].t(), target_lengths)
    stop = (out.detach() @ P["decoder.stop_net.weight"].t() + P["decoder.stop_net.bias"]).squeeze(-1)
    return mels, impute(stop, target_lengths), align


# --------------------------------------------------------------------------- postnet
def postnet_forward(P, cfg, inputs, input_lengths, train=False, bn_state=None):
    """tacotron.py:81-90 (A11).  inputs [B,T,M] -> [B,T,M].

    train=True uses batch statistics over all B*T positions (padding included) and, if
    bn_state (a dict) is given, writes the updated running_mean / running_var /
    num_batches_tracked into it (momentum 0.1, unbiased running variance).
    """
    p = cfg.decoder_dropout_rate
    x = inputs.transpose(1, 2)
    n = cfg.n_postnet_layer
    for i in range(n):
        x = impute(x, input_lengths, channels_last=False)
        x = F.conv1d(x, P["postnet.conv_layers.%d.weight" % i], None, 1, 2)
        q = "postnet.batchnorm_layers.%d." % i
        if train:
            mean = x.mean(dim=(0, 2))
            var = x.var(dim=(0, 2), unbiased=False)
            if bn_state is not None:
                cnt = x.shape[0] * x.shape[2]
                bn_state[q + "running_mean"] = 0.9 * P[q + "running_mean"] + 0.1 * mean.detach()
                bn_state[q + "running_var"] = 0.9 * P[q + "running_var"] + 0.1 * var.detach() * cnt / (cnt - 1)
                bn_state[q + "num_batches_tracked"] = P[q + "num_batches_tracked"] + 1
        else:
            mean, var = P[q + "running_mean"], P[q + "running_var"]
        x = (x - mean[None, :, None]) / torch.sqrt(var[None, :, None] + 1e-5)
        x = x * P[q + "weight"][None, :, None] + P[q + "bias"][None, :, None]
        if i != n - 1:
            x = torch.tanh(x)
        x = _drop(x, p, train, "postnet.conv", i, channels_first=True)
    return x.transpose(2, 1)


# --------------------------------------------------------------------------- model / loss
def tacotron_forward(P, cfg, batch, train=True, decoder_train=None, bn_state=None):
    """tacotron.py:126-133 (A13)."""
    dtrain = train if decoder_train is None else decoder_train
    enc = encoder_forward(P, cfg, batch["inputs"], batch["input_lengths"], batch.get("input_spk_ids"),
                          batch.get("input_language_vecs"), train)
    mel_bef, stop, align = decoder_forward(P, cfg, enc, batch["input_lengths"], batch["mel_targets"],
                                           batch["target_lengths"], False, dtrain)
    res = postnet_forward(P, cfg, mel_bef, batch["target_lengths"], train, bn_state)
    return {"mel_bef": mel_bef, "mel_aft": mel_bef + res, "stop_logits": stop, "alignments": align}


def l2_member(name):
    """tacotron.py:144-146 -- which parameters enter the L2 term."""
    return ("weight" in name and "layer_norm" not in name and "batchnorm" not in name
            and "encoder.speaker_embed" not in name and "encoder.embed" not in name)


_NON_PARAM = ("running_mean", "running_var", "num_batches_tracked")


def is_parameter(name):
    return not name.endswith(_NON_PARAM)


def guided_attention_loss(encdec_aligns, input_lengths, target_lengths, sigma):
    """EXTENSION (absent upstream; parity unpinned by the reference).  Guided-attention loss of
    Tachibana et al. 2017 ("Efficiently trainable text-to-speech ...", eq. 3) applied to every head of every
    encoder-decoder attention layer, masked to the valid (frame, byte) rectangle of each utterance:
        W[n, t] = 1 - exp(-(n / N_b - t / T_b)^2 / (2 sigma^2)),   loss = mean over layers, heads and valid (b, n, t) of A * W
    encdec_aligns: list of [B, H, S, T] tensors (attention.py:81 `align`, softmax weights before dropout)."""
    A0 = encdec_aligns[0]
    B, H, S, T = A0.shape
    n = torch.arange(S, dtype=torch.float32)[None, :, None] / input_lengths[:, None, None].float()
    t = torch.arange(T, dtype=torch.float32)[None, None, :] / target_lengths[:, None, None].float()
    W = 1.0 - torch.exp(-((n - t) ** 2) / (2.0 * sigma * sigma))
    valid = (length_mask(input_lengths, S)[:, :, None] * length_mask(target_lengths, T)[:, None, :]).float()
    W = (W * valid)[:, None]
    denom = valid.sum() * H * len(encdec_aligns)
    return sum((A * W).sum() for A in encdec_aligns) / denom


def compute_loss(P, cfg, mel_targets, target_lengths, outputs, input_lengths=None):
    """tacotron.py:136-158 (A14); + the guided-attention extension when cfg.guided_attention_weight > 0."""
    bef = mask_reduce(((outputs["mel_bef"] - mel_targets) ** 2).mean(-1), target_lengths)
    aft_e = ((outputs["mel_aft"] - mel_targets) ** 2).mean(-1)
    aft_s = mask_reduce(aft_e, target_lengths, per_sample=True)
    aft = mask_reduce(aft_e, target_lengths)
    l2 = cfg.reg_weight * sum((p ** 2).sum() / 2 for n, p in P.items() if is_parameter(n) and l2_member(n))
    T = mel_targets.shape[1]
    stop_target = (torch.arange(T)[None, :] == target_lengths[:, None] - 1).float()
    ce = F.binary_cross_entropy_with_logits(outputs["stop_logits"], stop_target, reduction="none",
                                            pos_weight=torch.tensor([5.0]))
    ce = mask_reduce(ce, target_lengths)
    res = {"loss": bef + aft + l2 + ce, "bef_loss": bef, "aft_loss": aft, "aft_losses": aft_s,
           "mse_loss": (bef + aft) / 2, "l2": l2, "stop_loss": ce}
    gw = getattr(cfg, "guided_attention_weight", 0.0)
    if gw > 0:
        res["ga_loss"] = gw * guided_attention_loss(outputs["alignments"]["encdec"], input_lengths, target_lengths,
                                                    cfg.guided_attention_sigma)
        res["loss"] = res["loss"] + res["ga_loss"]
    return res


def learning_rate_schedule(global_step, cfg):
    """tacotron.py:176-179 (A15)."""
    step = max(global_step - cfg.warmup_steps, 0)
    return max(cfg.min_lr / cfg.max_lr, cfg.lr_decay_rate ** (step / cfg.lr_decay_step))


def adam_step(P, grads, state, step_index, cfg, beta1=0.9, beta2=0.999):
    """torch.optim.Adam(lr=max_lr, eps=adam_eps) under LambdaLR (train.py:130-131,188-189).

    step_index = number of optimizer steps already taken (0 for the first).  In place on P/state.
    """
    lr = cfg.max_lr * learning_rate_schedule(step_index, cfg)
    t = step_index + 1
    bc1, bc2 = 1 - beta1 ** t, 1 - beta2 ** t
    for n, g in grads.items():
        m = state.setdefault(n + ".m", torch.zeros_like(g))
        v = state.setdefault(n + ".v", torch.zeros_like(g))
        m.mul_(beta1).add_(g, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(cfg.adam_eps)
        P[n].data.addcdiv_(m, denom, value=-lr / bc1)


def to_torch_state(np_state, requires_grad=False):
    P = {}
    for n, a in np_state.items():
        t = torch.from_numpy(np.array(a))
        if requires_grad and is_parameter(n):
            t.requires_grad_(True)
        P[n] = t
    return P


def to_torch_batch(np_batch):
    return {k: (torch.from_numpy(np.asarray(v)) if not isinstance(v, list) else v) for k, v in np_batch.items()}


def train_step(P, cfg, batch, opt_state, step_index, train=True):
    """One reference training step (train.py:171-174,188-189): fwd, loss, bwd, Adam.  In place."""
    bn_state = {}
    out = tacotron_forward(P, cfg, batch, train=train, bn_state=bn_state)
    losses = compute_loss(P, cfg, batch["mel_targets"], batch["target_lengths"], out, batch["input_lengths"])
    # EXTENSION: frozen encoder (few-shot fine-tuning) = encoder parameters receive no update
    names = [n for n in P if is_parameter(n) and not (getattr(cfg, "freeze_encoder", False) and n.startswith("encoder."))]
    gl = torch.autograd.grad(losses["loss"], [P[n] for n in names], allow_unused=True)
    grads = {n: (g if g is not None else torch.zeros_like(P[n])) for n, g in zip(names, gl)}
    with torch.no_grad():
        adam_step(P, grads, opt_state, step_index, cfg)
        for k, v in bn_state.items():
            P[k] = v.detach()
    return out, losses, grads


# --------------------------------------------------------------------------- AR decode
def eval_batch(P, cfg, batch, decoder_train=False):
    """synthesize.py:17-72 (A17): the reference's cache-free autoregressive loop.

    Re-runs the full decoder over the growing prefix each step exactly as the reference does.
    decoder_train=False corresponds to dropout rates 0 / model.eval().
    """
    with torch.no_grad():
        B = batch["inputs"].shape[0]
        tl = torch.ones(B, dtype=torch.int32)
        finished = torch.zeros(B, dtype=torch.bool)
        mels = torch.zeros(B, 0, cfg.num_mels)
        enc = encoder_forward(P, cfg, batch["inputs"], batch["input_lengths"], batch.get("input_spk_ids"),
                              batch.get("input_language_vecs"), False)
        align = None
        while not bool(torch.all(finished)) and mels.shape[1] < cfg.max_generation_frames:
            dec_in = torch.cat([mels, torch.zeros(B, 1, cfg.num_mels)], dim=1)
            mel_bef, stop_logits, align = decoder_forward(P, cfg, enc, batch["input_lengths"], dec_in, tl,
                                                          leave_one=True, train=decoder_train)
            stop = stop_logits[:, -1] > 0
            mels = torch.cat([mels, mel_bef[:, -1:]], dim=1)
            finished = finished | stop
            tl = torch.where(finished, tl, tl + 1)
        mel_aft = mels + postnet_forward(P, cfg, mels, tl, train=False)
        return {"mel_pre": mels, "mel_aft": mel_aft, "alignments": align,
                "input_lengths": batch["input_lengths"], "generated_lengths": tl}
