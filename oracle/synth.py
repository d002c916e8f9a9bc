"""Deterministic synthetic weights and batches (TEST INFRASTRUCTURE ONLY).

Both the golden-fixture generator (which feeds them to the reference) and the tests
(which feed them to the oracle / the HIP path) draw weights and inputs from this module,
so fixtures need to store outputs only.  Everything is NumPy default_rng -- reproducible
on any machine without the reference.

param_shapes() restates the reference's state_dict layout (transformer/tacotron.py:8-123,
transformer/modules.py:23-106, transformer/attention.py:30-51); a CPU test checks it
against the (name, shape) list captured from the reference.
"""
import numpy as np


def param_shapes(cfg):
    """Ordered [(name, shape, kind)] for every state_dict entry of Tacotron(cfg).

    kind in {'embed','w','b','ln_w','ln_b','bn_w','bn_b','bn_rm','bn_rv','bn_nbt','scalar'}.
    Order = the reference's nn.Module registration order.
    """
    out = []
    De, Dh = cfg.embed_size, cfg.encoder_hidden
    Dd = cfg.decoder_hidden
    mem = Dh + (cfg.speaker_embedding_size if cfg.multi_speaker else 0) + \
        (cfg.language_embedding_size if cfg.multi_lingual else 0)

    def ln(prefix, n):
        out.append((prefix + ".weight", (n,), "ln_w"))
        out.append((prefix + ".bias", (n,), "ln_b"))

    # Encoder (tacotron.py:9-19)
    out.append(("encoder.embed.weight", (cfg.vocab_size, De), "embed"))
    if cfg.multi_speaker:
        out.append(("encoder.speaker_embed.weight", (cfg.max_num_speaker, cfg.speaker_embedding_size), "embed"))
        out.append(("encoder.speaker_layer.weight", (cfg.speaker_embedding_size,) * 2, "w"))
        out.append(("encoder.speaker_layer.bias", (cfg.speaker_embedding_size,), "b"))
    if cfg.multi_lingual:
        out.append(("encoder.language_embed.weight", (cfg.language_embedding_size, cfg.max_num_language), "w"))
        out.append(("encoder.language_layer.weight", (cfg.language_embedding_size,) * 2, "w"))
        out.append(("encoder.language_layer.bias", (cfg.language_embedding_size,), "b"))
    # TransformerEncoder (modules.py:24-47): ModuleLists are registered list by list
    p = "encoder.encoder."
    out.append((p + "pe_scale", (), "scalar"))
    for i in range(cfg.n_encoder_layer):
        n = De if i == 0 else Dh
        out.append((p + "self_attentions.%d.qkv_transform.weight" % i, (3 * n, n), "w"))
        out.append((p + "self_attentions.%d.output_transform.weight" % i, (n, n), "w"))
    for i in range(cfg.n_encoder_layer):
        ln(p + "attn_layer_norms.%d" % i, De if i == 0 else Dh)
    for i in range(cfg.n_encoder_layer):
        out.append((p + "ffn_layers.%d.input_layer.weight" % i, (4 * Dh, Dh), "w"))
        out.append((p + "ffn_layers.%d.output_layer.weight" % i, (Dh, 4 * Dh), "w"))
    for i in range(cfg.n_encoder_layer):
        ln(p + "ffn_layer_norms.%d" % i, Dh)
    ln(p + "output_layer_norm", Dh)
    # Decoder (tacotron.py:94-105)
    out.append(("decoder.prenet.dense0.weight", (cfg.prenet_hidden, cfg.num_mels), "w"))
    out.append(("decoder.prenet.dense0.bias", (cfg.prenet_hidden,), "b"))
    out.append(("decoder.prenet.dense1.weight", (cfg.prenet_hidden, cfg.prenet_hidden), "w"))
    out.append(("decoder.prenet.dense1.bias", (cfg.prenet_hidden,), "b"))
    out.append(("decoder.prenet.dense_final.weight", (Dd, cfg.prenet_hidden), "w"))
    p = "decoder.decoder."
    out.append((p + "pe_scale", (), "scalar"))
    for i in range(cfg.n_decoder_layer):
        n = mem if i == 0 else Dd
        out.append((p + "self_attentions.%d.qkv_transform.weight" % i, (3 * n, n), "w"))
        out.append((p + "self_attentions.%d.output_transform.weight" % i, (n, n), "w"))
    for i in range(cfg.n_decoder_layer):
        ln(p + "attn_layer_norms.%d" % i, mem if i == 0 else Dd)
    for i in range(cfg.n_decoder_layer):
        out.append((p + "encdec_attentions.%d.q_transform.weight" % i, (Dd, Dd), "w"))
        out.append((p + "encdec_attentions.%d.kv_transform.weight" % i, (2 * Dd, Dd), "w"))
        out.append((p + "encdec_attentions.%d.output_transform.weight" % i, (Dd, Dd), "w"))
    for i in range(cfg.n_decoder_layer):
        ln(p + "encdec_layer_norms.%d" % i, mem if i == 0 else Dd)
    for i in range(cfg.n_decoder_layer):
        out.append((p + "ffn_layers.%d.input_layer.weight" % i, (4 * Dd, Dd), "w"))
        out.append((p + "ffn_layers.%d.output_layer.weight" % i, (Dd, 4 * Dd), "w"))
    for i in range(cfg.n_decoder_layer):
        ln(p + "ffn_layer_norms.%d" % i, Dd)
    ln(p + "output_layer_norm", Dd)
    out.append(("decoder.mel_net.weight", (cfg.num_mels, Dd), "w"))
    out.append(("decoder.stop_net.weight", (1, Dd), "w"))
    out.append(("decoder.stop_net.bias", (1,), "b"))
    # Postnet (tacotron.py:69-79)
    for i in range(cfg.n_postnet_layer):
        cin = cfg.num_mels if i == 0 else cfg.postnet_hidden
        cout = cfg.num_mels if i == cfg.n_postnet_layer - 1 else cfg.postnet_hidden
        out.append(("postnet.conv_layers.%d.weight" % i, (cout, cin, 5), "w"))
    for i in range(cfg.n_postnet_layer):
        cout = cfg.num_mels if i == cfg.n_postnet_layer - 1 else cfg.postnet_hidden
        q = "postnet.batchnorm_layers.%d." % i
        out.append((q + "weight", (cout,), "bn_w"))
        out.append((q + "bias", (cout,), "bn_b"))
        out.append((q + "running_mean", (cout,), "bn_rm"))
        out.append((q + "running_var", (cout,), "bn_rv"))
        out.append((q + "num_batches_tracked", (), "bn_nbt"))
    return out


def synthetic_state(cfg, seed=1234):
    """{name: np.ndarray} with non-degenerate values for every state_dict entry."""
    rng = np.random.default_rng(seed)
    st = {}
    for name, shape, kind in param_shapes(cfg):
        if kind == "embed":
            a = rng.standard_normal(shape) * 0.5
        elif kind == "w":
            fan_in = int(np.prod(shape[1:]))
            a = rng.standard_normal(shape) * (1.0 / np.sqrt(fan_in))
        elif kind in ("b", "ln_b", "bn_b"):
            a = rng.standard_normal(shape) * 0.1
        elif kind in ("ln_w", "bn_w"):
            a = 1.0 + rng.standard_normal(shape) * 0.1
        elif kind == "bn_rm":
            a = rng.standard_normal(shape) * 0.1
        elif kind == "bn_rv":
            a = 0.5 + rng.random(shape)
        elif kind == "bn_nbt":
            st[name] = np.asarray(3, dtype=np.int64)
            continue
        elif kind == "scalar":
            a = np.asarray(1.0 + 0.25 * rng.standard_normal())
        else:
            raise ValueError(kind)
        st[name] = np.asarray(a, dtype=np.float32)
    return st


from benchdata import synthetic_batch  # noqa: E402,F401  (one definition: the benchmark inputs are drawn from the neutral module)


def synthetic_corpus(dirpath, seed=0, n=48):
    """A tiny corpus in the reference's on-disk layout (corpora/process_corpus.py:296-348, dataloader.py:313-332,413-416):
    mels.zip of <name>.npy float32 [T, 80] members, metadata.train.txt lines `name.npy|frames|text|lang`,
    lang_id.json / spk_id.json.  Deterministic in (seed, n); used by the golden generator and by the tests that re-create
    the same files on the GPU box.  Returns the paths."""
    import io
    import json
    import os
    import zipfile
    rng = np.random.default_rng(seed)
    langs = ["en-us", "de-de", "fr-fr"]
    spks = ["spkA", "spkB", "spkC", "spkD"]
    words = ["hello", "world", "grüße", "café", "byte", "speech", "über", "naïve", "tts", "模型"]
    os.makedirs(dirpath, exist_ok=True)
    zpath, mpath = os.path.join(dirpath, "mels.zip"), os.path.join(dirpath, "metadata.train.txt")
    lines = []
    with zipfile.ZipFile(zpath, "w") as zf:
        for i in range(n):
            spk = spks[int(rng.integers(0, len(spks)))]
            lang = langs[int(rng.choice(3, p=[0.6, 0.25, 0.15]))]
            T = int(rng.integers(20, 120))
            mel = np.clip(rng.standard_normal((T, 80)), -4, 4).astype(np.float32)
            name = "%s_%04d.npy" % (spk, i)
            buf = io.BytesIO()
            np.save(buf, mel)
            zf.writestr(name, buf.getvalue())
            text = " ".join(words[int(j)] for j in rng.integers(0, len(words), size=int(rng.integers(1, 6))))
            lines.append("%s|%d|%s|%s" % (name, T, text, lang))
    with open(mpath, "w", encoding="utf-8") as f:
        f.write("\n".join(lines) + "\n")
    lang_ids = {l: i for i, l in enumerate(langs)}
    spk_ids = {s: i for i, s in enumerate(spks)}
    with open(os.path.join(dirpath, "lang_id.json"), "w") as f:
        json.dump(lang_ids, f)
    with open(os.path.join(dirpath, "spk_id.json"), "w") as f:
        json.dump(spk_ids, f)
    return {"zip": zpath, "meta": mpath, "lang_ids": lang_ids, "spk_ids": spk_ids}
