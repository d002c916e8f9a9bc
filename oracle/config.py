"""Model hyper-parameters used by the oracle (TEST INFRASTRUCTURE ONLY).

Restates the model-size / optimiser subset of the reference's global HParams
(hyperparams.py:19-66).  A config is a plain types.SimpleNamespace.
"""
from types import SimpleNamespace

# hyperparams.py:4,19,24-35,54-66
_DEFAULTS = dict(
    num_mels=80,
    max_generation_frames=1100,
    vocab_size=6000,
    embed_size=512,
    encoder_hidden=512,
    decoder_hidden=768,
    n_encoder_layer=6,
    n_decoder_layer=6,
    n_attention_head=8,
    transformer_dropout_rate=0.1,
    decoder_dropout_rate=0.5,
    prenet_hidden=256,
    postnet_hidden=512,
    n_postnet_layer=5,
    reg_weight=5e-9,
    multi_speaker=True,
    max_num_speaker=1000,
    speaker_embedding_size=128,
    multi_lingual=True,
    max_num_language=100,
    language_embedding_size=128,
    warmup_steps=50000,
    max_lr=1e-3,
    min_lr=1e-5,
    lr_decay_step=550000,
    lr_decay_rate=1e-2,
    adam_eps=5e-8,
    # extensions named by the north-star, absent upstream (default OFF)
    guided_attention_weight=0.0,
    guided_attention_sigma=0.2,
    freeze_encoder=False,
)

# Small configurations used by the golden fixtures.  TINY exercises head sizes 32 (encoder)
# and 64 (decoder); TINY96 exercises the real head sizes 64 (encoder) and 96 (decoder).
TINY = ("vocab_size=260,embed_size=64,encoder_hidden=64,decoder_hidden=128,"
        "speaker_embedding_size=32,language_embedding_size=32,n_attention_head=2,"
        "n_encoder_layer=2,n_decoder_layer=2,prenet_hidden=32,postnet_hidden=48,"
        "n_postnet_layer=3,max_num_speaker=8,max_num_language=8,"
        "transformer_dropout_rate=0.0,decoder_dropout_rate=0.0")
TINY96 = ("vocab_size=260,embed_size=128,encoder_hidden=128,decoder_hidden=192,"
          "speaker_embedding_size=32,language_embedding_size=32,n_attention_head=2,"
          "n_encoder_layer=2,n_decoder_layer=2,prenet_hidden=32,postnet_hidden=48,"
          "n_postnet_layer=3,max_num_speaker=8,max_num_language=8,"
          "transformer_dropout_rate=0.0,decoder_dropout_rate=0.0")


def default_config():
    return SimpleNamespace(**_DEFAULTS)


def make_config(overrides=""):
    """overrides: 'a=1,b=2.0' string in the reference's HParams.parse syntax (scalars only)."""
    cfg = default_config()
    if overrides:
        for kv in overrides.split(","):
            k, v = kv.split("=")
            k = k.strip()
            old = getattr(cfg, k)
            if isinstance(old, bool):
                val = v.strip().lower() in ("1", "true")
            elif isinstance(old, int):
                val = int(v)
            else:
                val = float(v)
            setattr(cfg, k, val)
    return cfg
