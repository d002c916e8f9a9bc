"""Synthetic benchmark inputs: batches in the dataloader's contract (dataloader.py:498-508) from NumPy's
default_rng -- reproducible on any machine, no reference and no test infrastructure needed.  bench.py / bench_decode.py draw their
inputs from here; the oracle's `synth.synthetic_batch` is this function (tests and benchmarks see the same data)."""
import numpy as np


def synthetic_batch(cfg, B, S, T, seed=0, in_lens=None, tgt_lens=None, n_spk=None, n_lang=None):
    """Batch dict in the dataloader's contract (dataloader.py:498-508), NumPy arrays.

    inputs ~ U{3..255} with sos=2 first and eos=1 last (utils/text.py:3-19), zero padded;
    mel_targets ~ N(0,1) clipped to [-4,4], zero beyond length.
    """
    rng = np.random.default_rng(seed)
    if in_lens is None:
        in_lens = np.round(np.linspace(S, max(2, 0.8 * S), B)).astype(np.int64)
    if tgt_lens is None:
        tgt_lens = np.round(np.linspace(T, max(1, 0.8 * T), B)).astype(np.int64)
    in_lens = np.asarray(in_lens, dtype=np.int64)
    tgt_lens = np.asarray(tgt_lens, dtype=np.int64)
    hi = min(256, cfg.vocab_size)
    inputs = rng.integers(3, hi, size=(B, S)).astype(np.int64)
    mels = np.clip(rng.standard_normal((B, T, cfg.num_mels)), -4, 4).astype(np.float32)
    for b in range(B):
        inputs[b, 0] = 2
        inputs[b, in_lens[b] - 1] = 1
        inputs[b, in_lens[b]:] = 0
        mels[b, tgt_lens[b]:] = 0
    n_spk = n_spk or cfg.max_num_speaker
    n_lang = n_lang or cfg.max_num_language
    spk = rng.integers(0, n_spk, size=(B,)).astype(np.int64)
    lang_ids = rng.integers(0, n_lang, size=(B,))
    lang = np.zeros((B, cfg.max_num_language), dtype=np.float32)
    lang[np.arange(B), lang_ids] = 1.0
    return {"inputs": inputs, "input_lengths": in_lens, "mel_targets": mels,
            "target_lengths": tgt_lens, "input_spk_ids": spk, "input_language_vecs": lang,
            "names": ["utt%03d" % i for i in range(B)]}
