"""Batch producer: oracle restatement vs the reference-generated goldens, product vs oracle."""
import os

import numpy as np
import pytest

from oracle import batching as OB

G = os.path.join(os.path.dirname(__file__), "golden")


def _hp():
    import hyperparams
    hp = hyperparams.hparams
    hp.override_from_dict(hyperparams.DEFAULTS)
    return hp


def _samples(g):
    return [{"name": str(g["names"][i]), "input": g["ex%d_input" % i], "mel_target": g["ex%d_mel" % i],
             "target_length": int(len(g["ex%d_mel" % i])), "language_vec": g["ex%d_lang" % i], "speaker_id": int(g["ex%d_spk" % i])}
            for i in range(int(g["n"]))]


def test_oracle_packer_matches_reference_goldens():
    g6, g7 = dict(np.load(os.path.join(G, "g6_misc.npz"))), dict(np.load(os.path.join(G, "g7_batching.npz")))
    sizes = [len(b) for b in OB.pack_into_batches(g6["pack_in_lens"], g6["pack_tgt_lens"], 8000, 7000000)]
    assert sizes == list(g6["pack_sizes"])
    for key, kw in (("pk_sizes_tight", {}), ("pk_sizes_single", {"single": True})):
        sizes = [len(b) for b in OB.pack_into_batches(g7["pk_in"], g7["pk_tgt"], 1200, 400000, **kw)]
        assert sizes == list(g7[key]), key
    sizes = [len(b) for b in OB.pack_into_batches(g7["pk_in"], None, 1200, 400000)]
    assert sizes == list(g7["pk_sizes_notarget"])
    for s, r in zip(g7["adapt_steps"], g7["adapt_rates"]):
        assert OB.adapt_rate(int(s), 1000, 2000, 0.25) == pytest.approx(float(r), abs=1e-12)
    assert [OB.adapt_rate(s, 30000, 30000, 0.25) for s in (29999, 30000)] == list(g7["adapt_rates_default"])


def test_oracle_collate_matches_reference_golden():
    g = dict(np.load(os.path.join(G, "g7_batching.npz")))
    b = OB.collate(_samples(g))
    for k in ("inputs", "input_lengths", "mel_targets", "target_lengths", "input_spk_ids", "input_language_vecs"):
        assert b[k].shape == g["batch_" + k].shape and np.array_equal(b[k], g["batch_" + k]), k
        assert str(g["dtype_" + k]) == "torch." + str(b[k].dtype), k
    assert b["names"] == [str(n) for n in g["names"]]


def test_product_packer_and_collate_match_oracle():
    from b2s_hip import batching as PB
    hp = _hp()
    g6, g7 = dict(np.load(os.path.join(G, "g6_misc.npz"))), dict(np.load(os.path.join(G, "g7_batching.npz")))
    packer = PB.BatchPacker.from_hparams(hp)
    assert [len(b) for b in packer.pack(g6["pack_in_lens"], g6["pack_tgt_lens"])] == list(g6["pack_sizes"])
    rng = np.random.default_rng(3)
    for trial in range(20):                                     # random unsorted lengths, random caps, with / without targets
        n = int(rng.integers(1, 80))
        il, tl = rng.integers(1, 200, size=n), rng.integers(1, 900, size=n)
        fl, ql = int(rng.integers(300, 9000)), int(rng.integers(20000, 8000000))
        for tgt in (tl, None):
            for single in (False, True):
                ref = [b for b in OB.pack_into_batches(il, tgt, fl, ql, single) if b]      # the reference's empty leading batch dropped
                got = PB.BatchPacker(fl, ql).pack(il, tgt, single)
                assert got == ref, (trial, tgt is None, single)
    tight = PB.BatchPacker(1200, 400000)
    assert [len(b) for b in tight.pack(g7["pk_in"], g7["pk_tgt"])] == list(g7["pk_sizes_tight"])
    assert [len(b) for b in tight.pack(g7["pk_in"], g7["pk_tgt"], single=True)] == list(g7["pk_sizes_single"][1:]) and g7["pk_sizes_single"][0] == 0
    # collate: identical arrays; with shape quanta only the padding grows
    samples = _samples(g7)
    b = PB.collate(samples, hp)
    for k in ("inputs", "input_lengths", "mel_targets", "target_lengths", "input_spk_ids", "input_language_vecs"):
        assert b[k].dtype == g7["batch_" + k].dtype and np.array_equal(b[k], g7["batch_" + k]), k
    q = PB.collate(samples, hp, s_quantum=16, t_quantum=32)
    S, T = b["inputs"].shape[1], b["mel_targets"].shape[1]
    assert q["inputs"].shape[1] % 16 == 0 and q["mel_targets"].shape[1] % 32 == 0
    assert 0 <= q["inputs"].shape[1] - S < 16 and 0 <= q["mel_targets"].shape[1] - T < 32
    assert np.array_equal(q["inputs"][:, :S], b["inputs"]) and not q["inputs"][:, S:].any()
    assert np.array_equal(q["mel_targets"][:, :T], b["mel_targets"]) and not q["mel_targets"][:, T:].any()
    assert np.array_equal(q["target_lengths"], b["target_lengths"])
    # sharding and the adapt ramp
    assert PB.shard(list(range(10)), 1, 4) == [1, 5, 9] and PB.shard(list(range(5)), 0, 1) == list(range(5))
    hp.parse("adapt_start_step=1000,adapt_end_step=2000,final_adapt_rate=0.25")
    for s, r in zip(g7["adapt_steps"], g7["adapt_rates"]):
        assert PB.adapt_rate(int(s), hp) == pytest.approx(float(r), abs=1e-12)
    _hp()


@pytest.mark.gpu
def test_device_stager_round_trip_and_model_accepts_staged_batches():
    import torch
    from b2s_hip import batching as PB
    hp = _hp()
    g7 = dict(np.load(os.path.join(G, "g7_batching.npz")))
    samples = _samples(g7)
    stager = PB.DeviceStager("cuda", depth=2)
    b1 = PB.collate(samples[:5], hp)
    b2 = PB.collate(samples[5:], hp)
    stager.put(b1); stager.put(b2)
    with pytest.raises(RuntimeError):
        stager.put(b1)
    for ref in (b1, b2):
        dev = stager.next()
        torch.cuda.synchronize()
        for k, v in ref.items():
            if isinstance(v, np.ndarray):
                assert dev[k].is_cuda and np.array_equal(dev[k].cpu().numpy(), v), k
            else:
                assert dev[k] == v
    # pinned buffers are reused: once every slot of the ring (depth + 1) has seen a shape, staging it again allocates nothing
    for _ in range(3):
        stager.put(b1); stager.next()
    n_pinned = len(stager._pinned)
    for _ in range(4):
        stager.put(b1); stager.next()
    assert len(stager._pinned) == n_pinned
