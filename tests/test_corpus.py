"""On-disk formats and feeders against tests/golden/g8_feeders.json, which holds the batch
sequences the REFERENCE's Feeder / FeederEval produce on the synthetic corpus of oracle/synth.py (same files are
re-created here from the same seed)."""
import json
import os

import numpy as np
import pytest

from oracle import synth

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "g8_feeders.json"), encoding="utf-8"))
CFGS = {
    "balanced": "bucket_size=16,batch_frame_limit=400,batch_frame_quad_limit=60000,data_warmup_steps=5,"
                "target_length_lower_bound=40,target_length_upper_bound=100",
    "plain": "bucket_size=16,batch_frame_limit=400,batch_frame_quad_limit=60000,balanced_training=false,data_warmup_steps=0",
    "adapt": "bucket_size=16,batch_frame_limit=400,batch_frame_quad_limit=60000,data_warmup_steps=0,"
             "adapt_start_step=0,adapt_end_step=0,final_adapt_rate=0.5",
}


def _hp(over=""):
    import hyperparams
    hp = hyperparams.hparams
    hp.override_from_dict(hyperparams.DEFAULTS)
    if over:
        hp.parse(over)
    return hp


@pytest.fixture(scope="module")
def corpus(tmp_path_factory):
    return synth.synthetic_corpus(str(tmp_path_factory.mktemp("corpus")), seed=3, n=48)


def test_metadata_and_zip_readers(corpus, tmp_path):
    from b2s_hip import corpus as C
    rows = C.read_metadata(corpus["meta"], "nlti")
    assert len(rows) == 48 and rows[:3] == G["meta_head"]
    assert all(r["i"] == "de-de" for r in C.read_metadata(corpus["meta"], "nlti", inc_lang=["de-de"]))
    assert all(C.speaker_of(r["n"]) == "spkA" for r in C.read_metadata(corpus["meta"], "nlti", inc_spk=["spkA"]))
    with pytest.raises(ValueError):
        C.read_metadata(corpus["meta"], "xyz")
    tab = tmp_path / "m.txt"                                   # tab separated + phone column
    tab.write_text("a_1.npy\t12\thi\tHH AY\ten-us\n", encoding="utf-8")
    assert C.read_metadata(str(tab), "nltpi") == [{"n": "a_1.npy", "l": "12", "t": "hi", "p": "HH AY", "i": "en-us"}]
    mz = C.MelZip(corpus["zip"])
    mel = mz.load(rows[0]["n"])
    assert mel.dtype == np.float32 and mel.shape == (int(rows[0]["l"]), 80)
    hp = _hp()
    ex = C.make_example(rows[0], mz, hp, corpus["spk_ids"], corpus["lang_ids"])
    assert ex["name"] == rows[0]["n"][:-4] and ex["input"][0] == 2 and ex["input"][-1] == 1      # sos ... eos
    assert bytes(ex["input"][1:-1].astype(np.uint8)).decode("utf-8") == rows[0]["t"]
    assert ex["language_vec"].argmax() == corpus["lang_ids"][rows[0]["i"]] and ex["language_vec"].sum() == 1
    assert ex["speaker_id"] == corpus["spk_ids"][C.speaker_of(rows[0]["n"])]
    # output side: <name>.npy trimmed to the generated length
    paths = C.save_mels(["u1", "u2"], np.arange(2 * 7 * 80, dtype=np.float32).reshape(2, 7, 80), [5, 7], str(tmp_path / "out"))
    assert [np.load(p).shape for p in paths] == [(5, 80), (7, 80)] and np.load(paths[0]).dtype == np.float32


@pytest.mark.parametrize("tag", sorted(CFGS))
@pytest.mark.parametrize("rank,world", [(0, 1), (1, 2)])
def test_train_feeder_reproduces_reference_batch_sequence(corpus, tag, rank, world):
    from b2s_hip import corpus as C
    hp = _hp(CFGS[tag])
    kw = dict(adapt_lang=["fr-fr"]) if tag == "adapt" else {}
    f = C.TrainFeeder(corpus["zip"], corpus["meta"], hp, corpus["spk_ids"], corpus["lang_ids"], rank=rank, world_size=world, **kw)
    got = []
    for _ in range(3):
        got.extend(f.next_group())
    ref = G["train/%s/r%dw%d" % (tag, rank, world)]
    assert [b["names"] for b in got] == [b["names"] for b in ref]
    for b, r in zip(got, ref):
        assert int(b["inputs"].sum()) == r["in_sum"] and abs(float(b["mel_targets"].astype(np.float64).sum()) - r["mel_sum"]) < 1e-6
        assert b["target_lengths"].tolist() == r["tl"] and b["input_spk_ids"].tolist() == r["spk"]
        assert b["input_language_vecs"].argmax(1).tolist() == r["lang"]
    # state round trip: a restored feeder continues with the same batches
    st = f.state_dict()
    nxt = [b["names"] for b in f.next_group()]
    f.load_state_dict(st)
    assert [b["names"] for b in f.next_group()] == nxt
    _hp()


def test_eval_feeder_reproduces_reference(corpus):
    from b2s_hip import corpus as C
    hp = _hp("batch_frame_limit=400,batch_frame_quad_limit=60000")
    fe = C.EvalFeeder(corpus["zip"], corpus["meta"], hp, corpus["spk_ids"], corpus["lang_ids"], shuffle=True, keep_order=True, pick_partial=True)
    got = fe.fetch_data()
    assert [b["names"] for b in got] == [b["names"] for b in G["eval/partial"]]
    assert [b["target_lengths"].tolist() for b in got] == [b["tl"] for b in G["eval/partial"]]
    fe = C.EvalFeeder(corpus["zip"], corpus["meta"], hp, corpus["spk_ids"], corpus["lang_ids"], eval_lang=["de-de"], shuffle=False, target_spk="spkB")
    got = fe.fetch_data()
    assert [b["names"] for b in got] == [b["names"] for b in G["eval/de_as_spkB"]]
    assert [b["input_spk_ids"].tolist() for b in got] == [b["spk"] for b in G["eval/de_as_spkB"]]
    _hp()


def test_feeder_thread_delivers_batches(corpus):
    from b2s_hip import corpus as C
    hp = _hp(CFGS["plain"])
    f = C.TrainFeeder(corpus["zip"], corpus["meta"], hp, corpus["spk_ids"], corpus["lang_ids"], queue_size=4)
    f.start()
    seen = [f.get_batch()["names"] for _ in range(6)]
    assert seen == [b["names"] for b in G["train/plain/r0w1"][:6]]
    _hp()
