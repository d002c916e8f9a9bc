"""End to end on the GPU: disk (synthetic corpus in the reference's formats) -> TrainFeeder -> pinned staging -> fused HIP
trainer -> checkpoint -> EvalFeeder -> KV-cached decode -> `<name>.npy` files."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import synth, TINY


def test_corpus_to_trained_model_to_mel_files(tmp_path):
    import hyperparams
    from transformer.tacotron import Tacotron, initialize_variables
    from b2s_hip import corpus as C, batching as PB
    from b2s_hip.trainer import HipTrainer
    from utils import checkpoint
    import synthesize
    hp = hyperparams.hparams
    hp.override_from_dict(hyperparams.DEFAULTS)
    hp.parse(TINY.replace("max_num_speaker=8", "max_num_speaker=8").replace("max_num_language=8", "max_num_language=100") +
             ",bucket_size=16,batch_frame_limit=400,batch_frame_quad_limit=60000,balanced_training=false,data_warmup_steps=0,"
             "max_lr=0.002,max_generation_frames=24")
    c = synth.synthetic_corpus(str(tmp_path / "corpus"), seed=3, n=48)
    torch.manual_seed(0)
    m = Tacotron(hp)
    initialize_variables(m)
    m = m.cuda().train()
    tr = HipTrainer(m, hp)
    feeder = C.TrainFeeder(c["zip"], c["meta"], hp, c["spk_ids"], c["lang_ids"], queue_size=4)
    feeder.start()
    stager = PB.DeviceStager("cuda")
    stager.put(feeder.get_batch())
    losses = []
    for step in range(12):
        batch = stager.next()
        stager.put(feeder.get_batch())                      # next batch's H2D copy overlaps this step
        losses.append(float(tr.train_step(batch)[0]))
    assert np.isfinite(losses).all() and np.mean(losses[-4:]) < np.mean(losses[:4]), losses
    path = checkpoint.save_model(str(tmp_path), m, tr, tr.sched, tr.global_step)
    m2 = Tacotron(hp).cuda()
    assert checkpoint.load_model(path, m2, None, None, "cuda") == 12
    m2.eval()
    fe = C.EvalFeeder(c["zip"], c["meta"], hp, c["spk_ids"], c["lang_ids"], eval_lang=["fr-fr"], shuffle=False)
    nb = fe.fetch_data()[0]
    dev = {k: (torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) else v) for k, v in nb.items()}
    out = synthesize.eval_batch(m2, dev, use_bar=False)
    files = C.save_mels(out["names"], out["mel_aft"], out["generated_lengths"], str(tmp_path / "out"))
    assert [os.path.basename(f)[:-4] for f in files] == nb["names"]
    for f, n in zip(files, out["generated_lengths"]):
        mel = np.load(f)
        # (an utterance that never stops reports max_generation_frames + 1, the reference's off-by-one: synthesize.py:56-61)
        assert mel.shape == (min(int(n), hp.max_generation_frames), hp.num_mels) and mel.dtype == np.float32 and np.isfinite(mel).all()
    hp.override_from_dict(hyperparams.DEFAULTS)
