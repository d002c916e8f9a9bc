"""Autoregressive mel / stop-token decoding with the reference's entry point (synthesize.py:17-72).

eval_batch(model_eval, data) encodes once and then generates frame by frame until every utterance has
emitted a stop token or hp.max_generation_frames is reached; the postnet runs once at the end.  Results
(mel_pre, mel_aft, generated_lengths incl. the reference's off-by-one for never-stopping samples, last-step
alignments) follow the reference.
"""
import copy
import logging
import time

import torch

from hyperparams import hparams as hp


def eval_batch(model_eval, data, use_bar=True, bar_interval=10):
    with torch.no_grad():
        tic = time.time()
        batch = copy.copy(data)
        device = batch['inputs'].device
        batch_size = batch['inputs'].shape[0]
        target_lengths = torch.ones([batch_size], dtype=torch.int32, device=device)
        finished = torch.zeros([batch_size], dtype=torch.bool, device=device)
        mels = torch.zeros([batch_size, 0, hp.num_mels], dtype=torch.float32, device=device)
        enc_outputs = model_eval.encoder(batch['inputs'], batch['input_lengths'], batch.get('input_spk_ids'),
                                         batch.get('input_language_vecs'))
        align = None
        steps = 0
        while mels.shape[1] < hp.max_generation_frames:
            if steps % 8 == 0 and bool(torch.all(finished)):        # host check every 8 steps (one D2H sync)
                break
            decoder_input = torch.cat([mels, torch.zeros([batch_size, 1, hp.num_mels], device=device)], dim=1)
            mel_bef, stop_logits, align = model_eval.decoder(enc_outputs, batch['input_lengths'], decoder_input,
                                                             target_lengths, leave_one=True)
            stop = stop_logits[:, -1] > 0
            mels = torch.cat([mels, mel_bef[:, -1:]], dim=1)
            finished = torch.logical_or(finished, stop)
            target_lengths = torch.where(finished, target_lengths, target_lengths + 1)
            steps += 1
            if bool(torch.all(finished)):
                break
        mel_aft = mels + model_eval.postnet(mels, target_lengths)
        alignments = {k: [a.cpu().numpy() for a in align[k]] for k in ('self', 'encdec')} if align is not None else None
        toc = time.time()
        total_length = target_lengths.sum().item()
        logging.info("Time: %.4f, Samples: %d, Length: %d, Max length: %d, Real-time Factor: %.4f" % (
            toc - tic, mels.shape[0], total_length, target_lengths.max().item(), (toc - tic) / max(total_length, 1) * 80))
        return {'names': data.get('names'), 'mel_pre': mels.cpu().numpy(), 'mel_aft': mel_aft.cpu().numpy(),
                'alignments': alignments, 'input_lengths': list(batch['input_lengths'].cpu().numpy()),
                'generated_lengths': list(target_lengths.cpu().numpy())}
