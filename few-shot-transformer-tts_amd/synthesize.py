"""Autoregressive mel / stop-token decoding with the reference's entry point (synthesize.py:17-72).

eval_batch(model_eval, data) encodes once, then generates frame by frame until every utterance has emitted a
stop token or hp.max_generation_frames is reached, and runs the postnet once at the end.  On MI355X the frame
loop is a KV-cached single-position decoder step captured in a hipGraph (csrc/decode.hip): the step index, stop
flags and lengths live in device memory and the host polls `all finished` once every `sync_interval` frames
instead of once per frame (the reference syncs with torch.all every frame and recomputes the whole prefix).

Results follow the reference: mel_pre / mel_aft [B, T_gen, M], generated_lengths including the reference's
off-by-one for samples that never stop, exact zeros after a sample's stop, encoder-decoder alignments
[B, H, S, T_gen] per layer.  Decoder self-attention alignments (which the reference returns but nothing reads)
are produced only with keep_self_alignments=True.
"""
import copy
import ctypes as C
import logging
import time

import torch

from hyperparams import hparams as hp
from b2s_hip import lib as L
from b2s_hip.engine import _i32


def eval_batch(model_eval, data, use_bar=True, bar_interval=10, use_graph=True, sync_interval=16,
               keep_self_alignments=False):
    with torch.no_grad():
        tic = time.time()
        batch = copy.copy(data)
        device = batch['inputs'].device
        B = batch['inputs'].shape[0]
        eng = model_eval.engine()
        lib = eng.lib
        in32 = _i32(batch['input_lengths'])
        enc_outputs = model_eval.encoder(batch['inputs'], batch['input_lengths'], batch.get('input_spk_ids'),
                                         batch.get('input_language_vecs')).contiguous()
        S = enc_outputs.shape[1]
        max_frames = int(hp.max_generation_frames)
        train = bool(model_eval.decoder.training)          # the reference synthesises with decoder.train() (eval.py:116-117)
        nbytes = lib.b2s_decode_ws_bytes(eng.handle, B, S, max_frames, int(keep_self_alignments))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        stream = torch.cuda.Stream(device=device)          # hipGraph capture needs a non-default stream
        stream.wait_stream(torch.cuda.current_stream())
        state = L.P()
        frames, done = C.c_int(0), C.c_int(0)
        with torch.cuda.stream(stream):
            L.check(lib.b2s_decode_begin(eng.handle, L.ptr(enc_outputs), L.ptr(in32), B, S, max_frames, int(train), eng.next_seed(),
                                         int(keep_self_alignments), L.ptr(ws), nbytes, stream.cuda_stream, C.byref(state)))
            try:
                shown = 0
                while frames.value < max_frames and not done.value:
                    n = min(sync_interval, max_frames - frames.value)
                    L.check(lib.b2s_decode_run(eng.handle, state, n, int(use_graph), stream.cuda_stream))
                    L.check(lib.b2s_decode_status(state, C.byref(frames), C.byref(done), stream.cuda_stream))
                    if bar_interval != -1 and not use_bar and frames.value // bar_interval > shown:
                        shown = frames.value // bar_interval
                        print(frames.value)
                lengths = torch.empty(B, dtype=torch.int32, device=device)
                L.check(lib.b2s_decode_fetch(eng.handle, state, 0, L.ptr(torch.empty(1, device=device)), L.ptr(lengths), stream.cuda_stream))
                stream.synchronize()
                # the reference stops at the first frame count where every sample has stopped
                t_gen = int(lengths.max().item()) if done.value else max_frames
                t_gen = min(t_gen, frames.value)
                mels = torch.empty(B, t_gen, hp.num_mels, dtype=torch.float32, device=device)
                L.check(lib.b2s_decode_fetch(eng.handle, state, t_gen, L.ptr(mels), L.ptr(lengths), stream.cuda_stream))
                H = eng.cfg.n_attention_head
                alignments = {'self': [], 'encdec': []}
                for layer in range(eng.cfg.n_decoder_layer):
                    a = torch.empty(B, H, S, t_gen, dtype=torch.float32, device=device)
                    L.check(lib.b2s_decode_alignment(eng.handle, state, 1, layer, t_gen, L.ptr(a), stream.cuda_stream))
                    alignments['encdec'].append(a)
                    if keep_self_alignments:
                        a = torch.empty(B, H, t_gen, t_gen, dtype=torch.float32, device=device)
                        L.check(lib.b2s_decode_alignment(eng.handle, state, 0, layer, t_gen, L.ptr(a), stream.cuda_stream))
                        alignments['self'].append(a)
                stream.synchronize()
            finally:
                lib.b2s_decode_end(state)
        torch.cuda.current_stream().wait_stream(stream)
        mel_aft = model_eval.postnet(mels, lengths, _fuse_add=True)        # mels + postnet(mels), BN in eval mode
        for key in ('self', 'encdec'):
            alignments[key] = [a.cpu().numpy() for a in alignments[key]]
        toc = time.time()
        total_length = int(lengths.sum().item())
        logging.info("Time: %.4f, Samples: %d, Length: %d, Max length: %d, Real-time Factor: %.4f" % (
            toc - tic, B, total_length, int(lengths.max().item()), (toc - tic) / max(total_length, 1) * 80))
        return {'names': data.get('names'), 'mel_pre': mels.cpu().numpy(), 'mel_aft': mel_aft.cpu().numpy(),
                'alignments': alignments, 'input_lengths': list(batch['input_lengths'].cpu().numpy()),
                'generated_lengths': list(lengths.cpu().numpy())}


def eval_batch_recompute(model_eval, data):
    """The reference's cache-free loop on top of the same HIP decoder (full prefix recomputed every frame).
    Test utility: the KV-cached hipGraph path above must agree with it when dropout is off."""
    with torch.no_grad():
        batch = copy.copy(data)
        device = batch['inputs'].device
        B = batch['inputs'].shape[0]
        target_lengths = torch.ones([B], dtype=torch.int32, device=device)
        finished = torch.zeros([B], dtype=torch.bool, device=device)
        mels = torch.zeros([B, 0, hp.num_mels], dtype=torch.float32, device=device)
        enc = model_eval.encoder(batch['inputs'], batch['input_lengths'], batch.get('input_spk_ids'), batch.get('input_language_vecs'))
        align = None
        while not bool(torch.all(finished)) and mels.shape[1] < hp.max_generation_frames:
            dec_in = torch.cat([mels, torch.zeros([B, 1, hp.num_mels], device=device)], dim=1)
            mel_bef, stop_logits, align = model_eval.decoder(enc, batch['input_lengths'], dec_in, target_lengths, leave_one=True)
            stop = stop_logits[:, -1] > 0
            mels = torch.cat([mels, mel_bef[:, -1:]], dim=1)
            finished = torch.logical_or(finished, stop)
            target_lengths = torch.where(finished, target_lengths, target_lengths + 1)
        mel_aft = model_eval.postnet(mels, target_lengths, _fuse_add=True)
        return {'mel_pre': mels.cpu().numpy(), 'mel_aft': mel_aft.cpu().numpy(),
                'alignments': {k: [a.cpu().numpy() for a in align[k]] for k in ('self', 'encdec')},
                'generated_lengths': list(target_lengths.cpu().numpy())}
