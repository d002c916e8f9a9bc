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
import os
import time
import traceback

import torch

from hyperparams import hparams as hp
from b2s_hip import lib as L
from b2s_hip.engine import _i32


def _to_host(tensors):
    """Device tensors -> NumPy arrays through page-locked staging buffers: all copies are queued asynchronously and waited for once.
    (A pageable `.cpu()` of the 2 GB of alignments of a 64 x 1000-frame job ran at ~9 GB/s and cost 40 % of the job; torch's host
    allocator caches the pinned blocks, so only the first call pays for pinning.)  Falls back to pageable copies if pinning fails."""
    try:
        host = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in tensors]
    except RuntimeError:
        return [t.cpu().numpy() for t in tensors]
    for h, t in zip(host, tensors):
        h.copy_(t, non_blocking=True)
    # every copy was queued on the current stream of ITS source device (not necessarily the current device): wait for each of those
    for dev in {t.device for t in tensors if t.is_cuda}:
        torch.cuda.current_stream(dev).synchronize()
    return [h.numpy() for h in host]


def _lane_bounds(B, lanes):
    lanes = max(1, min(int(lanes), B))
    base, extra = divmod(B, lanes)
    out, lo = [], 0
    for i in range(lanes):
        hi = lo + base + (1 if i < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def eval_batch(model_eval, data, use_bar=True, bar_interval=10, use_graph=True, sync_interval=16,
               keep_self_alignments=False, lanes=None, device_results=False):
    """lanes > 1: the batch is decoded as that many independent sub-batches, each with its own KV cache, hipGraph and HIP
    stream.  Utterances are independent (synthesize.py:23-47 never mixes batch rows), so the split changes no result; a
    lane whose utterances have all stopped ends early and its remaining frames are the zeros the reference writes after a
    stop.  Measured on MI355X (64 x 1000 frames): 1 / 2 / 4 lanes = 1.03 / 1.01 / 1.06 ms per frame step -- the frame loop
    is bound by the ~5 us the command processor needs per kernel dispatch, which concurrent streams share, so the default
    stays one lane (B2S_DECODE_LANES overrides); the option is for batches too large for one KV-cache allocation.
    device_results=True returns torch tensors on the device instead of NumPy arrays (no host copy)."""
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
        if lanes is None:
            lanes = int(os.environ.get("B2S_DECODE_LANES", "1"))
        H, NL, NM = eng.cfg.n_attention_head, eng.cfg.n_decoder_layer, hp.num_mels

        class Lane(object):
            pass
        Ls = []
        for lo, hi in _lane_bounds(B, lanes):
            ln = Lane()
            ln.lo, ln.hi, ln.B = lo, hi, hi - lo
            ln.enc = enc_outputs[lo:hi].contiguous()
            ln.in32 = in32[lo:hi].contiguous()
            ln.nbytes = lib.b2s_decode_ws_bytes(eng.handle, ln.B, S, max_frames, int(keep_self_alignments))
            ln.ws = torch.empty(ln.nbytes, dtype=torch.uint8, device=device)
            ln.stream = torch.cuda.Stream(device=device)       # hipGraph capture needs a non-default stream
            ln.stream.wait_stream(torch.cuda.current_stream())
            ln.state, ln.frames, ln.done, ln.active = L.P(), C.c_int(0), C.c_int(0), True
            Ls.append(ln)
        try:
            for ln in Ls:
                L.check(lib.b2s_decode_begin(eng.handle, L.ptr(ln.enc), L.ptr(ln.in32), ln.B, S, max_frames, int(train), eng.next_seed("decode"),
                                             int(keep_self_alignments), L.ptr(ln.ws), ln.nbytes, ln.stream.cuda_stream, C.byref(ln.state)))
            shown = 0
            bar = None
            if use_bar and bar_interval != -1:                  # the reference's progress display (synthesize.py:33-34,47-49)
                try:
                    import tqdm
                    bar = tqdm.tqdm()
                except ImportError:
                    bar = None
            while any(ln.active for ln in Ls):
                # A failing step ends the loop, it does not lose the job (synthesize.py:36,52-54: `except: traceback.print_exc(); break`):
                # what was generated up to the last completed status read is returned.
                try:
                    for ln in Ls:                               # enqueue every lane's next frames before waiting on any of them
                        if ln.active:
                            n = min(sync_interval, max_frames - ln.frames.value)
                            L.check(lib.b2s_decode_run(eng.handle, ln.state, n, int(use_graph), ln.stream.cuda_stream))
                    for ln in Ls:
                        if ln.active:
                            L.check(lib.b2s_decode_status(ln.state, C.byref(ln.frames), C.byref(ln.done), ln.stream.cuda_stream))
                            ln.active = ln.frames.value < max_frames and not ln.done.value
                except Exception:
                    traceback.print_exc()
                    logging.error("eval_batch: decode step failed after %d frames; returning what was generated" % max(ln.frames.value for ln in Ls))
                    for ln in Ls:
                        ln.active = False
                        try:                                    # frames completed before the failure (a dead device leaves the last count)
                            L.check(lib.b2s_decode_status(ln.state, C.byref(ln.frames), C.byref(ln.done), ln.stream.cuda_stream))
                        except Exception:
                            pass
                    break
                f = max(ln.frames.value for ln in Ls)
                if bar_interval != -1 and f // bar_interval > shown:
                    if bar is not None:
                        bar.update((f // bar_interval - shown) * bar_interval)
                    elif not use_bar:
                        print(f)
                    shown = f // bar_interval
            if bar is not None:
                bar.close()
            for ln in Ls:
                ln.lengths = torch.empty(ln.B, dtype=torch.int32, device=device)
                with torch.cuda.stream(ln.stream):
                    L.check(lib.b2s_decode_fetch(eng.handle, ln.state, 0, L.ptr(torch.empty(1, device=device)), L.ptr(ln.lengths),
                                                 ln.stream.cuda_stream))
            for ln in Ls:
                ln.stream.synchronize()
                # the reference stops at the first frame count where every sample has stopped
                ln.t_gen = int(ln.lengths.max().item()) if ln.done.value else max_frames
                ln.t_gen = min(ln.t_gen, ln.frames.value)
            t_gen = max(ln.t_gen for ln in Ls)
            if t_gen <= 0:      # (a failure before the first frame: the reference loop has no `align` to return either and raises)
                raise L.B2SError("eval_batch: no frame was generated")
            one = len(Ls) == 1
            mels = torch.empty(B, t_gen, NM, dtype=torch.float32, device=device) if one else \
                torch.zeros(B, t_gen, NM, dtype=torch.float32, device=device)
            lengths = torch.empty(B, dtype=torch.int32, device=device)
            alignments = {'self': [], 'encdec': []}
            for layer in range(NL):
                alignments['encdec'].append((torch.empty if one else torch.zeros)(B, H, S, t_gen, dtype=torch.float32, device=device))
                if keep_self_alignments:
                    alignments['self'].append((torch.empty if one else torch.zeros)(B, H, t_gen, t_gen, dtype=torch.float32, device=device))
            torch.cuda.synchronize(device)                      # the zero fills above ran on the caller's stream
            for ln in Ls:
                with torch.cuda.stream(ln.stream):
                    st = ln.stream.cuda_stream
                    m_l = mels if one else torch.empty(ln.B, ln.t_gen, NM, dtype=torch.float32, device=device)
                    L.check(lib.b2s_decode_fetch(eng.handle, ln.state, ln.t_gen, L.ptr(m_l), L.ptr(ln.lengths), st))
                    if not one:
                        mels[ln.lo:ln.hi, :ln.t_gen].copy_(m_l)
                    lengths[ln.lo:ln.hi].copy_(ln.lengths)
                    for layer in range(NL):
                        a = alignments['encdec'][layer] if one else torch.empty(ln.B, H, S, ln.t_gen, dtype=torch.float32, device=device)
                        L.check(lib.b2s_decode_alignment(eng.handle, ln.state, 1, layer, ln.t_gen, L.ptr(a), st))
                        if not one:
                            alignments['encdec'][layer][ln.lo:ln.hi, :, :, :ln.t_gen].copy_(a)
                        if keep_self_alignments:
                            a = alignments['self'][layer] if one else torch.empty(ln.B, H, ln.t_gen, ln.t_gen, dtype=torch.float32, device=device)
                            L.check(lib.b2s_decode_alignment(eng.handle, ln.state, 0, layer, ln.t_gen, L.ptr(a), st))
                            if not one:
                                alignments['self'][layer][ln.lo:ln.hi, :, :ln.t_gen, :ln.t_gen].copy_(a)
            for ln in Ls:
                ln.stream.synchronize()
        finally:
            for ln in Ls:
                if ln.state:
                    lib.b2s_decode_end(ln.state)
        for ln in Ls:
            torch.cuda.current_stream().wait_stream(ln.stream)
        mel_aft = model_eval.postnet(mels, lengths, _fuse_add=True)        # mels + postnet(mels), BN in eval mode
        if device_results:
            # results stay in HBM as torch tensors (the reference returns NumPy arrays: synthesize.py:57-61 -- for 64 x 1000 frames
            # that is a 2 GB device-to-host copy of the alignments, which a caller that goes on working on the GPU skips)
            torch.cuda.synchronize(device)
            return {'names': data.get('names'), 'mel_pre': mels, 'mel_aft': mel_aft, 'alignments': alignments,
                    'input_lengths': batch['input_lengths'], 'generated_lengths': lengths}
        n_self = len(alignments['self'])
        host = _to_host(alignments['self'] + alignments['encdec'] + [mels, mel_aft, lengths])
        alignments = {'self': host[:n_self], 'encdec': host[n_self:-3]}
        mel_pre_h, mel_aft_h, lengths_h = host[-3:]
        toc = time.time()
        total_length = int(lengths_h.sum())
        logging.info("Time: %.4f, Samples: %d, Length: %d, Max length: %d, Real-time Factor: %.4f" % (
            toc - tic, B, total_length, int(lengths_h.max()), (toc - tic) / max(total_length, 1) * 80))
        return {'names': data.get('names'), 'mel_pre': mel_pre_h, 'mel_aft': mel_aft_h,
                'alignments': alignments, 'input_lengths': list(batch['input_lengths'].cpu().numpy()),
                'generated_lengths': list(lengths_h)}


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


def save_eval_results(names, mel_pre, mel_aft, alignments, input_lengths, generated_lengths,
                      output_dir, save_trimmed_wave=False, n_plot_alignment=None):
    """Write the results of eval_batch to `output_dir` (reference synthesize.py:75-106, same signature and file names):
    `<name>.npy` = mel_aft[i][:generated_lengths[i]] always; `<name>.wav` (+ `<name>_trim.wav`), `<name>_mel.png` and
    `<name>_align.png` when the reference's vocoder / plotting helpers (`utils.audio`, `utils.infolog`: out of this
    package's scope) are importable, i.e. when a reference checkout follows this package on sys.path.  A failing sample
    is logged and skipped, as in the reference; samples are written by a small thread pool."""
    import threading
    import traceback
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np
    try:
        from utils.audio import mel2wav, save_wav, trim_silence_intervals
    except Exception:                       # librosa / soundfile / the reference checkout are optional
        mel2wav = save_wav = trim_silence_intervals = None
    try:
        from utils.infolog import plot_attn, plot_mel
    except Exception:                       # matplotlib / fastdtw are optional
        plot_attn = plot_mel = None
    os.makedirs(output_dir, exist_ok=True)

    def save_one(i):
        try:
            name, n = names[i], int(generated_lengths[i])
            mel = np.asarray(mel_aft[i])[:n]
            np.save(os.path.join(output_dir, '%s.npy' % name), mel)
            if mel2wav is not None:
                wav = mel2wav(mel)
                save_wav(wav, os.path.join(output_dir, '%s.wav' % name))
                if save_trimmed_wave:
                    save_wav(trim_silence_intervals(wav), os.path.join(output_dir, '%s_trim.wav' % name))
            if plot_mel is not None:
                plot_mel(os.path.join(output_dir, '%s_mel.png' % name), mel)
                if n_plot_alignment is None or i < n_plot_alignment:
                    aligns = [np.asarray(a[i]).transpose([0, 2, 1]) for a in alignments["encdec"]]
                    plot_attn(aligns, os.path.join(output_dir, '%s_align.png' % name), enc_length=input_lengths[i], dec_length=n)
        except Exception:
            logging.error('Fail to produce eval output: ' + str(names[i]))
            logging.error(traceback.format_exc())

    tic = time.time()
    with ThreadPoolExecutor(max_workers=4) as pool:
        list(pool.map(save_one, range(len(names))))
    logging.info('[%s] Finished saving evals in %.2f secs: ' % (threading.current_thread().name, time.time() - tic) + str(names))
