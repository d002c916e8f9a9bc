"""Decode leg of bench.py: autoregressive eval_batch-equivalent (encoder + KV-cached hipGraph frame loop + postnet).

Workload (BASELINE.json configs[3]): 64 utterances x 1000 mel frames, S=160, stop bias -100 so that exactly
max_generation_frames=1000 steps run, default hparams, bf16, dropout at the reference's rates (the reference
synthesises with decoder.train()).  Metric: generated frames per second over the whole job.
"""
import json
import os
import time

import numpy as np
import torch

PEAK_HBM_GBS = 8000.0


def csrc_sha16():
    """sha256 (first 16 hex digits) over the kernel sources (csrc/*.hip, *.h in name order): stamped into every PMC summary, so that a bench
    line can say whether the counters it quotes were collected on the kernels it is timing (the GPU box has no .git to ask)."""
    import hashlib
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "few-shot-transformer-tts_amd", "csrc")
    h = hashlib.sha256()
    for n in sorted(os.listdir(d)):
        if n.endswith((".hip", ".h")):
            h.update(n.encode()); h.update(open(os.path.join(d, n), "rb").read())
    return h.hexdigest()[:16]


def load_pmc(suffix):
    """Newest committed PMC summary profiles/rNN_<suffix> -> (file name, commit it was collected at [+ whether the kernel sources changed
    since], rows); (None, None, []) if none."""
    pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    names = sorted(n for n in (os.listdir(pdir) if os.path.isdir(pdir) else []) if n.endswith(suffix) and n[0] == "r" and n[1:3].isdigit())
    if not names:
        return None, None, []
    d = json.load(open(os.path.join(pdir, names[-1])))
    if isinstance(d, dict):
        stamp = d.get("commit", "unknown")
        if d.get("csrc_sha16"):
            stamp += "; kernel sources unchanged since" if d["csrc_sha16"] == csrc_sha16() else "; KERNEL SOURCES CHANGED since that collection"
        else:
            stamp += "; collected before the source stamp existed (kernel sources have changed since)"
        return names[-1], stamp, d["rows"]
    return names[-1], "unrecorded (round-2 file)", d


def decode_bytes_per_step(B, S, t, e, Dd=768, Ld=6, weights=49.9e6):
    """weights once + self-KV read + cross-KV read + KV append."""
    return weights * e + Ld * 2 * B * Dd * e * (t + S + 1)


def cpu_baseline_decode(frames=60, B=8, S=100):
    from oracle import b2s_oracle as O, synth, make_config
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    cfg = make_config("max_generation_frames=%d,transformer_dropout_rate=0.0,decoder_dropout_rate=0.0" % frames)
    st = synth.synthetic_state(cfg, 1)
    st["decoder.stop_net.bias"] = np.full((1,), -100.0, dtype=np.float32)
    P = O.to_torch_state(st)
    nb = synth.synthetic_batch(cfg, B, S, 4, seed=0, n_spk=1, n_lang=1)
    nb.pop("mel_targets"); nb.pop("target_lengths")
    t0 = time.time()
    r = O.eval_batch(P, cfg, O.to_torch_batch(nb))
    dt = time.time() - t0
    return {"value": round(B * frames / dt, 1), "unit": "mel-frames/s", "cores": cores, "kind": "port",
            "sample": "oracle eval_batch (reference algorithm, no KV cache), B=%d S=%d, %d frames, dropout 0; cost grows ~quadratically "
                      "per frame, so 64x1000 is not run on the CPU" % (B, S, frames), "seconds": round(dt, 2)}


def run_decode(args, rank, world, device):
    from hyperparams import hparams as hp
    from transformer.tacotron import Tacotron, initialize_variables
    import synthesize
    from benchdata import synthetic_batch
    B, S, frames = 64, 160, 1000
    hp.parse("compute_dtype=%s,max_generation_frames=%d" % (args.dtype, frames))
    torch.manual_seed(0)
    model = Tacotron(hp)
    initialize_variables(model)
    with torch.no_grad():
        model.decoder.stop_net.bias.fill_(-100.0)
    model = model.to(device)
    model.eval()
    model.decoder.train()                      # the reference's synthesis mode (eval.py:116-117): decoder dropout live
    nb = synthetic_batch(hp, B, S, 4, seed=rank, in_lens=[S] * B, n_spk=1, n_lang=1)
    nb.pop("mel_targets"); nb.pop("target_lengths")
    batch = {k: (torch.from_numpy(np.asarray(v)).to(device) if not isinstance(v, list) else v) for k, v in nb.items()}
    # untimed warm-up: one full-size job (the caching allocator then holds the 1000-frame KV caches / alignment buffers and the frame
    # graph has been instantiated once; a short job would leave those one-time costs inside the first timed repetition)
    r = synthesize.eval_batch(model, batch, use_bar=False, bar_interval=-1, sync_interval=64, device_results=True)
    int(torch.clamp(r["generated_lengths"], max=frames).sum().item())      # (first use of these torch ops loads their code objects: ~0.1 s)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    reps = max(1, args.steps // 10)
    t0 = time.perf_counter()
    total = 0
    for _ in range(reps):
        # results stay in HBM (the timed job ends when mels, lengths and alignments are complete on the device); the previous job's
        # results are released first, as a serving loop would, so that the allocator can hand the same blocks to this one
        r = None
        r = synthesize.eval_batch(model, batch, use_bar=False, bar_interval=-1, sync_interval=64, device_results=True)
        total += int(torch.clamp(r["generated_lengths"], max=frames).sum().item())
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    assert bool(torch.isfinite(r["mel_aft"]).all())
    # the same job with the reference's return contract (NumPy arrays on the host: + a 2 GB device-to-host copy of the alignments)
    r = None
    r = synthesize.eval_batch(model, batch, use_bar=False, bar_interval=-1, sync_interval=64)       # (untimed: pins the host staging buffers)
    r = None
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(reps):                              # same number of repetitions as the device-resident figure
        r = None
        r = synthesize.eval_batch(model, batch, use_bar=False, bar_interval=-1, sync_interval=64)
    host_elapsed = (time.perf_counter() - t1) / reps
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    assert np.isfinite(r["mel_aft"]).all()
    if rank == 0:
        e = 2 if args.dtype == "bf16" else 4
        step_ms = elapsed / reps / frames * 1e3
        avg_bytes = float(np.mean([decode_bytes_per_step(B, S, t, e) for t in range(frames)]))
        ach = avg_bytes / (step_ms * 1e-3) / 1e9
        # HBM-side bytes per frame from the committed PMC run (tools/gpu_pmc_decode.sh: FETCH_SIZE x 2 + WRITE_SIZE, separate passes),
        # summed over the frame's kernels
        traffic = traffic_source = None
        pmc, pmc_commit, pmc_rows = load_pmc("decode_bf16_pmc_hbm_traffic.json")
        if pmc and args.dtype == "bf16":
            rows = [r for r in pmc_rows if r["kernel"].startswith("k_df_")]
            traffic_source = "profiles/%s @ commit %s (committed rocprofv3 --pmc run, not measured in this process)" % (pmc, pmc_commit)
            nfr = sum(r["launches"] for r in rows if r["kernel"].startswith("k_df_final"))
            if nfr:
                traffic = round(sum(r["launches"] * (r["fetch_MB_per_launch_corrected_x2"] + (r["WRITE_SIZE_KB_per_launch"] or 0) / 1024)
                                    for r in rows) / nfr * 1024 * 1024)
        out = {"metric": "autoregressive decode mel-frames/sec (encoder + KV-cached hipGraph frame loop + postnet)",
               "value": round(world * total / elapsed, 1), "unit": "mel-frames/s", "n_gpus": world, "steps": reps * frames,
               "warmup": args.warmup, "ms_per_step": round(step_ms, 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": "eval_batch: %d utterances x %d frames, S=%d, stop bias -100, decoder dropout on, default hparams"
                                      % (B, frames, S), "parallelism": "replicas x%d" % world},
               "value_incl_host_copy": round(B * frames / host_elapsed, 1),      # the reference's return contract (synthesize.py:57-61: NumPy)
               "ms_per_step_incl_host_copy": round(host_elapsed / frames * 1e3, 4),
               "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                            "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": traffic, "traffic_unit": "bytes per frame (HBM side, PMC)", "traffic_source": traffic_source,
                            "bytes_per_step_avg": avg_bytes,
                            "note": "algorithmic bytes per frame step (weights + KV) / wall time per frame incl. encoder and postnet; results "
                                    "(mels, lengths, alignments) complete in HBM -- value_incl_host_copy adds the reference's NumPy return (PCIe)"}}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_decode()
    if world > 1:
        torch.distributed.destroy_process_group()
    if rank == 0:
        import ctypes
        ctypes.CDLL(None).fflush(None)      # keep the JSON line last (RCCL prints through C stdio)
        print(json.dumps(out), flush=True)
