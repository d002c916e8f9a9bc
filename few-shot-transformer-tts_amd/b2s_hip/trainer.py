"""Fused data-parallel training step on the HIP engine (the bench / production path).

Replaces the reference's `m(**batch) -> compute_loss -> loss.backward() -> optim.step()` + DistributedDataParallel
(train.py:118-131,165-191) with: one pass of engine calls that enqueue every forward / backward kernel, RCCL
all-reduce of each backward stage's gradient range launched from the engine's stage hook while later stages still
compute (one process per GPU, torch.distributed backend "nccl" = RCCL over xGMI), and one fused multi-tensor Adam
kernel (bias-corrected, L2 folded in for the reference's member set, gradient mean = 1/world folded in).
Semantics match the reference: per-rank loss normalisation and per-rank BatchNorm statistics, gradients averaged
across ranks, LambdaLR schedule `learning_rate_schedule`.
"""
import ctypes as C
import os
import weakref

import torch

from . import lib as L
from .engine import _i32, encoder_stream
from .dp import GradBucketer

_HOOK_T = C.CFUNCTYPE(None, C.c_int, C.c_void_p)


class _SchedView(object):
    def __init__(self, trainer):
        self._t = trainer

    @property
    def last_epoch(self):
        return self._t.global_step

    def get_last_lr(self):
        return [self._t.hp.max_lr * self._t.lr_lambda(self._t.global_step)]

    def state_dict(self):           # torch.optim.lr_scheduler.LambdaLR.state_dict() layout (lambdas are not serialised)
        t = self._t
        return {"base_lrs": [t.hp.max_lr], "last_epoch": t.global_step, "_step_count": t.global_step + 1,
                "_get_lr_called_within_step": False, "_last_lr": self.get_last_lr(), "lr_lambdas": [None]}

    def load_state_dict(self, sd):
        self._t.global_step = int(sd["last_epoch"])


class HipTrainer(object):
    def __init__(self, model, hp, beta1=0.9, beta2=0.999, bucket_mb=32.0, grad_payload=None, dist=None, tail_adam=True, overlap_encoder=True,
                 dp_mode=None, side_stream=True):
        from transformer.tacotron import learning_rate_schedule
        self.model, self.hp = model, hp
        # tail_adam: with the encoder backward on its own stream, this stream is idle from the end of the decoder backward until the
        # encoder's last weight gradients are done (~0.9 ms); the decoder / postnet update (HBM-bound) runs there, behind the second
        # stream's last decoder weight-gradient group, and only the encoder group's update follows the encoder backward.  False: one
        # optimizer launch after the whole backward pass.  (Schedules measured and dropped, profiles/NOTES_r02.md / r04.md: the update
        # on the second stream under the encoder backward, the update overlapped with the next forward pass.)
        self.tail_adam = bool(tail_adam)
        # overlap_encoder: the encoder is 5 % of the step's FLOPs in ~30 launches per pass of ~112 workgroups -- a fifth of the step on
        # a quarter of the chip.  Its forward runs on a stream of its own beside the decoder's prenet and first self-attention (which
        # do not read the encoder output), its backward starts as soon as d(memory) is complete, beside the first decoder layer's
        # self-attention backward, the prenet backward and their weight gradients.  False: the single chain.
        self.overlap_encoder = bool(overlap_encoder)
        # side_stream: during the decoder backward the encoder stream is idle; the dK / dV kernels of the encoder-decoder attentions (224
        # workgroups, ~30 us each, feeding only memory-side gradients) run there instead of in the decoder's chain (b2s_model_set_side_stream)
        self.side_stream = bool(side_stream) and os.environ.get("B2S_SIDE_STREAM", "1") != "0"      # (B2S_SIDE_STREAM=0: A/B switch, as B2S_COMPACT)
        self._enc_stream = None
        self.eng = model.engine()
        self.eng.ensure_bound()
        self.lib = self.eng.lib
        self.lr_lambda = lambda step: learning_rate_schedule(step, hp)
        self.beta1, self.beta2 = beta1, beta2
        g = self.eng._gflat
        self.exp_avg = torch.zeros_like(g)
        self.exp_avg_sq = torch.zeros_like(g)
        n = len(self.eng.names)
        m_ptrs, v_ptrs = (L.P * n)(), (L.P * n)()
        for i, name in enumerate(self.eng.names):
            if name in self.eng.param_offsets:
                off, _ = self.eng.param_offsets[name]
                m_ptrs[i] = self.exp_avg.data_ptr() + 4 * off
                v_ptrs[i] = self.exp_avg_sq.data_ptr() + 4 * off
        L.check(self.lib.b2s_adam_bind(self.eng.handle, m_ptrs, v_ptrs, n))
        # parameters replaced later (.to(), load_state_dict(assign=True), .data swaps) re-bind the engine; the C side rebuilds
        # the fused optimizer's pointer table against the new tensors (b2s_model_bind), the moments stay these buffers
        self.eng._trainer = weakref.ref(self)
        self._hook_error = None
        self.grad_probe = None          # test aid: called as grad_probe(flat_fp32_gradients, bf16_wire_or_None) right before the optimizer step
        self.global_step = 0
        self.freeze_encoder = bool(self.eng.cfg.freeze_encoder)
        self._one = torch.ones(1, dtype=torch.float32, device=g.device)
        self.last_ga_loss = None
        # dist: an object with torch.distributed's interface (get_world_size / broadcast / all_reduce(async_op=True) -> work.wait());
        # default: torch.distributed when a process group is initialised; False: never (bench.py's single-GPU leg inside a multi-rank run).  (tests/test_gpu_dp_race.py passes an asynchronous stand-in.)
        if dist is False:                                  # explicitly local: no exchange even when a process group is initialised
            self.dist = None
        elif dist is not None:
            self.dist = dist
        else:
            self.dist = torch.distributed if (torch.distributed.is_available() and torch.distributed.is_initialized()) else None
        self.world = self.dist.get_world_size() if self.dist else 1
        self.bucketer = None
        self.dp_mode = "allreduce"
        self._hook = _HOOK_T(self._on_stage)      # keep a reference: ctypes callbacks must outlive their use
        if self.world > 1 or (self.dist and os.environ.get("B2S_FORCE_DP")):    # B2S_FORCE_DP: 1-rank group, test aid
            # grad_payload "bf16" (default for world > 1): gradients travel as bf16 -- 167 MB instead of 334 MB per step over xGMI,
            # whose point-to-point links (7 x ~153 GB/s per GPU) bound the ring all-reduce -- converted by HIP
            # kernels on the exchange stream around each bucket's all-reduce; everything inside a rank (accumulation, Adam moments,
            # masters) stays fp32.  "fp32": the exact mean of the rank gradients (the reference's DDP arithmetic).
            # Default "fp32" (the reference's arithmetic: a drop-in must not change what DDP computes); B2S_GRAD_PAYLOAD overrides the
            # default, the constructor argument overrides both (bench.py selects "bf16" for its bf16 performance lines and says so;
            # tests/test_gpu_dp_race.py::test_bf16_wire_tracks_fp32_wire_over_200_steps is the convergence-level evidence for it).
            payload = grad_payload or os.environ.get("B2S_GRAD_PAYLOAD", "fp32")
            # the exchange's channel kernels hold CUs while the backward (and the next forward) runs: GEMM tiles that do not need all 256
            # CUs for a whole round (b2s_gemm_set_tile_policy; measured with tools/cu_loss.py, profiles/NOTES_r03.md)
            if self.world > 1:
                L.check(self.lib.b2s_gemm_set_tile_policy(4))
                self._set_tile_policy = True               # (process-wide switch: close() restores the per-shape choice)
            lib = self.lib
            def pack(src, dst):
                L.check(lib.b2s_cast(1, src.data_ptr(), dst.data_ptr(), src.numel(), L.stream()))
            def unpack(src, dst):
                L.check(lib.b2s_cast_back(1, src.data_ptr(), dst.data_ptr(), src.numel(), L.stream()))
            # dp_mode "rs_ag" (B2S_DP_MODE=rs_ag): reduce-scatter of the gradient buckets, Adam on this rank's 1/world slice of every bucket, all-gather
            # of the updated parameters through a flat fp32 wire (masters stay replicated).  Default "allreduce" (DESIGN.md (e) prices both).
            self.dp_mode = dp_mode or os.environ.get("B2S_DP_MODE", "allreduce")
            if self.dp_mode not in ("allreduce", "rs_ag"):
                raise ValueError("dp_mode must be 'allreduce' or 'rs_ag'")
            if self.dp_mode == "rs_ag" and self.freeze_encoder:
                raise L.B2SError("dp_mode='rs_ag' with a frozen encoder is not supported (its stages never report: the bucket plan is not static)")
            on_gpu = self.eng._gflat.is_cuda
            # the collectives are launched from a stream of their own: the engine orders it behind each stage's gradient work,
            # the backward pass itself never waits for the second stream on their account
            # -- the engine's own second stream: a stage's gradients are complete exactly there (its weight-gradient group is the stage's last
            # work), so the pack kernel and the collective's launch need no extra ordering, and the process stays at four active streams
            # (main, second, encoder, RCCL's).  A fifth one (a dedicated exchange stream: the round-3 layout)
            # oversubscribes the hardware queues: 12.6 ms per step instead of 7.9 with the encoder on its own stream (profiles/NOTES_r04.md)
            self._hook_stream = None
            if on_gpu:
                aux = self.lib.b2s_model_second_stream(self.eng.handle)
                if aux:
                    self._hook_stream = torch.cuda.ExternalStream(aux, device=self.eng._gflat.device)
                else:
                    self._hook_stream = torch.cuda.Stream(device=self.eng._gflat.device)
            # bf16 payload: the fused Adam reads the all-reduced gradients straight from the wire buffer (no unpack kernel, no fp32 re-read of
            # 334 MB: ~0.15 ms per step at N > 1)
            consume = payload == "bf16" and on_gpu
            self.bucketer = GradBucketer(self.eng._gflat, self.eng.stage_ranges, self.eng.n_stages(),
                                         bucket_mb * 1024 * 1024 / 4, dist=self.dist, payload=payload,
                                         pack=pack if on_gpu else None, unpack=unpack if on_gpu else None, stream=self._hook_stream,
                                         consume_wire=consume, mode=self.dp_mode)
            enc0 = self.eng.stage_ranges.get(3 + self.eng.cfg.n_decoder_layer)            # first encoder stage: the flat buffer's decoder | encoder boundary
            if self.tail_adam and not self.freeze_encoder and enc0 is not None and enc0[0] > 0 and self.dp_mode == "allreduce":
                self.bucketer.split = enc0[0]
            if self.dp_mode == "rs_ag":
                # every bucket is cut into `world` equal slices of whole 8-element groups; the flat layout's slots are 64-element aligned, so
                # world sizes 1, 2, 4 and 8 always divide -- anything else is refused here, with the reason, not inside a collective
                for lo, hi in self.bucketer.plan():
                    if (hi - lo) % (8 * self.bucketer.world):
                        raise ValueError("dp_mode='rs_ag' at world size %d: gradient bucket [%d, %d) of %d elements does not split into %d equal "
                                         "slices of a multiple of 8 elements (parameter slots are 64-element aligned: world sizes 1, 2, 4, 8 "
                                         "always work); use dp_mode='allreduce'" % (self.bucketer.world, lo, hi, hi - lo, self.bucketer.world))
                own = self.bucketer.owned_ranges()
                lo = (C.c_int64 * len(own))(*[a for a, _ in own])
                hi = (C.c_int64 * len(own))(*[b for _, b in own])
                L.check(self.lib.b2s_adam_shard(self.eng.handle, self.eng._gflat.data_ptr(), lo, hi, len(own)))
                self.param_wire = torch.empty_like(self.eng._gflat)
            if consume:
                L.check(self.lib.b2s_adam_set_grad_wire(self.eng.handle, self.bucketer.wire.data_ptr(), self.eng._gflat.data_ptr()))
            if self.world > 1 and self.dist.get_rank() == 0:
                import sys
                print("[b2s] data-parallel gradient payload: %s%s (HipTrainer(grad_payload=...) / B2S_GRAD_PAYLOAD; fp32 = the reference's exact "
                      "mean of the rank gradients, train.py:125)" % (payload, ", consumed by the optimizer from the wire buffer" if consume else ""),
                      file=sys.stderr)
            L.check(self.lib.b2s_model_set_stage_hook(self.eng.handle, C.cast(self._hook, L.P), None,
                                                      self._hook_stream.cuda_stream if self._hook_stream is not None else None))
            # broadcast parameters and BN buffers from rank 0 once (DDP constructor semantics, train.py:125)
            with torch.no_grad():
                for t in self.eng._tensors():
                    self.dist.broadcast(t.data, 0)
            self.eng._versions = None
            self.eng.ensure_bound()                   # re-sync the compute-dtype shadows with the broadcast values
        # DDP(broadcast_buffers=True) re-broadcasts rank 0's BatchNorm buffers before EVERY forward (train.py:125): one 17 KB message
        # per step.  B2S_BN_BROADCAST=0 keeps the running statistics rank-local between steps (rank 0's -- the ones the reference
        # checkpoints -- are identical either way).
        self._bn_buffers = [b for _, b in model.named_buffers()]
        self.bn_broadcast = self.world > 1 and os.environ.get("B2S_BN_BROADCAST", "1") != "0"

    # ------------------------------------------------------------------ checkpoint interchange (utils/checkpoint.py)
    def _param_names(self):
        return [n for n, _ in self.model.named_parameters()]           # = torch.optim.Adam(m.parameters()) index order

    def sync(self):
        """Kept for callers written against the overlapped-optimizer schedules: every update now runs on the stream train_step was
        called on, so plain stream order already covers it (no-op)."""

    def close(self):
        """Undo the process-wide settings this trainer made (the data-parallel GEMM tile policy); the trainer stays usable."""
        if getattr(self, "_set_tile_policy", False):
            L.check(self.lib.b2s_gemm_set_tile_policy(0))
            self._set_tile_policy = False

    def gather_optimizer_state(self):
        """dp_mode='rs_ag': COLLECTIVE -- call on EVERY rank before rank 0 serialises the optimizer (state_dict / utils.checkpoint.save_model).
        The sharded update advances exp_avg / exp_avg_sq only on the slices a rank owns; this all-gathers both moment buffers in place, bucket by
        bucket, exactly as all_gather_params does for the parameters, so that every rank holds the whole Adam state of the current step.
        A no-op in every other mode."""
        if self.bucketer is None or self.bucketer.mode != "rs_ag":
            return
        works = []
        for buf in (self.exp_avg, self.exp_avg_sq):
            for lo, hi in self.bucketer.plan():
                a, b = self.bucketer.shard_of(lo, hi)
                works.append(self.dist.all_gather_into_tensor(buf[lo:hi], buf[a:b], group=self.bucketer.group, async_op=True))
        for w in works:
            w.wait()
        self._moments_gathered_at = self.global_step

    def state_dict(self):
        """Optimizer state in torch.optim.Adam.state_dict() layout (what train.py:130 + checkpoint.py:27 write), so a
        checkpoint written from the fused trainer resumes under the reference loop and vice versa.  Parameters that
        were never updated (no step yet, frozen encoder) have no entry, as in torch."""
        self.sync()
        if (self.bucketer is not None and self.bucketer.mode == "rs_ag" and self.bucketer.world > 1 and self.global_step > 0 and
                getattr(self, "_moments_gathered_at", None) != self.global_step):
            raise L.B2SError("dp_mode='rs_ag': this rank holds the Adam moments of its own 1/%d slices only -- a checkpoint written now would resume "
                             "with zero moments everywhere else.  Call trainer.gather_optimizer_state() on EVERY rank (it is a collective) after "
                             "the step, then state_dict() / utils.checkpoint.save_model on the saving rank." % self.bucketer.world)
        names = self._param_names()
        state = {}
        if self.global_step > 0:
            for i, n in enumerate(names):
                if self.freeze_encoder and n.startswith("encoder."):
                    continue
                off, cnt = self.eng.param_offsets[n]
                shape = self.eng.grad_view(n).shape
                state[i] = {"step": torch.tensor(float(self.global_step)),
                            "exp_avg": self.exp_avg[off:off + cnt].view(shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + cnt].view(shape).clone()}
        lr = self.hp.max_lr * self.lr_lambda(self.global_step)
        group = {"lr": lr, "betas": (self.beta1, self.beta2), "eps": self.hp.adam_eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "initial_lr": self.hp.max_lr, "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        self.sync()
        names = self._param_names()
        groups = sd["param_groups"]
        order = [i for g in groups for i in g["params"]]
        if len(order) != len(names):
            raise ValueError("optimizer state covers %d parameters, the model has %d" % (len(order), len(names)))
        if groups and "betas" in groups[0]:
            self.beta1, self.beta2 = (float(b) for b in groups[0]["betas"])
        steps = set()
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for pos, idx in enumerate(order):
            st = sd["state"].get(idx)
            if not st:
                continue
            off, cnt = self.eng.param_offsets[names[pos]]
            self.exp_avg[off:off + cnt].copy_(st["exp_avg"].reshape(-1).to(self.exp_avg))
            self.exp_avg_sq[off:off + cnt].copy_(st["exp_avg_sq"].reshape(-1).to(self.exp_avg_sq))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("per-parameter Adam step counts differ (%s): not a state the fused optimizer can resume" % sorted(steps))
        self.global_step = steps.pop() if steps else 0
        self.eng._versions = None          # parameters were (re)loaded alongside: refresh the bf16 shadows on the next step

    @property
    def sched(self):
        """LambdaLR-shaped view of the trainer's step counter for utils.checkpoint.save_model / load_model."""
        return _SchedView(self)

    # ------------------------------------------------------------------ gradient exchange
    def _on_stage(self, stage, _user):
        # called from C through ctypes: an exception escaping here would be printed and swallowed, the bucket never reduced
        # and the step applied to un-averaged gradients -- keep it and re-raise from train_step before the optimizer runs
        try:
            if self.bucketer is not None and self._hook_error is None:
                self.bucketer.stage_done(stage)
        except BaseException as e:          # noqa: B902 (must not propagate into ctypes)
            self._hook_error = e

    def _consume_partial_step(self, step_no, cause, what=None):
        """A failure AFTER the decoder / postnet update of step `step_no` was issued (a collective of the encoder buckets, the final
        update): those groups are at step_no, the encoder group is not.  The step cannot be retried under the same number (Adam's bias
        correction of the updated groups has advanced; b2s_adam_step_groups refuses a second update of a group in one step), so it is
        marked consumed -- the next train_step runs as step_no + 1 on every group -- and the error says so."""
        self.global_step = step_no
        self.last_step_tail_update = True
        what = what or "its decoder / postnet optimizer update had been issued: a PARTIAL update was applied (encoder parameters keep their step-%d values)" % (step_no - 1)
        raise RuntimeError("training step %d failed after %s.  The step counter has advanced to %d so that "
                           "training can continue; restore a checkpoint if the replicas must stay bit-identical." %
                           (step_no, what, step_no)) from cause

    # ------------------------------------------------------------------ one step
    def train_step(self, batch):
        """batch: the dataloader dict (dataloader.py:498-508) on the device.  Returns the 7 loss values (device).
        Optional key "target_lengths_host": the host copy of batch["target_lengths"] (list / CPU tensor; the dataloader has it before the batch
        is sent to the device, b2s_hip/batching.py keeps it) -- the decoder segment then runs on ragged rows, sum(target_lengths) instead of
        B x T per row-wise kernel, without a device-to-host read."""
        eng, lib = self.eng, self.lib
        # Re-bind / re-sync on THIS stream before anything forks off it: a full weight sync (parameters loaded or changed behind the
        # optimizer's back: utils.checkpoint.load_model, load_state_dict) re-casts every bf16 shadow, and both the encoder stream and
        # the decoder's first kernels on this stream read shadows -- left to the encoder_forward call below it would run on the
        # encoder stream only, unordered against the decoder prenet / first self-attention.
        eng.ensure_bound()
        if self.bn_broadcast and self._bn_buffers:
            from .dp import broadcast_buffers
            broadcast_buffers(self._bn_buffers, self.dist, 0)
        # the fused Adam kernel rewrote the fp32 masters AND their bf16 shadows; only conv re-layouts remain
        L.check(lib.b2s_model_sync_weights(eng.handle, L.stream(), int(self.global_step > 0)))
        # (the stager's int32 device copies when the batch carries them: two cast kernels less at the head of the step's critical path)
        in32 = batch["input_lengths_i32"] if "input_lengths_i32" in batch else _i32(batch["input_lengths"])
        tgt32 = batch["target_lengths_i32"] if "target_lengths_i32" in batch else _i32(batch["target_lengths"])
        cur = torch.cuda.current_stream()
        ovl = self.overlap_encoder
        if ovl and self._enc_stream is None:
            # ONE encoder stream per device and process (as the library's second stream, engine.hip: aux_stream_of): HIP assigns a stream its hardware
            # queue at creation, and a fresh stream per trainer put the third trainer of a process on the main stream's queue (13.3 instead of 6.5 ms)
            self._enc_stream = encoder_stream(eng._gflat.device)
            # idle during the decoder backward: the memory-side dK / dV kernels of the encoder-decoder attentions run there (include/b2s_hip.h)
            L.check(lib.b2s_model_set_side_stream(eng.handle, C.c_void_p(self._enc_stream.cuda_stream) if self.side_stream else None))
        enc_s = self._enc_stream if ovl else None
        if enc_s is not None:
            enc_s.wait_stream(cur)                         # (weights synced above; the previous step's optimizer update)
            with torch.cuda.stream(enc_s):
                mem, c_enc = eng.encoder_forward(batch["inputs"], in32, batch.get("input_spk_ids"), batch.get("input_language_vecs"),
                                                 True, eng.next_seed("encoder"), not self.freeze_encoder)
                mem_ready = torch.cuda.Event()
                mem_ready.record(enc_s)
            mels, stop, c_dec = eng.decoder_forward(mem, in32, batch["mel_targets"], tgt32, True, eng.next_seed("decoder"), True, memory_ready=mem_ready, padded_unobserved=True,
                                                    target_lengths_host=batch.get("target_lengths_host"))
        else:
            mem, c_enc = eng.encoder_forward(batch["inputs"], in32, batch.get("input_spk_ids"), batch.get("input_language_vecs"),
                                             True, eng.next_seed("encoder"), not self.freeze_encoder)
            mels, stop, c_dec = eng.decoder_forward(mem, in32, batch["mel_targets"], tgt32, True, eng.next_seed("decoder"), True, padded_unobserved=True,
                                                    target_lengths_host=batch.get("target_lengths_host"))
        aft, c_post = eng.postnet_forward(mels, tgt32, mels, True, eng.next_seed("postnet"), True)
        vals, per = eng.loss_forward(mels, aft, stop, batch["mel_targets"], tgt32)
        guided = eng.guided_enabled()
        if guided:
            self.last_ga_loss = eng.guided_loss(c_dec, add_to=vals)       # vals[0] (total loss) += weight * guided loss
        # (1 = B2S_ZERO_GRADS_OVERWRITE_DW: this step runs every backward segment exactly once)
        eng._reclaim_lent()                                  # (a .grad the autograd path lent out of the flat buffer gets its own copy first)
        L.check(lib.b2s_zero_grads(eng.handle, L.stream(), 1))
        eng._needs_zero = False
        if self.bucketer is not None:
            self.bucketer.begin_step()
        self._hook_error = None
        partial = False                                      # the decoder / postnet part of this step's optimizer update has been issued
        step_no = self.global_step + 1
        try:
            dbef, daft, dstop = eng.loss_backward(mels, aft, stop, batch["mel_targets"], tgt32, None)
            din = eng.postnet_backward(c_post, daft, defer_join=True)           # (the decoder backward below takes over the second stream's join)
            dmel = eng.add3(din, daft, dbef)                  # (same order of additions as two b2s_add calls)
            enc_bwd_s = enc_s if (enc_s is not None and not self.freeze_encoder) else None
            dmem_done = None
            if enc_bwd_s is not None:
                dmem_done = torch.cuda.Event()
                dmem_done.record(cur)                          # (torch creates the HIP event at its first record: the library re-records this handle)
            dmem = eng.decoder_backward(c_dec, dmel, dstop, mem.shape, self._one if guided else None, not self.freeze_encoder,
                                        defer_join=not self.freeze_encoder, dmem_done=dmem_done)    # (encoder_backward below joins the second stream)
            lr = self.hp.max_lr * self.lr_lambda(self.global_step)
            step_no = self.global_step + 1
            adam = (lr, step_no, self.beta1, self.beta2, self.hp.adam_eps, self.hp.reg_weight, 1.0 / self.world)
            self._last_adam = adam                          # (tests: the stand-in process group of test_gpu_rs_ag.py replays the peer's shard update)
            # (data parallel: the same schedule once the decoder / postnet buckets' all-reduces are complete -- GradBucketer.split keeps a
            # bucket from straddling the group boundary, wait_prefix orders this stream behind exactly those collectives)
            tail = (self.tail_adam and enc_bwd_s is not None and
                    (self.bucketer is None or getattr(self.bucketer, "split", None) is not None))
            if tail:
                L.check(lib.b2s_model_mark_grads_ready(eng.handle))     # decoder / postnet gradients are final behind this point of the second stream
            if not self.freeze_encoder:
                if enc_bwd_s is not None:
                    enc_bwd_s.wait_event(dmem_done)            # d(memory) only: the rest of the decoder backward runs beside the encoder's
                    with torch.cuda.stream(enc_bwd_s):
                        eng.encoder_backward(c_enc, dmem)
                    if tail and self.bucketer is not None:
                        # Every stage has reported by now (the encoder backward's drain fired the last hooks).  Everything that can be
                        # checked WITHOUT waiting is checked before any part of the update is issued: a hook that failed, a stage that did
                        # not report, ranges that do not tile the buffer -- each falls back to the single update after the whole exchange
                        # (which then refuses the step as a whole).  What remains after the partial update is the collectives themselves.
                        tail = (self._hook_error is None and self.bucketer.covers_all(expect_all=not self.freeze_encoder) and
                                self.bucketer.wait_prefix(self.bucketer.split))
                    if tail:
                        # issued only now (a failure in the encoder backward call above leaves the step unapplied), but ordered behind the
                        # mark only: on the device it runs beside the encoder backward
                        L.check(lib.b2s_adam_step_groups(eng.handle, *adam, 2 | 4, 1, L.stream()))
                        partial = True
                    cur.wait_stream(enc_bwd_s)                 # (its last stage joined the engine's second stream)
                else:
                    eng.encoder_backward(c_enc, dmem)
            elif enc_s is not None:
                cur.wait_stream(enc_s)
        except BaseException as e:
            # the backward entry points hand work to each other (deferred joins): whatever is still queued points into contexts that are
            # freed below -- drop it and rejoin the second stream before anything else touches the engine (b2s_model_backward_abort)
            try:
                if enc_s is not None:
                    cur.wait_stream(enc_s)
                lib.b2s_model_backward_abort(eng.handle, L.stream())
                if self.bucketer is not None:
                    self.bucketer.abort()
            finally:
                for c in (c_post, c_dec, c_enc):
                    if c is not None:
                        c.free()
                eng._needs_zero = True
            if partial:
                self._consume_partial_step(step_no, e)
            raise
        for c in (c_post, c_dec, c_enc):
            if c is not None:
                c.free()
        shard_updated = False
        try:
            if self._hook_error is not None:
                err, self._hook_error = self._hook_error, None
                raise RuntimeError("gradient exchange failed in the backward stage hook; the optimizer step was NOT applied") from err
            if self.bucketer is not None:
                self.bucketer.finish(expect_all=not self.freeze_encoder)
            if self.grad_probe is not None:
                self.grad_probe(eng._gflat, self.bucketer.wire if (self.bucketer is not None and self.bucketer.consume_wire) else None)
            if partial:
                L.check(lib.b2s_adam_step_groups(eng.handle, *adam, 1, 0, L.stream()))
            else:
                L.check(lib.b2s_adam_step(eng.handle, *adam, L.stream()))
            shard_updated = self.bucketer is not None and self.bucketer.mode == "rs_ag"     # this rank's slices are at step_no from here on
            if self.bucketer is not None and self.bucketer.mode == "rs_ag":
                # sharded update: this rank's slices -> parameter wire -> all-gather -> the other ranks' slices into masters / shadows
                L.check(lib.b2s_param_wire(eng.handle, self.param_wire.data_ptr(), 0, L.stream()))
                self.bucketer.all_gather_params(self.param_wire)
                L.check(lib.b2s_param_wire(eng.handle, self.param_wire.data_ptr(), 1, L.stream()))
        except BaseException as e:
            if self.bucketer is not None:
                self.bucketer.abort()
            eng._needs_zero = True
            if partial:
                self._consume_partial_step(step_no, e)
            if shard_updated:
                # rs_ag: the sharded Adam has run, the parameter exchange behind it has not completed -- a retry under the same step number would
                # apply Adam to the owned slices a second time while the other ranks' slices never arrived
                self._consume_partial_step(step_no, e, what="the sharded optimizer update of this rank's slices had been issued but before the "
                                           "all-gather of the updated parameters completed: masters and shadows of the OTHER ranks' slices are stale")
            raise
        self.global_step = step_no
        self.last_step_tail_update = bool(partial)         # (tests: which optimizer schedule the step took)
        eng._needs_zero = True
        self.last_aft_losses = per
        return vals
