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

import torch

from . import lib as L
from .engine import _i32
from .dp import GradBucketer

_HOOK_T = C.CFUNCTYPE(None, C.c_int, C.c_void_p)


class HipTrainer(object):
    def __init__(self, model, hp, beta1=0.9, beta2=0.999, bucket_mb=32.0):
        from transformer.tacotron import learning_rate_schedule
        self.model, self.hp = model, hp
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
        self.global_step = 0
        self.freeze_encoder = bool(self.eng.cfg.freeze_encoder)
        self._one = torch.ones(1, dtype=torch.float32, device=g.device)
        self.last_ga_loss = None
        self.dist = torch.distributed if (torch.distributed.is_available() and torch.distributed.is_initialized()) else None
        self.world = self.dist.get_world_size() if self.dist else 1
        self.bucketer = None
        self._hook = _HOOK_T(self._on_stage)      # keep a reference: ctypes callbacks must outlive their use
        if self.world > 1 or (self.dist and os.environ.get("B2S_FORCE_DP")):    # B2S_FORCE_DP: 1-rank group, test aid
            self.bucketer = GradBucketer(self.eng._gflat, self.eng.stage_ranges, self.eng.n_stages(),
                                         bucket_mb * 1024 * 1024 / 4, dist=self.dist)
            L.check(self.lib.b2s_model_set_stage_hook(self.eng.handle, C.cast(self._hook, L.P), None))
            # broadcast parameters and BN buffers from rank 0 once (DDP constructor semantics, train.py:125)
            with torch.no_grad():
                for t in self.eng._tensors():
                    self.dist.broadcast(t.data, 0)
            self.eng._versions = None
            self.eng.ensure_bound()                   # re-sync the compute-dtype shadows with the broadcast values

    # ------------------------------------------------------------------ gradient exchange
    def _on_stage(self, stage, _user):
        if self.bucketer is not None:
            self.bucketer.stage_done(stage)

    # ------------------------------------------------------------------ one step
    def train_step(self, batch):
        """batch: the dataloader dict (dataloader.py:498-508) on the device.  Returns the 7 loss values (device)."""
        eng, lib = self.eng, self.lib
        # the fused Adam kernel rewrote the fp32 masters AND their bf16 shadows; only conv re-layouts remain
        L.check(lib.b2s_model_sync_weights_ex(eng.handle, L.stream(), int(self.global_step > 0)))
        in32, tgt32 = _i32(batch["input_lengths"]), _i32(batch["target_lengths"])
        mem, c_enc = eng.encoder_forward(batch["inputs"], in32, batch.get("input_spk_ids"), batch.get("input_language_vecs"),
                                         True, eng.next_seed(), not self.freeze_encoder)
        mels, stop, c_dec = eng.decoder_forward(mem, in32, batch["mel_targets"], tgt32, True, eng.next_seed(), True)
        aft, c_post = eng.postnet_forward(mels, tgt32, mels, True, eng.next_seed(), True)
        vals, per = eng.loss_forward(mels, aft, stop, batch["mel_targets"], tgt32)
        guided = eng.guided_enabled()
        if guided:
            self.last_ga_loss = eng.guided_loss(c_dec, add_to=vals)       # vals[0] (total loss) += weight * guided loss
        L.check(lib.b2s_zero_grads(eng.handle, L.stream()))
        eng._needs_zero = False
        if self.bucketer is not None:
            self.bucketer.begin_step()
        dbef, daft, dstop = eng.loss_backward(mels, aft, stop, batch["mel_targets"], tgt32, None)
        din = eng.postnet_backward(c_post, daft)
        dmel = eng.add(eng.add(din, daft), dbef)
        dmem = eng.decoder_backward(c_dec, dmel, dstop, mem.shape, self._one if guided else None, not self.freeze_encoder)
        if not self.freeze_encoder:
            eng.encoder_backward(c_enc, dmem)
        for c in (c_post, c_dec, c_enc):
            if c is not None:
                c.free()
        if self.bucketer is not None:
            self.bucketer.finish()
        lr = self.hp.max_lr * self.lr_lambda(self.global_step)
        self.global_step += 1
        L.check(lib.b2s_adam_step(eng.handle, lr, self.global_step, self.beta1, self.beta2, self.hp.adam_eps,
                                  self.hp.reg_weight, 1.0 / self.world, L.stream()))
        eng._needs_zero = True
        self.last_aft_losses = per
        return vals
