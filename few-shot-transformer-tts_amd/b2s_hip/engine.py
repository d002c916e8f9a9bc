"""Host glue between the nn.Module surface (transformer/tacotron.py) and libb2s_hip.so.

HipEngine owns one b2s_model handle for a Tacotron module tree: it binds the device pointers of the
module's parameters/buffers (re-binding when .to()/.cuda() replaced them, re-syncing the compute-dtype
weight shadows when an optimizer changed them in place), owns the flat fp32 gradient buffer the HIP
backward kernels accumulate into, and exposes each model segment as a torch.autograd.Function so
`loss.backward()` / `optim.step()` in a train.py-style driver work unchanged.  PyTorch only provides
device memory, streams and the autograd graph edges; no math runs in torch.
"""
import ctypes as C

import os
import torch

from . import lib as L

DTYPES = {"fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1}
# extensions absent from the reference's hyperparams.py: an HParams object without them means "off"
_EXT_DEFAULTS = {"guided_attention_weight": 0.0, "guided_attention_sigma": 0.2, "freeze_encoder": False}


def config_from_hparams(hp):
    cfg = L.Config()
    for name, _ in L.Config._fields_:
        if name in ("compute_dtype",):
            continue
        v = getattr(hp, name, _EXT_DEFAULTS[name]) if name in _EXT_DEFAULTS else getattr(hp, name)
        setattr(cfg, name, int(v) if isinstance(v, bool) else v)
    cd = getattr(hp, "compute_dtype", "fp32")
    if cd not in DTYPES:
        raise ValueError("compute_dtype must be one of %s" % sorted(DTYPES))
    cfg.compute_dtype = DTYPES[cd]
    return cfg


def _i32(t):
    """lengths as contiguous int32 on the device (the batch dict holds int64, eval_batch int32)."""
    if t.dtype != torch.int32:
        t = t.to(torch.int32)
    return t.contiguous()


class _Ctx(object):
    """Owns a b2s_ctx handle plus every tensor the HIP side still points into."""

    def __init__(self, handle, keep):
        self.handle = handle
        self.keep = keep

    def free(self):
        if self.handle:
            L.load().b2s_ctx_free(self.handle)
            self.handle = None
        self.keep = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class HipEngine(object):
    def __init__(self, root, hp):
        self.lib = L.load()
        self.root = root
        self.hp = hp
        self.cfg = config_from_hparams(hp)
        self.dtype = self.cfg.compute_dtype
        h = L.P()
        L.check(self.lib.b2s_model_create(C.byref(self.cfg), C.byref(h)))
        self.handle = h
        n = self.lib.b2s_model_num_tensors(h)
        self.names, self.shapes, self.kinds = [], [], []
        buf = C.create_string_buffer(256)
        shape = (C.c_int64 * 8)()
        nd, kind = C.c_int(), C.c_int()
        for i in range(n):
            L.check(self.lib.b2s_model_tensor_info(h, i, buf, 256, shape, C.byref(nd), C.byref(kind)))
            self.names.append(buf.value.decode())
            self.shapes.append(tuple(shape[k] for k in range(nd.value)))
            self.kinds.append(kind.value)
        sd = root.state_dict()
        mine = [(k, tuple(v.shape)) for k, v in sd.items()]
        theirs = list(zip(self.names, self.shapes))
        if mine != theirs:
            raise L.B2SError("module state_dict layout differs from the HIP model layout")
        self._sig = None
        self._slots = None
        self._versions = None
        self._gflat = None
        self._gviews = {}
        self._lent = []                 # parameters whose .grad may be a lent view of the flat gradient buffer (grad_out)
        self._needs_zero = True
        self._bwd_seen = set()            # segment kinds back-propagated since the gradient buffer was last zeroed (autograd path)
        self._trainer = None              # weakref to an attached HipTrainer (its Adam moments mirror the flat gradient layout)
        self._seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFF
        self._calls = 0
        self.seeds_used = {}
        self.index = {n_: i for i, n_ in enumerate(self.names)}

    def __del__(self):
        try:
            if self.handle:
                self.lib.b2s_model_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ------------------------------------------------------------------ binding
    def _tensors(self):
        # (owning module's _parameters / _buffers dict, attribute) per bound tensor, resolved once: walking
        # named_parameters() on every call cost ~0.8 ms x 4 calls per training step, a third of the host's launch time.
        # Looking the attribute up each time still sees a Parameter object that was replaced on its module.
        if self._slots is None:
            mods = dict(self.root.named_modules())
            slots = []
            for n in self.names:
                owner, _, attr = n.rpartition(".")
                m = mods[owner]
                slots.append((m._parameters if attr in m._parameters else m._buffers, attr))
            self._slots = slots
        return [d[a] for d, a in self._slots]

    def ensure_bound(self):
        ts = self._tensors()
        dev = ts[0].device
        if dev.type != "cuda":
            raise L.B2SError("model parameters are on %s: the byte2speech hot path has no CPU fallback; "
                             "move the model to a HIP device (model.cuda())" % dev)
        sig = tuple(t.data_ptr() for t in ts)
        if sig != self._sig:
            for t, n, k in zip(ts, self.names, self.kinds):
                want = torch.int64 if k == 2 else torch.float32
                if t.dtype != want or not t.is_contiguous() or t.device != dev:
                    raise L.B2SError("tensor %s must be contiguous %s on %s" % (n, want, dev))
            # flat gradient layout = backward execution order, so that every backward stage owns one contiguous
            # range (a data-parallel bucket) that is complete when b2s_model_set_stage_hook fires for it.  Every tensor starts on a
            # 256-byte boundary (its slot is padded with zeros that nothing reads): without that the three 1-element parameters
            # (stop_net.bias, the two pe_scale) left 95 % of the gradient -- and with it the Adam moments and the bf16 wire buffer of
            # the gradient exchange -- off the 16-byte grid that the vector paths of the optimizer, the weight-gradient epilogues
            # and the collectives want
            order = sorted((i for i, k in enumerate(self.kinds) if k == 1), key=lambda i: (self.stage_of(self.names[i]), i))
            al = 64                                                           # elements (256 bytes); the packed layout measured +0.21 ms per step
            slot = lambda n: (n + al - 1) // al * al
            total = sum(slot(ts[i].numel()) for i in order)
            if self._gflat is None or self._gflat.device != dev:
                if self._gflat is not None and self._trainer is not None and self._trainer() is not None:
                    raise L.B2SError("the model moved from %s to %s under an attached HipTrainer: its optimizer state lives on "
                                     "the old device; build a new HipTrainer (load_state_dict carries the state over)"
                                     % (self._gflat.device, dev))
                self._gflat = torch.zeros(total, dtype=torch.float32, device=dev)
            self._gviews = {}
            data = (L.P * len(ts))(*[t.data_ptr() for t in ts])
            grads = (L.P * len(ts))()
            off = 0
            self.stage_ranges = {}
            self.param_offsets = {}
            for i in order:
                t, n = ts[i], self.names[i]
                v = self._gflat[off:off + t.numel()].view(t.shape)
                self._gviews[n] = v
                grads[i] = v.data_ptr()
                self.param_offsets[n] = (off, t.numel())
                st = self.stage_of(n)
                lo, hi = self.stage_ranges.get(st, (off, off))
                self.stage_ranges[st] = (min(lo, off), off + slot(t.numel()))
                off += slot(t.numel())
            L.check(self.lib.b2s_model_bind(self.handle, data, grads, len(ts)))
            L.check(self.lib.b2s_model_set_grad_slot_padding(self.handle, 4 * (al - 1) if al > 1 else 0))   # the gaps between the slots above
            self._sig = sig
            self._versions = None
        # parameters only: buffers (BatchNorm running statistics) have no compute-dtype shadow -- the kernels read them in place -- and the
        # data-parallel trainer rewrites them before every step (broadcast_buffers), which must not trigger a re-cast of every weight
        vers = tuple(t._version for t, k in zip(ts, self.kinds) if k == 1)
        if vers != self._versions:
            L.check(self.lib.b2s_model_sync_weights(self.handle, L.stream(), 0))
            self._versions = vers
        return dev

    def n_stages(self):
        return 5 + self.cfg.n_decoder_layer + self.cfg.n_encoder_layer

    def stage_of(self, name):
        """Backward stage that completes the gradient of parameter `name` (see b2s_model_set_stage_hook)."""
        Ld, Le = self.cfg.n_decoder_layer, self.cfg.n_encoder_layer
        parts = name.split(".")
        if parts[0] == "postnet":
            return 0
        if parts[0] == "decoder":
            if parts[1] in ("mel_net", "stop_net"):
                return 1
            if parts[1] == "prenet":
                return 2 + Ld
            if parts[2] == "output_layer_norm":
                return 1
            if parts[2] == "pe_scale":
                return 2 + Ld
            return 2 + (Ld - 1 - int(parts[3]))
        if parts[1] in ("speaker_embed", "speaker_layer", "language_embed", "language_layer"):
            return 3 + Ld
        if parts[1] == "embed":
            return 4 + Ld + Le
        if parts[2] == "output_layer_norm":
            return 3 + Ld
        if parts[2] == "pe_scale":
            return 4 + Ld + Le
        return 4 + Ld + (Le - 1 - int(parts[3]))

    def next_seed(self, segment=None):
        """Seed of the next segment call's dropout masks.  seeds_used[segment] keeps the last one handed out per segment ("encoder",
        "decoder", "postnet", "decode"): with it and b2s_dropout_site a checker can regenerate every mask of the step."""
        self._calls += 1
        seed = (self._seed * 0x9E3779B97F4A7C15 + self._calls * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        if segment is not None:
            self.seeds_used[segment] = seed
        return seed

    def begin_backward(self):
        if self._needs_zero:
            self._reclaim_lent()
            L.check(self.lib.b2s_zero_grads(self.handle, L.stream(), 0))
            self._needs_zero = False
            self._bwd_seen = set()

    def _reclaim_lent(self):
        """Before the flat gradient buffer is cleared for a new backward pass: a parameter whose .grad still IS a view of that buffer (grad_out lent it and
        the caller has not dropped it: gradient accumulation over several backward passes, a loop that zeroes with set_to_none=False) gets a copy of
        its own -- the semantics of the cloned gradients autograd would have kept.  The reference's loop (train.py:171-174: zero_grad sets .grad to
        None before every backward) never gets here with anything to copy."""
        lent, self._lent = self._lent, []
        lo = self._gflat.data_ptr() if self._gflat is not None else 0
        hi = lo + (self._gflat.numel() * 4 if self._gflat is not None else 0)
        for p in lent:
            g = p.grad
            if g is not None and lo <= g.data_ptr() < hi:
                p.grad = g.clone()

    def _claim_backward(self, kind):
        """Autograd path: the segment functions hand autograd VIEWS of the one flat gradient buffer, which is zeroed once per
        backward pass (first backward call after a forward).  Two forward passes back-propagated in one pass --
        (lossA + lossB).backward() -- would accumulate both into the same range and return it twice (2 x (A + B)), and
        backward(retain_graph=True) twice likewise: refuse instead of returning wrong gradients.  The reference loop
        (train.py:171-174: one forward, one backward, one step) never does either."""
        if kind in self._bwd_seen:
            raise L.B2SError("second backward of the %s segment into the same gradient buffer: back-propagate each forward pass "
                             "on its own (one forward per backward; retain_graph re-use is not supported)" % kind)
        self._bwd_seen.add(kind)

    def grad_view(self, name):
        return self._gviews[name]

    def grad_out(self, name, param):
        """The gradient a segment function hands to autograd.  A parameter without a .grad gets a FRESH view of the flat buffer: nobody else holds that
        tensor object, so autograd keeps the view as .grad instead of cloning it (the cached views were cloned: 162 copy kernels and 334 MB per step
        of the reference's loop, 0.6 of its 9.8 ms); the view is reclaimed -- replaced by a copy -- if it is still some .grad when the buffer is next
        cleared (_reclaim_lent).  A parameter that already has a .grad gets the cached view: autograd adds it to what is there."""
        v = self._gviews[name]
        if param is None or param.grad is not None or torch.is_grad_enabled():
            return v
        off, n = self.param_offsets[name]
        self._lent.append(param)
        return self._gflat[off:off + n].view(v.shape)

    # ------------------------------------------------------------------ segments (raw, no autograd)
    def encoder_forward(self, inputs, lens32, spk, lang, train, seed, keep_ctx):
        dev = self.ensure_bound()
        B, S = inputs.shape
        inputs = inputs.contiguous()
        spk = spk.contiguous() if spk is not None else None
        lang = lang.contiguous() if lang is not None else None
        nbytes = self.lib.b2s_encoder_ws_bytes(self.handle, B, S)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        mem = torch.empty(B, S, self.cfg.decoder_hidden, dtype=torch.float32, device=dev)
        h = L.P()
        L.check(self.lib.b2s_encoder_forward(self.handle, L.ptr(inputs), L.ptr(lens32), L.ptr(spk), L.ptr(lang), B, S,
                                             int(train), seed, L.ptr(ws), nbytes, L.ptr(mem), L.stream(),
                                             C.byref(h) if keep_ctx else None))
        self._needs_zero = True
        return mem, (_Ctx(h, (ws, inputs, lens32, spk, lang)) if keep_ctx else None)

    def encoder_backward(self, ctx, dmem):
        self.begin_backward()
        L.check(self.lib.b2s_encoder_backward(self.handle, ctx.handle, L.ptr(dmem.contiguous()), L.stream()))

    def decoder_forward(self, memory, in32, targets, tgt32, train, seed, keep_ctx, memory_ready=None, padded_unobserved=False, target_lengths_host=None):
        """memory_ready: torch.cuda.Event recorded behind the encoder forward on ANOTHER stream; this stream waits for it only when the
        decoder first reads `memory` (b2s_decoder_forward: memory_ready).
        padded_unobserved: the caller will not ask for alignments of query rows >= target length (B2S_DEC_PADDED_UNOBSERVED): the
        attention kernels skip tiles of padded query rows.
        target_lengths_host (with padded_unobserved): the HOST copy of target_lengths (sequence of B ints, the values of tgt32) -- the segment then
        keeps its rows ragged (b2s_decoder_compact_rows: sum(target_lengths) rows per row-wise kernel instead of B x T); B2S_COMPACT=0 ignores it."""
        dev = self.ensure_bound()
        B, T, NM = targets.shape
        if padded_unobserved and target_lengths_host is not None and os.environ.get("B2S_COMPACT", "1") != "0":
            lens = (C.c_int32 * B)(*[int(x) for x in target_lengths_host])
            L.check(self.lib.b2s_decoder_compact_rows(self.handle, lens, B))
        S = memory.shape[1]
        memory = memory.contiguous()
        targets = targets.contiguous()
        nbytes = self.lib.b2s_decoder_ws_bytes(self.handle, B, S, T)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        mels = torch.empty(B, T, NM, dtype=torch.float32, device=dev)
        stop = torch.empty(B, T, dtype=torch.float32, device=dev)
        h = L.P()
        L.check(self.lib.b2s_decoder_forward(self.handle, L.ptr(memory), L.ptr(in32), L.ptr(targets), L.ptr(tgt32), B, S, T,
                                                int(bool(train)) | (2 if padded_unobserved else 0), seed, L.ptr(ws), nbytes, L.ptr(mels), L.ptr(stop),
                                                memory_ready.cuda_event if memory_ready is not None else None, L.stream(), C.byref(h)))
        self._needs_zero = True
        return mels, stop, _Ctx(h, (ws, memory, in32, targets, tgt32))

    def decoder_backward(self, ctx, dmels, dstop, mem_shape, d_guided=None, want_dmem=True, defer_join=False, dmem_done=None):
        """d_guided: device scalar d loss / d guided-attention loss (None: that term gets no gradient).
        defer_join: the next engine call is encoder_backward (B2S_DEC_BWD_DEFER_JOIN); ctx must stay alive until it returns.
        dmem_done: torch.cuda.Event recorded as soon as d(memory) is complete; the encoder backward may then run on another stream that
        waits for it (the last stages' weight-gradient work is handed over by this call: B2S_DEC_BWD_FLUSH_TAIL)."""
        self.begin_backward()
        dmem = torch.empty(mem_shape, dtype=torch.float32, device=dmels.device) if want_dmem else None
        flags = (0 if want_dmem else 1) | ((4 if dmem_done is not None else 2) if defer_join else 0)
        L.check(self.lib.b2s_decoder_backward(self.handle, ctx.handle, L.ptr(dmels.contiguous()),
                                                 L.ptr(dstop.contiguous()) if dstop is not None else None,
                                                 L.ptr(d_guided.contiguous()) if d_guided is not None else None,
                                                 flags, L.ptr(dmem), dmem_done.cuda_event if dmem_done is not None else None, L.stream()))
        return dmem

    def guided_enabled(self):
        return self.cfg.guided_attention_weight > 0

    def guided_loss(self, ctx, add_to=None):
        """weight * guided-attention loss of the decoder forward held in ctx (device scalar, shape [1]); also added to
        add_to[0] when given."""
        out = torch.empty(1, dtype=torch.float32, device=ctx.keep[0].device)
        L.check(self.lib.b2s_decoder_guided_loss(self.handle, ctx.handle, L.ptr(out), L.ptr(add_to), L.stream()))
        return out

    def decoder_alignment(self, ctx, which, layer, B, H, Lk, Lq):
        out = torch.empty(B, H, Lk, Lq, dtype=torch.float32, device=ctx.keep[0].device)
        L.check(self.lib.b2s_decoder_alignment(self.handle, ctx.handle, which, layer, L.ptr(out), L.stream()))
        return out

    def postnet_forward(self, inputs, len32, add, train, seed, keep_ctx):
        dev = self.ensure_bound()
        B, T, NM = inputs.shape
        inputs = inputs.contiguous()
        nbytes = self.lib.b2s_postnet_ws_bytes(self.handle, B, T)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        out = torch.empty(B, T, NM, dtype=torch.float32, device=dev)
        h = L.P()
        L.check(self.lib.b2s_postnet_forward(self.handle, L.ptr(inputs), L.ptr(len32), L.ptr(add), B, T, int(train), seed,
                                             L.ptr(ws), nbytes, L.ptr(out), L.stream(), C.byref(h) if keep_ctx else None))
        self._needs_zero = True
        return out, (_Ctx(h, (ws, inputs, len32, add)) if keep_ctx else None)

    def postnet_backward(self, ctx, dout, defer_join=False):
        """defer_join: the next engine call is decoder_backward (B2S_POST_BWD_DEFER_JOIN); ctx must stay alive until the
        second stream has been joined (end of the decoder / encoder backward)."""
        self.begin_backward()
        din = torch.empty_like(dout)
        L.check(self.lib.b2s_postnet_backward(self.handle, ctx.handle, L.ptr(dout.contiguous()), L.ptr(din), 1 if defer_join else 0,
                                                 L.stream()))
        return din

    def add(self, a, b):
        out = torch.empty_like(a)
        L.check(self.lib.b2s_add3(L.ptr(a.contiguous()), L.ptr(b.contiguous()), None, L.ptr(out), a.numel(), L.stream()))
        return out

    def add3(self, a, b, c):
        out = torch.empty_like(a)
        L.check(self.lib.b2s_add3(L.ptr(a.contiguous()), L.ptr(b.contiguous()), L.ptr(c.contiguous()), L.ptr(out), a.numel(), L.stream()))
        return out

    def loss_forward(self, bef, aft, stop, tgt, len32):
        dev = self.ensure_bound()
        B, T, _ = tgt.shape
        vals = torch.empty(7, dtype=torch.float32, device=dev)
        per = torch.empty(B, dtype=torch.float32, device=dev)
        scratch = torch.empty(8 + B, dtype=torch.float32, device=dev)
        L.check(self.lib.b2s_loss_forward(self.handle, L.ptr(bef.contiguous()), L.ptr(aft.contiguous()),
                                          L.ptr(stop.contiguous()), L.ptr(tgt.contiguous()), L.ptr(len32), B, T, L.ptr(vals),
                                          L.ptr(per), L.ptr(scratch), L.stream()))
        return vals, per

    def loss_backward(self, bef, aft, stop, tgt, len32, w3):
        B, T, _ = tgt.shape
        dbef = torch.empty_like(bef)
        daft = torch.empty_like(aft)
        dstop = torch.empty_like(stop)
        L.check(self.lib.b2s_loss_backward(self.handle, L.ptr(bef.contiguous()), L.ptr(aft.contiguous()),
                                           L.ptr(stop.contiguous()), L.ptr(tgt.contiguous()), L.ptr(len32), B, T, L.ptr(w3),
                                           L.ptr(dbef), L.ptr(daft), L.ptr(dstop), L.stream()))
        return dbef, daft, dstop

    def l2_backward(self, gscale):
        self.begin_backward()
        L.check(self.lib.b2s_l2_backward(self.handle, L.ptr(gscale), L.stream()))


# ====================================================================================== autograd plumbing
def _param_list(module):
    return [(n, p) for n, p in module.named_parameters()]


class EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, prefix, names, inputs, lens32, spk, lang, train, *params):
        need = any(ctx.needs_input_grad)          # grad mode is off inside forward(); this reflects the caller's mode
        mem, c = eng.encoder_forward(inputs, lens32, spk, lang, train, eng.next_seed("encoder"), need)
        ctx.eng, ctx.c, ctx.names, ctx.prefix = eng, c, names, prefix
        ctx.req = [p.requires_grad for p in params]
        ctx.params = params                       # (the module's leaf parameters: grad_out looks at their .grad)
        return mem

    @staticmethod
    def backward(ctx, dmem):
        eng = ctx.eng
        eng.begin_backward()
        eng._claim_backward("encoder")
        eng.encoder_backward(ctx.c, dmem)
        ctx.c.free()
        grads = tuple(eng.grad_out(ctx.prefix + n, p) if r else None for n, r, p in zip(ctx.names, ctx.req, ctx.params))
        return (None,) * 8 + grads


class DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, prefix, names, memory, in32, targets, tgt32, train, holder, *params):
        mels, stop, c = eng.decoder_forward(memory, in32, targets, tgt32, train, eng.next_seed("decoder"), True)
        holder.append(c)
        ctx.eng, ctx.c, ctx.names, ctx.prefix = eng, c, names, prefix
        ctx.req = [p.requires_grad for p in params]
        ctx.params = params
        ctx.mem_shape = memory.shape
        ctx.mem_req = memory.requires_grad
        guided = eng.guided_loss(c).reshape(()) if eng.guided_enabled() else None
        return mels, stop, guided

    @staticmethod
    def backward(ctx, dmels, dstop, dguided):
        eng = ctx.eng
        eng.begin_backward()
        eng._claim_backward("decoder")
        if dmels is None:
            dmels = torch.zeros(ctx.mem_shape[0], ctx.c.keep[3].shape[1], ctx.c.keep[3].shape[2], device=ctx.c.keep[0].device)
        dmem = eng.decoder_backward(ctx.c, dmels, dstop, ctx.mem_shape, dguided.reshape(1) if dguided is not None else None,
                                    ctx.mem_req)
        grads = tuple(eng.grad_out(ctx.prefix + n, p) if r else None for n, r, p in zip(ctx.names, ctx.req, ctx.params))
        return (None, None, None, dmem if ctx.mem_req else None, None, None, None, None, None) + grads


_ENC_STREAMS = {}          # device index -> the process-wide encoder stream (HIP assigns a stream its hardware queue at creation: one per process, see trainer.py)


def encoder_stream(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _ENC_STREAMS:
        _ENC_STREAMS[key] = torch.cuda.Stream(device=device)
    return _ENC_STREAMS[key]


class EncDecFn(torch.autograd.Function):
    """Encoder + decoder of one Tacotron.forward as ONE autograd node (tacotron.py:126-129), so that the encoder can run BESIDE the decoder the way it
    does under HipTrainer -- inside the node, invisible to autograd: the forward runs the encoder on the process-wide encoder stream while this stream
    runs the decoder's prenet and first self-attention (which do not read the encoder output; b2s_decoder_forward waits for `memory_ready` where it first
    needs it); the backward starts the encoder backward there as soon as d(memory) is complete (`dmem_done`), beside the first decoder layer's
    self-attention backward, the prenet backward and the last weight-gradient groups.  Both streams are joined before either method returns: nothing a
    caller can see is ever pending on the other stream.  Same kernels, same seeds, same results as EncoderFn + DecoderFn (which stay for the modules
    used on their own); the reference's loop went from 9.0 to 8.x ms per step (profiles/NOTES_r06.md section 9)."""

    @staticmethod
    def forward(ctx, eng, n_enc, enc_names, dec_names, inputs, in32, spk, lang, targets, tgt32, train, holder, *params):
        need_enc = any(ctx.needs_input_grad[12:12 + n_enc])          # (grad mode is off inside forward(); this reflects the caller's mode)
        cur = torch.cuda.current_stream()
        enc_s = encoder_stream(inputs.device)
        L.check(eng.lib.b2s_model_set_side_stream(eng.handle, C.c_void_p(enc_s.cuda_stream)))
        eng.ensure_bound()                               # (on THIS stream, before anything forks off it: shadows / re-layouts)
        enc_s.wait_stream(cur)
        with torch.cuda.stream(enc_s):
            mem, c_enc = eng.encoder_forward(inputs, in32, spk, lang, train, eng.next_seed("encoder"), need_enc)
            mem_ready = torch.cuda.Event()
            mem_ready.record(enc_s)
        mels, stop, c_dec = eng.decoder_forward(mem, in32, targets, tgt32, train, eng.next_seed("decoder"), True, memory_ready=mem_ready)
        cur.wait_stream(enc_s)                           # join (the decoder already waited for mem_ready; this covers everything else of that stream)
        mem.record_stream(cur)
        holder.append(c_dec)
        holder.append(mem.shape)
        ctx.eng, ctx.c_enc, ctx.c_dec, ctx.n_enc = eng, c_enc, c_dec, n_enc
        ctx.enc_names, ctx.dec_names = enc_names, dec_names
        ctx.req = [p.requires_grad for p in params]
        ctx.params = params
        ctx.mem_shape = mem.shape
        ctx.need_enc = need_enc
        guided = eng.guided_loss(c_dec).reshape(()) if eng.guided_enabled() else None
        return mels, stop, guided

    @staticmethod
    def backward(ctx, dmels, dstop, dguided):
        eng = ctx.eng
        eng.begin_backward()
        eng._claim_backward("decoder")
        if dmels is None:
            dmels = torch.zeros(ctx.mem_shape[0], ctx.c_dec.keep[3].shape[1], ctx.c_dec.keep[3].shape[2], device=ctx.c_dec.keep[0].device)
        dg = dguided.reshape(1) if dguided is not None else None
        if ctx.need_enc:
            eng._claim_backward("encoder")
            cur = torch.cuda.current_stream()
            enc_s = encoder_stream(dmels.device)
            dmem_done = torch.cuda.Event()
            dmem_done.record(cur)                        # (torch creates the HIP event at its first record: the library re-records this handle)
            dmem = eng.decoder_backward(ctx.c_dec, dmels, dstop, ctx.mem_shape, dg, True, defer_join=True, dmem_done=dmem_done)
            enc_s.wait_event(dmem_done)                  # d(memory) only: the rest of the decoder backward runs beside the encoder's
            with torch.cuda.stream(enc_s):
                eng.encoder_backward(ctx.c_enc, dmem)    # (its last stage joins the engine's second stream)
            cur.wait_stream(enc_s)
        else:
            eng.decoder_backward(ctx.c_dec, dmels, dstop, ctx.mem_shape, dg, False)
        if ctx.c_enc is not None:
            ctx.c_enc.free()
        names = ["encoder." + n for n in ctx.enc_names] + ["decoder." + n for n in ctx.dec_names]
        grads = tuple(eng.grad_out(n, p) if r else None for n, r, p in zip(names, ctx.req, ctx.params))
        return (None,) * 12 + grads


class PostnetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, prefix, names, inputs, len32, fuse_add, train, *params):
        need = any(ctx.needs_input_grad)
        out, c = eng.postnet_forward(inputs, len32, inputs if fuse_add else None, train, eng.next_seed("postnet"), need)
        ctx.eng, ctx.c, ctx.names, ctx.prefix, ctx.fuse = eng, c, names, prefix, fuse_add
        ctx.req = [p.requires_grad for p in params]
        ctx.params = params
        return out

    @staticmethod
    def backward(ctx, dout):
        eng = ctx.eng
        eng.begin_backward()
        eng._claim_backward("postnet")
        din = eng.postnet_backward(ctx.c, dout)
        if ctx.fuse:
            din = eng.add(din, dout)
        ctx.c.free()
        grads = tuple(eng.grad_out(ctx.prefix + n, p) if r else None for n, r, p in zip(ctx.names, ctx.req, ctx.params))
        return (None, None, None, din, None, None, None) + grads


class LossFn(torch.autograd.Function):
    """vals[7] = loss, bef, aft, mse, l2, stop, sum_len.  The L2 gradient is accumulated straight into the
    engine's flat gradient buffer (returned to autograd by the segment functions)."""

    @staticmethod
    def forward(ctx, eng, bef, aft, stop, tgt, len32):
        vals, per = eng.loss_forward(bef, aft, stop, tgt, len32)
        ctx.eng = eng
        ctx.save_for_backward(bef, aft, stop, tgt, len32)
        ctx.mark_non_differentiable(per)
        return vals, per

    @staticmethod
    def backward(ctx, g, _gper):
        eng = ctx.eng
        bef, aft, stop, tgt, len32 = ctx.saved_tensors
        # d/d(bef_loss), d/d(aft_loss), d/d(stop_loss), d/d(l2) of the scalar being differentiated
        w = torch.stack([g[0] + g[1] + 0.5 * g[3], g[0] + g[2] + 0.5 * g[3], g[0] + g[5], g[0] + g[4]]).contiguous()
        eng.begin_backward()
        dbef, daft, dstop = eng.loss_backward(bef, aft, stop, tgt, len32, w)
        eng.l2_backward(w[3:])
        return None, dbef, daft, dstop, None, None
