"""Data-parallel gradient exchange: contiguous gradient ranges are all-reduced as soon as the backward stage
that completes them has been enqueued (train.py:125 wraps the model in DistributedDataParallel; here the
engine's stage hook drives RCCL directly -- one process per GPU, torch.distributed backend "nccl" = RCCL over
xGMI, "gloo" in the CPU tests).

Semantics of the reference are kept: every rank normalises its loss by its own sum of lengths, BatchNorm
statistics stay per rank, and gradients are averaged over ranks (sum here, 1/world folded into the optimizer).
"""
import torch


def broadcast_buffers(buffers, dist=None, src=0, group=None):
    """DistributedDataParallel(broadcast_buffers=True) semantics (train.py:125): before a forward pass every rank's module buffers --
    here the postnet's BatchNorm running_mean / running_var / num_batches_tracked -- are overwritten with rank `src`'s.  The floating
    buffers travel as ONE flat message (17 KB at the default sizes: 5 x 2 x <=512 floats) and are scattered back with one
    multi-tensor copy; integer counters (num_batches_tracked) advance identically on every rank and are sent as one int64 message."""
    dist = dist if dist is not None else torch.distributed
    for dtype_ok in (lambda t: t.is_floating_point(), lambda t: not t.is_floating_point()):
        ts = [t for t in buffers if dtype_ok(t)]
        if not ts:
            continue
        flat = torch.cat([t.reshape(-1).to(ts[0].dtype) for t in ts])
        dist.broadcast(flat, src, group=group)
        parts = flat.split([t.numel() for t in ts])
        with torch.no_grad():
            torch._foreach_copy_([t.view(-1) if t.dim() else t.view(1) for t in ts], list(parts))


class GradBucketer(object):
    """Merges consecutive stage ranges of the flat gradient buffer into buckets of >= bucket_elems elements and
    launches one asynchronous all-reduce per bucket.  Stages must be reported in increasing order and their ranges
    must tile the flat buffer in that order (HipEngine lays gradients out that way)."""

    def __init__(self, flat, stage_ranges, n_stages, bucket_elems, group=None, dist=None, payload="fp32", pack=None, unpack=None,
                 stream=None, consume_wire=False, mode="allreduce"):
        """payload "bf16": every bucket is converted to a bf16 wire buffer before its all-reduce and back afterwards -- half the
        bytes over xGMI (167 MB instead of 334 MB per step at the default sizes); the sum over ranks is then taken in bf16,
        everything inside a rank (accumulation, Adam moments, masters) stays fp32.  pack(src_f32, dst_bf16) / unpack(src_bf16,
        dst_f32): conversion ops (the trainer passes the HIP kernels b2s_pack_bf16 / b2s_unpack_bf16; default: torch copies, for
        CPU tests).  stream: torch.cuda.Stream the collectives (and the pack kernels) are launched on -- the engine orders THAT
        stream behind the stage's gradient work (b2s_model_set_stage_hook), the backward's own stream is not held up."""
        self.flat, self.stage_ranges, self.n_stages = flat, stage_ranges, n_stages
        self.bucket_elems = int(bucket_elems)
        self.dist = dist if dist is not None else torch.distributed
        self.group = group
        if payload not in ("fp32", "bf16"):
            raise ValueError("payload must be 'fp32' or 'bf16'")
        self.payload = payload
        self.wire = torch.empty(flat.numel(), dtype=torch.bfloat16, device=flat.device) if payload == "bf16" else None
        self.pack = pack or (lambda src, dst: dst.copy_(src))
        self.unpack = unpack or (lambda src, dst: dst.copy_(src))
        self.stream = stream
        # consume_wire (bf16 payload): the optimizer reads the all-reduced gradients from the wire buffer itself (b2s_adam_set_grad_wire);
        # finish() then skips the unpack pass and the fp32 buffer keeps this rank's local gradients
        self.consume_wire = bool(consume_wire) and self.wire is not None
        self._pending = None
        self._works = []
        self.launched = []                       # (lo, hi) of every all-reduce of the current step, for tests / logs
        # split (element index, or None): a bucket never straddles it -- the trainer updates the parameters below it (decoder + postnet)
        # as soon as THEIR all-reduces are complete (wait_prefix), beside the encoder backward
        self.split = None
        # mode "rs_ag" (sharded optimizer): every bucket is reduce-SCATTERED in place -- rank r keeps the r-th 1/world slice of each bucket
        # (owned_ranges), runs the optimizer on those slices only, and the updated parameters come back through all_gather_params
        if mode not in ("allreduce", "rs_ag"):
            raise ValueError("mode must be 'allreduce' or 'rs_ag'")
        self.mode = mode
        self.world = self.dist.get_world_size(group) if mode == "rs_ag" else 1
        self.rank = self.dist.get_rank(group) if mode == "rs_ag" else 0

    def plan(self):
        """The buckets a complete backward pass will launch, in order: [(lo, hi)] -- a pure function of the stage ranges, the bucket size
        and `split` (the same merging rule stage_done applies)."""
        out, pending = [], None
        for stage in range(self.n_stages):
            rng = self.stage_ranges.get(stage)
            if rng is not None and rng[1] > rng[0]:
                pending = rng if pending is None else (pending[0], rng[1])
            if pending is not None and (pending[1] - pending[0] >= self.bucket_elems or stage == self.n_stages - 1 or pending[1] == self.split):
                out.append(pending)
                pending = None
        return out

    def shard_of(self, lo, hi, rank=None):
        """Slice of bucket [lo, hi) that `rank` owns after the reduce-scatter (equal slices; bucket lengths are multiples of 8 x world elements:
        the flat layout's slots are multiples of 64)."""
        n = hi - lo
        if n % (8 * self.world):
            raise ValueError("bucket [%d, %d) is not a multiple of 8 x world (%d) elements" % (lo, hi, self.world))
        sh = n // self.world
        r = self.rank if rank is None else rank
        return lo + r * sh, lo + (r + 1) * sh

    def owned_ranges(self, rank=None):
        return [self.shard_of(lo, hi, rank) for lo, hi in self.plan()]

    def all_gather_params(self, wire):
        """wire: the flat fp32 parameter wire (b2s_param_wire) with this rank's owned slices filled in; in-place all-gather, bucket by bucket."""
        works = []
        for lo, hi in self.plan():
            a, b = self.shard_of(lo, hi)
            works.append(self.dist.all_gather_into_tensor(wire[lo:hi], wire[a:b], group=self.group, async_op=True))
        for w in works:
            w.wait()

    def begin_step(self):
        self._pending, self._works, self.launched = None, [], []

    def _launch(self):
        # Called on the thread that enqueued the backward stage.  Ordering: before it fired the hook the engine made the hook's
        # stream (self.stream, or the backward's stream when there is none) wait for the event that completes the stage's
        # gradients (engine.hip: end_stage / hook_after_*); the conversion kernel (bf16 payload) is enqueued on that stream;
        # torch.distributed makes its communication stream wait for the current stream before the collective starts.
        lo, hi = self._pending
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                work = self._launch_on_current(lo, hi)
        else:
            work = self._launch_on_current(lo, hi)
        self._works.append((work, lo, hi))
        self.launched.append((lo, hi))
        self._pending = None

    def _launch_on_current(self, lo, hi):
        buf = self.flat[lo:hi]
        if self.wire is not None:
            buf = self.wire[lo:hi]
            self.pack(self.flat[lo:hi], buf)
        if self.mode == "rs_ag":
            a, b = self.shard_of(lo, hi)
            return self.dist.reduce_scatter_tensor(buf[a - lo:b - lo], buf, group=self.group, async_op=True)      # in place: the own slice holds the sum
        return self.dist.all_reduce(buf, group=self.group, async_op=True)

    def stage_done(self, stage):
        rng = self.stage_ranges.get(stage)
        if rng is not None and rng[1] > rng[0]:
            if self._pending is None:
                self._pending = rng
            else:
                assert rng[0] == self._pending[1], "stage ranges must be contiguous in backward order"
                self._pending = (self._pending[0], rng[1])
        if self._pending is not None and (self._pending[1] - self._pending[0] >= self.bucket_elems or
                                          stage == self.n_stages - 1 or self._pending[1] == self.split):
            self._launch()

    def wait_prefix(self, limit):
        """Make the current stream wait for every all-reduce launched so far that lies below element `limit` (and, unless the optimizer
        consumes the wire buffer, unpack it).  Returns True if those ranges tile [0, limit) exactly -- i.e. the gradients below `limit`
        are final on the current stream; False (nothing waited for) if a stage below `limit` has not reported yet."""
        pos = 0
        for lo, hi in self.launched:
            if lo != pos or hi > limit:
                break
            pos = hi
        if pos != limit:
            return False
        rest = []
        for w, lo, hi in self._works:
            if hi <= limit:
                w.wait()
                if self.wire is not None and not self.consume_wire:
                    self.unpack(self.wire[lo:hi], self.flat[lo:hi])
            else:
                rest.append((w, lo, hi))
        self._works = rest
        return True

    def covers_all(self, expect_all=True):
        """Non-raising, non-waiting form of finish()'s coverage check: do the ranges launched so far plus the pending one tile the whole flat
        buffer in order?  (expect_all=False: always True -- a frozen encoder's stages never report.)"""
        if not expect_all:
            return True
        pos = 0
        for lo, hi in self.launched + ([self._pending] if self._pending is not None else []):
            if lo != pos:
                return False
            pos = hi
        return pos == self.flat.numel()

    def finish(self, expect_all=True):
        """Launch what is still pending, wait for every all-reduce and (expect_all) check that the launched ranges tile the
        whole flat gradient buffer exactly once -- a stage that never reported would otherwise leave its gradients un-averaged
        and the replicas would drift apart silently."""
        if self._pending is not None:
            self._launch()
        for w, lo, hi in self._works:
            w.wait()
            if self.wire is not None and not self.consume_wire:
                self.unpack(self.wire[lo:hi], self.flat[lo:hi])
        self._works = []
        if self.mode == "rs_ag" and expect_all and self.launched != self.plan():
            raise RuntimeError("reduce-scatter mode: the step launched buckets %s, the optimizer's shard was bound to %s" % (self.launched[:4], self.plan()[:4]))
        if expect_all:
            pos = 0
            for lo, hi in self.launched:
                if lo != pos:
                    break
                pos = hi
            if pos != self.flat.numel():
                raise RuntimeError("gradient exchange covered [0, %d) of %d elements (ranges %s): a backward stage did not report"
                                   % (pos, self.flat.numel(), self.launched[:4]))

    def unpack_all(self):
        """consume_wire: write the reduced gradients of the last step back into the fp32 buffer (tests / diagnostics)."""
        if self.wire is not None:
            for lo, hi in self.launched:
                self.unpack(self.wire[lo:hi], self.flat[lo:hi])

    def abort(self):
        """Wait for what was launched and drop the rest (a stage hook failed; the step is being abandoned)."""
        for w, _lo, _hi in self._works:
            try:
                w.wait()
            except Exception:
                pass
        self._works, self._pending = [], None
