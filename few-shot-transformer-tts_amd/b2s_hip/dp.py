"""Data-parallel gradient exchange: contiguous gradient ranges are all-reduced as soon as the backward stage
that completes them has been enqueued (train.py:125 wraps the model in DistributedDataParallel; here the
engine's stage hook drives RCCL directly -- one process per GPU, torch.distributed backend "nccl" = RCCL over
xGMI, "gloo" in the CPU tests).

Semantics of the reference are kept: every rank normalises its loss by its own sum of lengths, BatchNorm
statistics stay per rank, and gradients are averaged over ranks (sum here, 1/world folded into the optimizer).
"""
import torch


class GradBucketer(object):
    """Merges consecutive stage ranges of the flat gradient buffer into buckets of >= bucket_elems elements and
    launches one asynchronous all-reduce per bucket.  Stages must be reported in increasing order and their ranges
    must tile the flat buffer in that order (HipEngine lays gradients out that way)."""

    def __init__(self, flat, stage_ranges, n_stages, bucket_elems, group=None, dist=None):
        self.flat, self.stage_ranges, self.n_stages = flat, stage_ranges, n_stages
        self.bucket_elems = int(bucket_elems)
        self.dist = dist if dist is not None else torch.distributed
        self.group = group
        self._pending = None
        self._works = []
        self.launched = []                       # (lo, hi) of every all-reduce of the current step, for tests / logs

    def begin_step(self):
        self._pending, self._works, self.launched = None, [], []

    def _launch(self):
        lo, hi = self._pending
        self._works.append(self.dist.all_reduce(self.flat[lo:hi], group=self.group, async_op=True))
        self.launched.append((lo, hi))
        self._pending = None

    def stage_done(self, stage):
        rng = self.stage_ranges.get(stage)
        if rng is not None and rng[1] > rng[0]:
            if self._pending is None:
                self._pending = rng
            else:
                assert rng[0] == self._pending[1], "stage ranges must be contiguous in backward order"
                self._pending = (self._pending[0], rng[1])
        if self._pending is not None and (self._pending[1] - self._pending[0] >= self.bucket_elems or
                                          stage == self.n_stages - 1):
            self._launch()

    def finish(self, expect_all=True):
        """Launch what is still pending, wait for every all-reduce and (expect_all) check that the launched ranges tile the
        whole flat gradient buffer exactly once -- a stage that never reported would otherwise leave its gradients un-averaged
        and the replicas would drift apart silently."""
        if self._pending is not None:
            self._launch()
        for w in self._works:
            w.wait()
        self._works = []
        if expect_all:
            pos = 0
            for lo, hi in self.launched:
                if lo != pos:
                    break
                pos = hi
            if pos != self.flat.numel():
                raise RuntimeError("gradient exchange covered [0, %d) of %d elements (ranges %s): a backward stage did not report"
                                   % (pos, self.flat.numel(), self.launched[:4]))

    def abort(self):
        """Wait for what was launched and drop the rest (a stage hook failed; the step is being abandoned)."""
        for w in self._works:
            try:
                w.wait()
            except Exception:
                pass
        self._works, self._pending = [], None
