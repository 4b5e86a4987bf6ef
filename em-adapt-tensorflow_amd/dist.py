"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

The reference has no distributed code; its only batch-scaling mechanism is gradient accumulation,
``accum += g / accum_num`` (deeplab.py:202).  Data parallelism maps micro-batches to ranks: every rank
accumulates its own shard's gradients and the accumulators are summed across ranks (the 1/world factor
is folded into the momentum kernel), which reproduces the global-batch mean because
``reduce_mean`` (deeplab.py:182) runs over equal-sized local shards.

The accumulators live in ONE flat buffer laid out in backward order (fc8 first), so a bucket is just a
contiguous slice.  As backward finishes a layer it calls ``segment_ready``; every bucket that lies wholly
below the completed watermark is all-reduced asynchronously (torch.distributed: backend "nccl" = RCCL,
which runs the collective on its own HIP stream behind the compute stream's current position), so the
fc8/fc7/fc6 gradients -- 77 % of the bytes -- travel while conv5..conv1 are still being differentiated.
``wait()`` joins the outstanding collectives before the optimiser step.  Works unchanged on CPU tensors
with the gloo backend (tests).
"""
import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, flat, segments, bucket_bytes=64 << 20, group=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("torch.distributed is not initialised")
        self.flat = flat
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.total = max(e for _, e in segments)
        per = max(int(bucket_bytes) // flat.element_size(), 1)
        # bucket boundaries are multiples of `per`, clipped to the used extent of the flat buffer
        self.buckets = [(s, min(s + per, self.total)) for s in range(0, self.total, per)]
        self._next_bucket = 0
        self._watermark = 0
        self._works = []
        self.launched_bytes = 0

    def segment_ready(self, start, end):
        """Gradients of flat[start:end] are final for this update; segments arrive in increasing offset order."""
        if start != self._watermark:
            raise RuntimeError("segments must complete in layout order (got %d, expected %d)" % (start, self._watermark))
        self._watermark = end
        self._launch_ready()

    def _launch_ready(self):
        while self._next_bucket < len(self.buckets):
            s, e = self.buckets[self._next_bucket]
            if e > self._watermark:
                break
            view = self.flat[s:e]
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.launched_bytes += view.numel() * view.element_size()
            self._next_bucket += 1

    def wait(self):
        """Join all collectives of this update (the compute stream waits; the host does not block on CUDA)."""
        if self._watermark < self.total:
            raise RuntimeError("wait() before every segment was reported ready")
        self._launch_ready()
        for w in self._works:
            w.wait()
        self._works = []
        self._next_bucket = 0
        self._watermark = 0


def all_reduce_max(value, device):
    """MAX over ranks of a host float (bench timing)."""
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
