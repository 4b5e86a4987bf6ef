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

WHEN the collectives may start is the caller's decision.  The default is launch-on-completion: the exchange has nearly the
whole backward pass (~55 ms in fp32) to hide in.  ``hold()`` / ``release()`` let a step keep completed buckets back until a
chosen layer instead (config ``allreduce_release_at``).  Why one might: the kernels of fc6 .. conv3 are laid out as exactly
one workgroup per CU with no register or LDS room for anything beside them, so the workgroups of a collective that occupy a
few CUs when such a kernel starts delay the workgroups displaced from those CUs by the collective's duration (a few hundred
microseconds per 32 MB bucket), whereas conv2_2 .. conv1_1 (~6.5 ms, thousands of small workgroups) lose a few CUs
gracefully -- but then the whole 260 MB exchange must fit under those 6.5 ms or stay exposed.  Neither policy has been
measured on more than one GPU (no multi-GPU node was available to this build), so the conservative one is the default and
bench.py reports ``allreduce.exposed_ms_per_step`` for whichever is selected (``--release-at``).

Timing is OFF by default (a long training run must not pile up HIP events); bench.py and the tests switch it on, and the
record kept is bounded.

Two exchange algorithms (round 4), both leaving the SUM over ranks in the fp32 accumulator, identical on every rank:

* ``algo="allreduce"`` (default): one ``dist.all_reduce`` per fp32 bucket -- RCCL's ring / tree, whose per-step bandwidth is that
  of ONE xGMI link (~153 GB/s per direction of the seven a GPU has).
* ``algo="direct"``: reduce-scatter and all-gather written out as two all-to-all steps, so every rank talks to all its peers at
  once, one xGMI link each ("Distributed communication backend"): rank j owns shard j of a bucket; (1)
  ``all_to_all_single`` sends shard j of the local gradients to rank j; (2) rank j adds the W contributions IN FP32, IN RANK
  ORDER (deterministic; each element is summed exactly once, by its owner, so all ranks end up with the same bits); (3)
  ``all_gather_into_tensor`` returns the summed shards.  The work runs on a SIDE stream (comm + the pack / sum kernels), ordered
  behind the point of the compute stream at which the bucket became final; ``wait()`` makes the compute stream wait for it.
  Round 5: the arithmetic of (1)-(3) on GPU tensors is two kernels of the HIP library behind the C ABI -- ``emadapt_grad_pack`` (accumulator
  slice -> send buffer, one pass) and ``emadapt_grad_owner_sum`` (W shards added in fp32 in rank order, rounded once, one pass) -- instead
  of 6 + W torch launches and ~8 passes per bucket; a fp32 bucket of whole shards is sent straight from the accumulator (no pack at
  all); and the gathered payload is NOT copied back into the accumulator: it lands in a full-size buffer (``gathered``) that
  ``emadapt_sgd_momentum_payload`` reads directly (``writeback=False``, what ADAPT uses; the default True keeps ``flat`` = the sum for
  callers that want it there).  CPU tensors (the gloo tests) run the same arithmetic as torch ops -- the model the kernels are tested
  against bit for bit (tests/test_gpu_dp.py).

``payload="bf16"`` (``direct`` only; BASELINE configs[2] / [3], whose bf16 backward pass -- ~7 ms -- cannot hide 260 MB per
update): both all-to-all steps carry bf16, halving the bytes on every link: the exchanged gradient is
``bf16(sum_r fp32(bf16(g_r)))`` -- contributions rounded once (RNE), accumulated in fp32, the sum rounded once.  Master weights,
the accumulator and Momentum stay fp32 (deeplab.py:202-205 semantics on a gradient with ~3 significant digits: the same precision
class as the bf16 backward pass that produced it).
"""
import collections

import torch
import torch.distributed as dist


def _ops():
    from . import ops   # (the HIP library; imported on first use so that CPU-only gloo tests never load it)
    return ops


class GradSync:
    def __init__(self, flat, segments, bucket_bytes=64 << 20, group=None, algo="allreduce", payload="fp32", writeback=True, prepacked=False,
                 boundaries=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("torch.distributed is not initialised")
        if algo not in ("allreduce", "direct") or payload not in ("fp32", "bf16"):
            raise ValueError("algo must be 'allreduce' or 'direct', payload 'fp32' or 'bf16' (got %r, %r)" % (algo, payload))
        if payload == "bf16" and algo != "direct":
            raise ValueError("the bf16 gradient payload accumulates in fp32 on arrival: it needs algo='direct' (a ring all-reduce "
                             "of bf16 would round the running sum at every hop)")
        if not flat.dtype.is_floating_point or (payload == "bf16" and flat.dtype != torch.float32):
            raise TypeError("the accumulator must be a floating-point buffer (fp32 for the bf16 payload)")
        self.flat = flat
        self.group = group
        self.algo = algo
        self.payload = payload
        # direct exchange only: False = the summed gradients stay in `gathered` (payload type) and flat[...] keeps the LOCAL values --
        # the optimiser reads `gathered` (ops.sgd_momentum_payload); True = they are also copied back into flat
        self.writeback = bool(writeback) or algo != "direct"
        # direct exchange without write-back only: True = the CALLER fills `sendfull` (payload type, laid out like `gathered`) with this
        # rank's contribution before segment_ready() -- ops.sgd_accumulate_pack does, when accum_num == 1 -- and flat is never read: the
        # accumulate pass, the accumulator's traffic and the pack pass of the exchange disappear; the sums come out as with the pack
        self.prepacked = bool(prepacked) and algo == "direct" and not self.writeback
        self.sendfull = None
        self.gathered = None
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.total = max(e for _, e in segments)
        per = max(int(bucket_bytes) // flat.element_size(), 1)
        if algo == "direct":   # shards of equal size: bucket boundaries on multiples of the world size
            per = max(per // self.world_size, 1) * self.world_size
        # bucket boundaries are multiples of `per`, clipped to the used extent of the flat buffer -- or, with `boundaries`, the offsets
        # given (round 6: ADAPT cuts the direct exchange into TWO buckets, fc8 + fc7 + fc6 = 77 % of the bytes, final when fc6's filter
        # gradient ends, and conv5_3 .. conv1_1: two collectives per phase instead of eight, each large enough to run at the links'
        # bandwidth; an offset is rounded down to a whole number of shards, so a bucket is complete no later than the segment it names)
        if boundaries:
            W = self.world_size if algo == "direct" else 1
            cuts = sorted({(int(b) // W) * W for b in boundaries if 0 < (int(b) // W) * W < self.total})
            edges = [0] + cuts + [self.total]
            self.buckets = [(edges[i], edges[i + 1]) for i in range(len(edges) - 1)]
        else:
            self.buckets = [(s, min(s + per, self.total)) for s in range(0, self.total, per)]
        self._next_bucket = 0
        self._watermark = 0
        self._works = []
        self._held = False
        self.launched_bytes = 0
        # measurement (bench.py): time the compute stream spends blocked in wait(), per update, from HIP events on the
        # compute stream; with `trace` on, also where in the backward pass each bucket was launched
        self.timing = False    # bench.py / tests: set True (CUDA tensors only) to collect the events stats() reads
        self.trace = False
        # per update: (backward-start, [launch events], wait-start, [after-bucket events]); the last `keep` updates only
        self._ev = collections.deque(maxlen=256)
        self._cur = None
        # direct exchange: staging sized for the largest bucket (padded to whole shards), a side stream for CUDA tensors
        self._side = None
        self._side_used = False
        if algo == "direct":
            W = self.world_size
            pdt = torch.bfloat16 if payload == "bf16" else flat.dtype   # ("fp32" = the accumulator's own type)
            n = max(-(-(e - s) // W) * W for s, e in self.buckets)
            self._send = torch.zeros(n, dtype=pdt, device=flat.device)
            self._recv = torch.empty(n, dtype=pdt, device=flat.device)
            self._shard = torch.empty(n // W, dtype=flat.dtype, device=flat.device)
            self._shard_p = torch.empty(n // W, dtype=pdt, device=flat.device) if payload == "bf16" else self._shard
            # the gathered payload of EVERY bucket, each at its own offset (all buckets but the last are whole shards, so bucket
            # [s, e) lives at [s, s + npad)): what the optimiser reads when nothing is written back
            s_last, e_last = self.buckets[-1]
            # the optimiser reads gathered[i] as the summed gradient of flat[i] (ops.sgd_momentum_payload over the whole flat buffer, or
            # slices of both at the same offsets): that holds because every bucket but the last is a whole number of shards, so its
            # gathered image is exactly as long as the bucket itself
            assert all((e - s) % W == 0 for s, e in self.buckets[:-1]), "buckets before the last must be whole shards"
            self.gathered = torch.zeros(s_last + -(-(e_last - s_last) // W) * W, dtype=pdt, device=flat.device)
            if self.prepacked:
                self.sendfull = torch.zeros_like(self.gathered)   # (the padding behind the last bucket stays zero: nobody writes it)
            if flat.is_cuda:
                self._side = torch.cuda.Stream(device=flat.device)
            assert self.gathered.numel() >= self.total and (self.sendfull is None or self.sendfull.numel() == self.gathered.numel())
        self.wire_bytes_per_update = self._wire_bytes()

    def _wire_bytes(self):
        """Bytes one rank SENDS per update (what its xGMI links carry): ring all-reduce 2 (W-1)/W x payload; direct: two
        all-to-all steps of (W-1)/W x payload each."""
        W = self.world_size
        el = 2 if self.payload == "bf16" else self.flat.element_size()
        return int(2 * (W - 1) / W * self.total * el)

    def _exchange_direct(self, s, e):
        """flat[s:e] summed over ranks by shard owners (see the module docstring) into gathered[s:s+npad] (and back into flat[s:e]
        with `writeback`).  Runs on the current stream."""
        W = self.world_size
        n = e - s
        npad = -(-n // W) * W
        sh = npad // W
        recv, gath = self._recv[:npad], self.gathered[s:s + npad]
        hip = self.flat.is_cuda
        if self.prepacked:
            send = self.sendfull[s:s + npad]         # filled by the caller, already in the payload type
        elif self.payload != "bf16" and npad == n:
            send = self.flat[s:e]                    # whole fp32 shards: sent as they lie in the accumulator
        else:
            send = self._send[:npad]
            if hip:
                _ops().grad_pack(self.flat[s:e], send)   # fp32 -> payload type (bf16: round to nearest even), zero padding: one pass
            else:
                send[:n].copy_(self.flat[s:e])
                if npad > n:
                    send[n:].zero_()
        dist.all_to_all_single(recv, send, group=self.group)
        out = self._shard_p[:sh]
        if hip:
            _ops().grad_owner_sum(recv, W, out)      # ranks 0, 1, 2, ... added in fp32, rounded once: one pass
        else:
            acc = self._shard[:sh]
            acc.copy_(recv[:sh])                     # contribution of rank 0, then 1, 2, ... : fixed order, fp32 adds
            for r in range(1, W):
                acc.add_(recv[r * sh:(r + 1) * sh])
            if self.payload == "bf16":
                out.copy_(acc)                       # the sum, rounded once
        dist.all_gather_into_tensor(gath, out, group=self.group)
        if self.writeback:
            self.flat[s:e].copy_(gath[:n])

    def _event(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def segment_ready(self, start, end):
        """Gradients of flat[start:end] are final for this update; segments arrive in increasing offset order."""
        if self.timing and self._cur is None:
            self._cur = dict(t0=self._event(), launch=[], wait0=None, done=[])
        if start != self._watermark:
            raise RuntimeError("segments must complete in layout order (got %d, expected %d)" % (start, self._watermark))
        self._watermark = end
        self._launch_ready()

    def hold(self):
        """Completed buckets are queued instead of launched, until release() (or wait())."""
        self._held = True

    def release(self):
        """Launch every bucket that is complete by now, and from here on each one as soon as it completes."""
        self._held = False
        self._launch_ready()

    def _launch_ready(self):
        if self._held:
            return
        while self._next_bucket < len(self.buckets):
            s, e = self.buckets[self._next_bucket]
            if e > self._watermark:
                break
            view = self.flat[s:e]
            if self.timing and self.trace:
                self._cur["launch"].append(self._event())
            if self.algo == "direct":
                if self._side is not None:   # behind the point at which this bucket became final, beside the rest of backward
                    self._side.wait_stream(torch.cuda.current_stream(self.flat.device))
                    with torch.cuda.stream(self._side):
                        self._exchange_direct(s, e)
                    self._side_used = True
                else:
                    self._exchange_direct(s, e)
            else:
                self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.launched_bytes += view.numel() * view.element_size()
            self._next_bucket += 1

    def wait(self):
        """Join all collectives of this update (the compute stream waits; the host does not block on CUDA)."""
        if self._watermark < self.total:
            raise RuntimeError("wait() before every segment was reported ready")
        self._held = False
        self._launch_ready()
        if self.timing and self._cur is not None:
            self._cur["wait0"] = self._event()
        for w in self._works:
            w.wait()   # the compute stream waits for the collective; the host does not block
            if self.timing and self.trace and self._cur is not None:
                self._cur["done"].append(self._event())
        if self._side_used:
            torch.cuda.current_stream(self.flat.device).wait_stream(self._side)
            self._side_used = False
        if self.timing and self._cur is not None:
            self._cur["wait1"] = self._event()
            self._ev.append(self._cur)
            self._cur = None
        self._works = []
        self._next_bucket = 0
        self._watermark = 0

    def stats(self, reset=True):
        """Since the last reset: updates seen, mean milliseconds per update the compute stream was blocked on the
        gradient exchange (the EXPOSED part of the all-reduce; 0 = fully hidden under backward), payload bytes per update,
        bucket count, and -- with `trace` on -- for the last update the offset of every bucket's launch from the start of
        the gradient exchange window and the stall observed when it was joined (ms)."""
        out = dict(updates=len(self._ev), world_size=self.world_size, buckets=len(self.buckets),
                   bytes_per_update=self.total * self.flat.element_size(), algo=self.algo, payload=self.payload,
                   wire_bytes_per_update=self.wire_bytes_per_update, exposed_ms=None, backward_window_ms=None)
        if self._ev:
            torch.cuda.synchronize(self.flat.device)
            out["exposed_ms"] = sum(e["wait0"].elapsed_time(e["wait1"]) for e in self._ev) / len(self._ev)
            out["backward_window_ms"] = sum(e["t0"].elapsed_time(e["wait0"]) for e in self._ev) / len(self._ev)
            last = self._ev[-1]
            if last["launch"]:
                marks = [last["wait0"]] + last["done"]
                out["trace"] = [dict(bucket=i, launched_at_ms=round(last["t0"].elapsed_time(l), 3),
                                     stall_at_join_ms=round(marks[i].elapsed_time(marks[i + 1]), 3) if i + 1 < len(marks) else None)
                                for i, l in enumerate(last["launch"])]
        if reset:
            self._ev.clear()
        return out


def all_reduce_max(value, device):
    """MAX over ranks of a host float (bench timing)."""
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
