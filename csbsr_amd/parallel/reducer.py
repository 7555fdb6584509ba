"""Data-parallel gradient exchange: one process per GPU, replicas hold their own weights, one all-reduce(sum)
per gradient bucket per step over torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in the CPU tests), averaged by 1/world.

Replaces the reference's single-process nn.DataParallel (train.py:108-112: per-step parameter broadcast +
ReduceAddCoalesced of 357 MB of gradients to GPU 0).  Buckets follow the order in which the
explicit backward finishes parameter groups -- the segmentation network's bucket is final before the SR
network's backward starts, and each KBPN stage's bucket (reverse stage order) is final when the LAST micro-batch's
backward has passed that stage -- so every all-reduce but the last runs on a side stream underneath the remaining
backward.  Buckets are preallocated flat fp32 buffers that the per-parameter gradient accumulators are views of
(``launch_flat``): no flatten / scatter copies.
"""
import os

import torch
import torch.distributed as dist


def _forced():
    """CSBSR_FORCE_DIST=1: issue every collective even in a one-rank group, so the real backend (RCCL on the GPU box) executes the
    N > 1 call sequence -- side-stream launches from inside the backward, in-place flat buckets -- on the one GPU a test box has."""
    return os.environ.get("CSBSR_FORCE_DIST") == "1"


class GradBucketReducer:
    def __init__(self, process_group=None, side_stream=None, force=None):
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.side_stream = side_stream
        self.force = (_forced() if force is None else bool(force)) and dist.is_initialized()
        self._pending = []
        # what was exchanged so far: collectives issued, of which on the side stream, and payload bytes (tests / bench JSON)
        self.stats = {"all_reduces": 0, "on_side_stream": 0, "bytes": 0, "steps": 0, "agreements": 0}
        self._exposed = []          # per step: (event at the main stream's arrival in finish(), event after its last wait)

    def agree_min(self, values):
        """Element-wise minimum of a short list of ints over the ranks (one tiny collective): every rank must take the same
        memory-dependent scheduling decision -- a rank that alone recomputes a forward makes all the others wait at the all-reduce."""
        if not self.active:
            return [int(v) for v in values]
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.pg)
        self.stats["agreements"] += 1
        return [int(v) for v in t.tolist()]

    def exposed_ms(self):
        """per finished step: how long the compute stream stood still in finish() until the last bucket had arrived, i.e. the part of
        the gradient exchange that was NOT hidden under the backward (call after a device synchronize)."""
        return [round(a.elapsed_time(b), 3) for a, b in self._exposed]

    @property
    def active(self):
        return self.world > 1 or self.force

    def _all_reduce(self, flat):
        self.stats["all_reduces"] += 1
        self.stats["bytes"] += flat.numel() * flat.element_size()
        if self.side_stream is not None and flat.is_cuda:
            self.stats["on_side_stream"] += 1
            self.side_stream.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(self.side_stream):
                return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def launch(self, grads):
        """Start the all-reduce of a list of fp32 gradient tensors (None entries are skipped).  Returns a handle."""
        live = [g for g in grads if g is not None]
        if not live or not self.active:
            h = (grads, None, None, None)
            self._pending.append(h)
            return h
        flat = torch.cat([g.reshape(-1) for g in live])
        work = self._all_reduce(flat)
        h = (grads, live, flat, work)
        self._pending.append(h)
        return h

    def launch_flat(self, flat):
        """Start the in-place all-reduce of one preallocated flat fp32 bucket (the gradient accumulators of a parameter group are
        views into it: no flatten copy before, no scatter copy after).  The caller must not touch the bucket until finish()."""
        if not self.active or flat is None or flat.numel() == 0:
            return None
        work = self._all_reduce(flat)
        h = (None, None, flat, work)
        self._pending.append(h)
        return h

    def finish(self):
        """Wait for every launched bucket and write the averaged values back in place."""
        timed = [h for h in self._pending if h[3] is not None and h[2].is_cuda]
        if timed:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        for grads, live, flat, work in self._pending:       # first every wait (that is what can be exposed), then the arithmetic
            if work is None:
                continue
            work.wait()
            if self.side_stream is not None and flat.is_cuda:
                torch.cuda.current_stream(flat.device).wait_stream(self.side_stream)
        if timed:
            ev1.record()
            self._exposed = (self._exposed + [(ev0, ev1)])[-64:]
        for grads, live, flat, work in self._pending:
            if work is None:
                continue
            if self.world > 1:          # (a one-rank group's sum is the value itself: forced runs stay bit-identical to undistributed ones)
                flat.mul_(1.0 / self.world)
            if live is None:            # launch_flat: the accumulators are views of the bucket
                continue
            off = 0
            for g in live:
                n = g.numel()
                g.copy_(flat[off:off + n].view_as(g))
                off += n
        self._pending = []
        self.stats["steps"] += 1


def broadcast_parameters(module, src=0, process_group=None, force=None):
    """Make every replica start from rank ``src``'s weights and buffers (once, not per step): the tensors are packed into one flat
    buffer per (parameter | buffer, dtype) group -- fp32 weights, fp32 BatchNorm running statistics, int64 batch counters: three
    broadcasts where one per tensor would be 410 (290 parameters + 120 buffers with PSPNet) -- and copied back in place.  Returns the
    number of broadcasts issued."""
    if not dist.is_initialized():
        return 0
    if dist.get_world_size(process_group) == 1 and not (_forced() if force is None else force):
        return 0
    groups = {}
    for kind, ts in (("param", module.parameters()), ("buffer", module.buffers())):
        for t in ts:
            groups.setdefault((kind, t.dtype, t.device), []).append(t.data)
    n = 0
    with torch.no_grad():
        for (_, dtype, device), ts in groups.items():
            flat = torch.cat([t.reshape(-1) for t in ts])
            dist.broadcast(flat, src=src, group=process_group)
            n += 1
            off = 0
            for t in ts:
                k = t.numel()
                t.copy_(flat[off:off + k].view_as(t))
                off += k
    return n
