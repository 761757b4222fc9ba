"""Multi-GPU sharding of the completion hot path: one process per GPU, shapes are independent.

Mirrors the reference's inference sharding (xgutils/plutil.py:123-139 `get_effective_visual_indices`: rank r takes
items r, r+G, r+2G, ...).  There is NO data-path collective: every rank holds a full weight replica (1.3 GB +
72 MB) and completes its own shapes; the only exchange is the optional final gather of the (B,L,2) token tensors /
timing scalars over torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" in CPU tests).
"""
from __future__ import annotations

import numpy as np
import torch


def effective_indices(indices, rank: int, world: int):
    """plutil.py:123-139: the items rank `rank` of `world` processes."""
    indices = np.asarray(indices)
    n = len(indices)
    cnt = -(-(n - rank) // world) if n > rank else 0
    return indices[rank + world * np.arange(cnt)]


def init_from_env(backend=None):
    """torch.distributed init from RANK/WORLD_SIZE/MASTER_* (torchrun); returns (rank, world, dist or None)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return 0, 1, None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"), rank=rank, world_size=world)
    return rank, world, dist


def gather_ragged_tokens(tokens: torch.Tensor, lengths: torch.Tensor, dist, world: int):
    """All-gather per-rank (B,Lpad,2) int32 token rows + (B,) lengths -> lists indexed by rank (rank-major order is the
    inverse of `effective_indices` striding: item i lives at [i % world][i // world])."""
    if dist is None or world == 1:
        return [tokens], [lengths]
    tl = [torch.empty_like(tokens) for _ in range(world)]
    ll = [torch.empty_like(lengths) for _ in range(world)]
    dist.all_gather(tl, tokens.contiguous())
    dist.all_gather(ll, lengths.contiguous())
    return tl, ll


def unshard(per_rank, n_items: int):
    """Inverse of the striding: per_rank[r][j] is item r + j*world."""
    world = len(per_rank)
    return [per_rank[i % world][i // world] for i in range(n_items)]


def allreduce_mean_(flat: torch.Tensor, dist):
    """DDP gradient synchronisation: ONE collective over the flat gradient buffer, mean over ranks (the reference gets
    this implicitly from Lightning's `accelerator="ddp"`, trainer.py:22,93).  RCCL on ROCm, gloo in CPU tests."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat)
        flat.div_(dist.get_world_size())
    return flat


class GradBuckets:
    """Bucketed gradient all-reduce overlapped with the backward pass (DDP training).

    The trainer's gradients live in ONE flat buffer laid out layer by layer; a bucket is a contiguous [lo, hi) slice of
    it.  `ready(name)` is called by the backward pass the moment the last gradient of that bucket has been enqueued on
    the compute stream and launches an asynchronous all-reduce of the slice (RCCL runs it on its own stream, ordered
    after the work already enqueued on the caller's stream), so the collective of layer l rides under the backward
    kernels of layers l-1, l-2, ...  `finish()` waits for every pending collective and applies the 1/world mean.
    Bucket = one transformer block (12.6 M floats = 50 MB): 26 collectives per step instead of 400 per-tensor ones,
    each large enough to run the ring at xGMI link rate.  gloo (CPU tests) takes the same path with SUM + scale.
    """

    def __init__(self, flat: torch.Tensor, ranges: dict, dist, single_rank_collectives=False):
        """single_rank_collectives: run the collectives even in a 1-rank process group (a no-op numerically; bench.py --force-dist
        uses it to exercise the RCCL ReduceOp.AVG bucket path on a 1-GPU box)."""
        self.flat, self.ranges, self.dist = flat, dict(ranges), dist
        inited = dist is not None and dist.is_initialized()
        self.world = dist.get_world_size() if inited else 1
        self.active = inited and (self.world > 1 or single_rank_collectives)
        self.pending, self.done = [], set()
        self.avg = False
        if self.active:
            self.avg = dist.get_backend() == "nccl"     # RCCL averages in the collective; gloo has no AVG
        self._wait_events = []                           # (before, after) HIP-event pairs around the waits of finish()

    def ready(self, name):
        if not self.active or name in self.done:
            return
        lo, hi = self.ranges[name]
        op = self.dist.ReduceOp.AVG if self.avg else self.dist.ReduceOp.SUM
        self.pending.append((name, self.dist.all_reduce(self.flat[lo:hi], op=op, async_op=True)))
        self.done.add(name)

    def finish(self):
        """Launch whatever was never marked ready, wait for everything, return the bucket names in launch order.  On a HIP
        device the compute stream's stall in these waits (collectives still running when the backward's last kernel is done
        = the EXPOSED communication of the step) is bracketed by two events; `wait_ms()` reads them."""
        if not self.active:
            return []
        for name in self.ranges:
            self.ready(name)
        ev = None
        if self.flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        order = []
        for name, work in self.pending:
            work.wait()
            order.append(name)
            if not self.avg:
                lo, hi = self.ranges[name]
                self.flat[lo:hi].div_(self.world)
        if ev is not None:
            ev[1].record()
            self._wait_events.append(ev)
        self.pending, self.done = [], set()
        return order

    def wait_ms(self, reset=True):
        """Mean time per finish() the compute stream spent waiting for gradient collectives since the last reset (ms); None
        when nothing was recorded (CPU tensors / inactive).  Synchronises."""
        if not self._wait_events:
            return None
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self._wait_events) / len(self._wait_events)
        if reset:
            self._wait_events = []
        return ms
