"""Multi-GPU sharding of the completion hot path: one process per GPU, shapes are independent.

Mirrors the reference's inference sharding (xgutils/plutil.py:123-139 `get_effective_visual_indices`: rank r takes
items r, r+G, r+2G, ...).  There is NO data-path collective: every rank holds a full weight replica (1.3 GB +
72 MB) and completes its own shapes; the only exchange is the optional final gather of the (B,L,2) token tensors /
timing scalars over torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" in CPU tests).
The single-shape option of SURVEY 8(e) - the sample_n sequences of ONE shape split over the ranks - is `sample_n_sharded`
below: its one exchange is the early-stop vote (one int per `check_every` steps) and the final gather of the tokens.
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


class _StagedWork:
    """reduce_scatter_tensor / all_gather_into_tensor of a DEVICE buffer over gloo, staged through host memory (test path)."""

    def __init__(self, dist, kind, flat, whole, part, op):
        (lo, hi), (slo, shi) = whole, part
        if kind == "rs":
            src = flat[lo:hi].cpu()
            out = torch.empty(shi - slo, dtype=flat.dtype)
            dist.reduce_scatter_tensor(out, src, op=op)
            flat[slo:shi].copy_(out)
        else:
            out = torch.empty(hi - lo, dtype=flat.dtype)
            dist.all_gather_into_tensor(out, flat[slo:shi].cpu())
            flat[lo:hi].copy_(out)

    def wait(self):
        return True


class GradBuckets:
    """Bucketed gradient synchronisation overlapped with the backward pass (DDP training).

    The trainer's gradients live in ONE flat buffer laid out layer by layer; a bucket is a contiguous [lo, hi) slice of
    it.  `ready(name)` is called by the backward pass the moment the last gradient of that bucket has been enqueued on
    the compute stream and launches an asynchronous collective on the slice (RCCL runs it on its own stream, ordered
    after the work already enqueued on the caller's stream), so the collective of layer l rides under the backward
    kernels of layers l-1, l-2, ...  `finish()` waits for every pending collective and applies the 1/world mean.
    Bucket = one transformer block (12.6 M floats = 50 MB): 26 collectives per step instead of 400 per-tensor ones,
    each large enough to run at xGMI link rate.  gloo (CPU tests) takes the same path with SUM + scale.

    mode "ring"  : `all_reduce` per bucket - every rank ends up with the whole mean gradient and updates every parameter.
    mode "rs_ag" : north_star's path - `reduce_scatter_tensor` per bucket IN PLACE (rank r's output is its own 1/N slice of
                   the bucket), the optimizer updates only that slice (`shard_ranges()`), and `all_gather_params()` sends the
                   updated parameters back through the same flat buffer (`all_gather_into_tensor`, in place).  Half the bytes
                   of a ring all-reduce cross the links before the optimizer can start, the other half after it; AdamW work
                   and moment traffic drop to 1/N per rank.  A bucket whose length the world size does not divide falls back
                   to "ring" (every rank then updates it in full).
    """

    def __init__(self, flat: torch.Tensor, ranges: dict, dist, single_rank_collectives=False, mode="ring", profile_waits=False):
        """single_rank_collectives: run the collectives even in a 1-rank process group (a no-op numerically; bench.py --force-dist
        uses it to exercise the RCCL bucket path on a 1-GPU box).  profile_waits: bracket the waits of finish() /
        all_gather_params() with HIP events for `wait_ms()` (bench.py only: a training run would grow the list without bound)."""
        assert mode in ("ring", "rs_ag")
        self.flat, self.ranges, self.dist, self.mode = flat, dict(ranges), dist, mode
        inited = dist is not None and dist.is_initialized()
        self.world = dist.get_world_size() if inited else 1
        self.rank = dist.get_rank() if inited else 0
        self.active = inited and (self.world > 1 or single_rank_collectives)
        self.pending, self.done = [], set()
        self.avg = False
        if self.active:
            self.avg = dist.get_backend() == "nccl"     # RCCL averages in the collective; gloo has no AVG
        self.profile_waits = bool(profile_waits)
        self._wait_events = []                           # (before, after) HIP-event pairs around the waits (profile_waits only)
        self._gather_events = []                         # the same around the waits for parameter all-gathers (mode rs_ag)
        self._gather_steps = 0                           # optimizer steps those events belong to
        self._param_works = {}                           # bucket name -> pending all-gather of its updated parameters

    def _staged(self):
        """gloo has no device path for reduce_scatter_tensor / all_gather_into_tensor: device buffers go through the host
        (the 2-process tests on a 1-GPU box; RCCL runs them in place on the device)."""
        return self.flat.is_cuda and self.dist.get_backend() == "gloo"

    def sharded(self, name):
        """True when bucket `name` is reduce-scattered (mode rs_ag and the world size divides its length)."""
        lo, hi = self.ranges[name]
        return self.active and self.mode == "rs_ag" and (hi - lo) % self.world == 0

    def shard(self, name):
        """[lo, hi) of the part of bucket `name` this rank owns (the whole bucket unless it is reduce-scattered)."""
        lo, hi = self.ranges[name]
        if not self.sharded(name):
            return lo, hi
        n = (hi - lo) // self.world
        return lo + self.rank * n, lo + (self.rank + 1) * n

    def shard_ranges(self):
        """The flat ranges this rank must update, in bucket order."""
        return [self.shard(name) for name in self.ranges]

    def ready(self, name):
        if not self.active or name in self.done:
            return
        lo, hi = self.ranges[name]
        op = self.dist.ReduceOp.AVG if self.avg else self.dist.ReduceOp.SUM
        if self.sharded(name):
            slo, shi = self.shard(name)
            if self._staged():
                work = _StagedWork(self.dist, "rs", self.flat, (lo, hi), (slo, shi), op)
            else:
                work = self.dist.reduce_scatter_tensor(self.flat[slo:shi], self.flat[lo:hi], op=op, async_op=True)
        else:
            work = self.dist.all_reduce(self.flat[lo:hi], op=op, async_op=True)
        self.pending.append((name, work))
        self.done.add(name)

    def wait_bucket(self, name):
        """Wait (stream-ordered on the CURRENT stream for RCCL) for the gradient collective of ONE bucket and apply the 1/world mean where
        the backend has no AVG - the per-bucket form of finish(), for an optimizer that updates a bucket as soon as its gradients are
        final (GPTTrainer's fused step).  True when a collective was pending for it."""
        if not self.active:
            return False
        for i, (n, work) in enumerate(self.pending):
            if n == name:
                ev = self._bracket()
                work.wait()
                if not self.avg:
                    lo, hi = self.shard(name)
                    self.flat[lo:hi].div_(self.world)
                if ev is not None:
                    ev[1].record()
                    self._wait_events.append(ev)
                del self.pending[i]
                return True
        return False

    def end_step(self):
        """Close the books of a step whose buckets were all consumed through wait_bucket()."""
        assert not self.pending, f"buckets still pending: {[n for n, _ in self.pending]}"
        self.done = set()
        if self.active and self.mode == "rs_ag":
            self._gather_steps += 1

    def launch_param_gather(self, name):
        """mode rs_ag: launch the all-gather of ONE bucket's updated parameters (the fused step: right after that bucket's sharded AdamW,
        while the backward pass of the blocks below is still running); consumed by wait_params(name)."""
        if self.active and self.mode == "rs_ag" and self.sharded(name):
            assert name not in self._param_works, f"the gather of bucket {name} was never consumed"
            self._param_works[name] = self._all_gather_bucket(self.flat, name)

    def _bracket(self):
        if self.profile_waits and self.flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
            return ev
        return None

    def finish(self):
        """Launch whatever was never marked ready, wait for everything, return the bucket names in launch order.  With
        profile_waits the compute stream's stall in these waits (collectives still running when the backward's last kernel is
        done = the EXPOSED communication of the step) is bracketed by two events; `wait_ms()` reads them."""
        if not self.active:
            return []
        for name in self.ranges:
            self.ready(name)
        ev = self._bracket()
        order = []
        for name, work in self.pending:
            work.wait()
            order.append(name)
            if not self.avg:
                lo, hi = self.shard(name)
                self.flat[lo:hi].div_(self.world)
        if ev is not None:
            ev[1].record()
            self._wait_events.append(ev)
        self.pending, self.done = [], set()
        return order

    def _all_gather_bucket(self, flat, name):
        lo, hi = self.ranges[name]
        slo, shi = self.shard(name)
        if self._staged():
            return _StagedWork(self.dist, "ag", flat, (lo, hi), (slo, shi), None)
        return self.dist.all_gather_into_tensor(flat[lo:hi], flat[slo:shi], async_op=True)

    def all_gather_flat(self, flat, profile=False):
        """In-place all-gather of every reduce-scattered bucket of `flat` - any buffer laid out like the gradient buffer (e.g. the
        AdamW moments when a checkpoint is written): rank r's slice of each bucket goes to every rank.  All collectives are
        launched, then waited for (blocking form; the parameters of a training step use launch_param_gathers / wait_params)."""
        if not self.active or self.mode != "rs_ag":
            return
        works = [self._all_gather_bucket(flat, name) for name in self.ranges if self.sharded(name)]
        ev = self._bracket() if profile else None
        for w in works:
            w.wait()
        if ev is not None:
            ev[1].record()
            self._gather_events.append(ev)
            self._gather_steps += 1

    def all_gather_params(self):
        """mode rs_ag, blocking form: all buckets gathered before the call returns to the compute stream."""
        self.all_gather_flat(self.flat, profile=True)

    def launch_param_gathers(self, order=None):
        """mode rs_ag, after the sharded optimizer step wrote the updated parameters of this rank's slices into the flat buffer:
        launch one all-gather per bucket in the order the NEXT forward reads the parameters (`order`: bucket names; default = the
        bucket order) and return without waiting.  RCCL runs them back to back on its own stream; `wait_params(name)` makes the
        compute stream wait for ONE bucket right before the first kernel that reads it, so the gather of bucket k rides under the
        forward kernels of the buckets before it.  -> the names of the buckets whose parameters are now in flight."""
        if not self.active or self.mode != "rs_ag":
            return []
        assert not self._param_works, "launch_param_gathers: the previous step's gathers were never consumed (wait_params / drain_params)"
        for name in (order or list(self.ranges)):
            if self.sharded(name):
                self._param_works[name] = self._all_gather_bucket(self.flat, name)
        self._gather_steps += 1
        return list(self._param_works)

    def wait_params(self, name):
        """Wait (stream-ordered for RCCL) for the parameter all-gather of bucket `name`; True when there was one pending - the
        caller then copies the bucket's gathered values out of the flat buffer.  The stall is what `gather_ms()` reports."""
        work = self._param_works.pop(name, None)
        if work is None:
            return False
        ev = self._bracket()
        work.wait()
        if ev is not None:
            ev[1].record()
            self._gather_events.append(ev)
        return True

    def params_in_flight(self):
        return list(self._param_works)

    def wait_ms(self, reset=True):
        """Mean time per finish() the compute stream spent waiting for gradient collectives since the last reset (ms); None
        when nothing was recorded (CPU tensors / inactive / profile_waits off).  Synchronises."""
        if not self._wait_events:
            return None
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self._wait_events) / len(self._wait_events)
        if reset:
            self._wait_events = []
        return ms

    def gather_ms(self, reset=True):
        """The same for the parameter all-gathers of mode rs_ag: mean EXPOSED wait per optimizer step (the sum of the per-bucket
        stalls of a step when the gathers are overlapped with the next forward)."""
        evs = self._gather_events
        if not evs:
            return None
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in evs) / max(self._gather_steps, 1)
        if reset:
            self._gather_events, self._gather_steps = [], 0
        return ms


def sample_n_sharded(gpt, c_tokens, Lc, sample_n, dist=None, **sample_kw):
    """SURVEY section 8(e), the single-shape option: the `sample_n` sequences of ONE condition (VisShapeFormer.compute_batch,
    shapeformer.py:222-260) split over the ranks.  c_tokens (1,Lpad,2) / Lc (1,) hold the condition once; rank r samples rows
    [S r / W, S (r + 1) / W) of the S copies with their GLOBAL row index (uniform stream, greedy row 0), the early stop is decided
    by an all-reduce of "all my rows have ended" at every check (the one exchange step this path has: one int per `check_every`
    steps), and the (rows, steps, 2) tokens + log-probabilities are gathered on every rank in global row order.  The result is the
    one `gpt.sample` of all S rows returns in a single process, bit for bit (tests/test_ddp_gpu.py)."""
    S = int(sample_n)
    world = 1 if dist is None or not dist.is_initialized() else dist.get_world_size()
    rank = 0 if world == 1 else dist.get_rank()
    if world > 1 and S < world and sample_kw.get("stop_early", True):      # decided identically on every rank: nobody enters a collective
        raise ValueError("sample_n_sharded: fewer sequences than ranks with the early stop on (a rank without rows cannot follow the stop votes)")
    lo, hi = S * rank // world, S * (rank + 1) // world
    biggest = -(-S // world)        # the largest shard of any rank: the limit is tested on a rank-independent number, so EVERY rank raises
    if biggest > 4 * gpt.MAX_CHAIN_ROWS:
        raise ValueError(f"sample_n_sharded: {biggest} rows on one rank (at most {4 * gpt.MAX_CHAIN_ROWS}: successive rounds stop independently)")
    if world > 1:
        # every rank must hold the SAME condition (the reference's inference sharding hands the ranks different items: enabling this
        # option under that split would gather rows of different shapes into one result): compare a checksum before anyone samples
        import zlib
        ct = torch.as_tensor(c_tokens)[:1].to("cpu", torch.int64).contiguous()
        n = int(torch.as_tensor(Lc).reshape(-1)[0])
        h = zlib.crc32(ct[0, :n].numpy().tobytes(), n) & 0x7fffffff
        t = torch.tensor([h, -h], dtype=torch.int64, device=gpt.dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)          # max(h) == -max(-h) = min(h) iff all ranks agree
        if int(t[0]) != -int(t[1]):
            raise ValueError("sample_n_sharded: the ranks hold different conditions (c_tokens / Lc); the sample_n split needs ONE shape on every rank")
    sample_kw = dict(sample_kw)
    sample_kw.setdefault("shared_prefix", "auto")
    res = None
    if hi > lo:
        rows = c_tokens[:1].expand(hi - lo, -1, -1).contiguous()
        lens = torch.as_tensor(Lc).reshape(-1)[:1].expand(hi - lo).contiguous()

        def ended_reduce(e):
            if world == 1:
                return e
            t = torch.tensor([1 if e else 0], dtype=torch.int32, device=gpt.dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())
        res = gpt.sample(rows, lens, row_offset=lo, rows_total=S, ended_reduce=ended_reduce, to_host=True, **sample_kw)
        res = {k: v for k, v in res.items() if k in ("samples", "log_prob", "steps")}
    if world == 1:
        return res
    parts = [None] * world
    dist.all_gather_object(parts, res)
    parts = [p for p in parts if p is not None]
    steps = parts[0]["steps"]
    assert all(p["steps"] == steps for p in parts), [p["steps"] for p in parts]
    return dict(samples=torch.cat([p["samples"] for p in parts], 0), log_prob=torch.cat([p["log_prob"] for p in parts], 0), steps=steps)


def sdf_query_sharded(vq, code_ind, grid_Q, dist=None, sigmoid=False):
    """SURVEY section 8(e), the other single-shape option: the Q^3 occupancy lattice of `decode_index` (vqdif.py:60-76 driven by
    shapeformer.py:382-391; dec.py:62-100) split into slabs of lattice planes over the ranks.  Every rank holds the same code grid and
    computes the decoder feature grid itself (UNet3D + Upsampler: 0.6 ms per shape, replicated - cheaper than shipping 33.5 MB per
    shape); rank r evaluates the planes [Q r / W, Q (r + 1) / W) of the slowest lattice index (`VQDIF.decode_index(x_range=...)`,
    csrc/sdf_query.hip: a contiguous range of lattice points) and the slabs are all-gathered - the one exchange of this path.
    -> dict(logits (B, Q^3, 1)) on every rank, bit for bit what `decode_index` returns in one process: a point's arithmetic does not
    depend on which slab it is in (tests/test_ddp_gpu.py)."""
    Q = int(grid_Q)
    world = 1 if dist is None or not dist.is_initialized() else dist.get_world_size()
    rank = 0 if world == 1 else dist.get_rank()
    if world == 1:
        return vq.decode_index(code_ind, grid_Q=Q, sigmoid=sigmoid)
    if Q < world:
        raise ValueError(f"sdf_query_sharded: {Q} lattice planes for {world} ranks")
    x0, x1 = Q * rank // world, Q * (rank + 1) // world
    part = vq.decode_index(code_ind, grid_Q=Q, sigmoid=sigmoid, x_range=(x0, x1))["logits"]      # (B, (x1 - x0) Q^2, 1)
    B = part.shape[0]
    planes = -(-Q // world)                       # slabs differ by at most one plane: pad to the largest, gather, cut
    pad = torch.zeros(B, planes * Q * Q, 1, device=part.device, dtype=part.dtype)
    pad[:, :part.shape[1]] = part
    staged = dist.get_backend() == "gloo" and pad.is_cuda      # gloo has no device path (the 2-process tests on a 1-GPU box)
    send = pad.cpu() if staged else pad
    got = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(got, send)
    out = torch.cat([got[r][:, :(Q * (r + 1) // world - Q * r // world) * Q * Q] for r in range(world)], 1)
    return dict(logits=out.to(part.device) if staged else out)
