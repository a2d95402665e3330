"""Data parallelism for the MTN train step: one process per GPU, identical replicas, minibatch sharded by sample,
ONE exchange per step — a sum all-reduce of the flat gradient buffer (RCCL over xGMI when the backend is "nccl";
gloo in the CPU tests) between ``loss.backward()`` and the optimiser step (the reference has no counterpart; the
insertion point is data_utils.py:153->154).

Exactness: every rank normalises its loss by the GLOBAL token counts (two scalars all-reduced before the loss), so
N ranks x local batch produce the gradient of one rank x concatenated batch.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring is per-link bound, so the gradient travels as a few
large buckets (default 4 x ~107 MB fp32 for the 106.65 M-parameter model) rather than many small ones, optionally
compressed to bf16 (halves the bytes on the links; the sum is then rounded — off by default).
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


ALLOW_EMULATION = False      # set by `bench.py --no-record` only: lets MTN_DP_EMULATE_WORLD (wrong results, timing probe) through


def init_distributed(backend: Optional[str] = None, force: Optional[bool] = None) -> tuple:
    """(rank, world_size, local_rank) from the torchrun environment; initialises the default process group.
    ``force`` (default: MTN_FORCE_DIST=1): create the group even for ONE rank, so that the RCCL path runs on a 1-GPU box."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if force is None:
        force = os.environ.get("MTN_FORCE_DIST") == "1"    # exercise the RCCL path with a single rank (1-GPU boxes)
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("MTN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw = {}
            try:       # the overlapped gradient exchange shares the GPU with backward kernels: give RCCL's stream priority
                opts = dist.ProcessGroupNCCL.Options()
                opts.is_high_priority_stream = True
                kw["pg_options"] = opts
            except Exception:
                pass
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local), **kw)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_items: int, rank: int, world: int) -> tuple:
    """Contiguous split of a global batch by rank (first ranks take the remainder)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class GradSync:
    """Bucketed sum all-reduce of a flat gradient buffer.  ``flat_grad_fn`` returns the buffer (so the object survives a
    re-flatten); works on CUDA (RCCL) and CPU (gloo) tensors alike — the CPU tests drive it with the oracle model."""

    def __init__(self, flat_grad_fn, group=None, n_buckets: int = 4, compress_bf16: bool = False, force: Optional[bool] = None):
        self.flat_grad_fn = flat_grad_fn
        self.group = group
        self.n_buckets = max(1, n_buckets)
        self.compress = compress_bf16
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if force is None:
            force = os.environ.get("MTN_FORCE_DIST") == "1"
        self.force = bool(force) and dist.is_initialized()       # one rank: issue the collectives anyway (RCCL on a 1-GPU box)

    def buckets(self, n: int):
        per = -(-n // self.n_buckets)
        per = (per + 1023) // 1024 * 1024
        return [(s, min(n, s + per)) for s in range(0, n, per)]

    def all_reduce_scalars(self, t: torch.Tensor):
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def broadcast_(self, t: torch.Tensor, src: int = 0):
        if self.world > 1:
            dist.broadcast(t, src=src, group=self.group)
        return t

    def reduce_range(self, lo: int, hi: int):
        """Asynchronous sum all-reduce of flat_grad[lo:hi] (issued behind everything already queued on the current stream;
        later work on the current stream does NOT wait for it).  Returns a handle for wait()."""
        if (self.world == 1 and not self.force) or hi <= lo:
            return None
        chunk = self.flat_grad_fn()[lo:hi]
        if self.compress:
            c16 = chunk.to(torch.bfloat16)
            return (dist.all_reduce(c16, op=dist.ReduceOp.SUM, group=self.group, async_op=True), chunk, c16)
        return (dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True), None, None)

    def wait(self, handles):
        """Make the current stream wait for the exchanges issued by reduce_range()."""
        for h in handles:
            if h is None:
                continue
            work, chunk, c16 = h
            work.wait()
            if c16 is not None:
                chunk.copy_(c16)

    def __call__(self):
        if self.world == 1 and not self.force:
            return
        g = self.flat_grad_fn()
        for s, e in self.buckets(g.numel()):
            chunk = g[s:e]
            if self.compress:
                c16 = chunk.to(torch.bfloat16)
                dist.all_reduce(c16, op=dist.ReduceOp.SUM, group=self.group)
                chunk.copy_(c16)
            else:
                dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group)


class ShardedOptimizerSync:
    """Data parallelism without a replicated optimiser pass (ZeRO-1 style), slice by slice:

        reduce-scatter of a gradient slice  ->  optimiser update of THIS rank's 1/N of the slice  ->  all-gather of the updated
        fp32 master weights of the slice

    instead of  all-reduce -> the full Adam pass on every rank.  The bytes on the links are those of the all-reduce (a ring
    all-reduce IS a reduce-scatter followed by an all-gather), but every rank streams 38 B/param of optimiser state for 1/N of
    the parameters instead of all of them, and the replicas are identical by construction (every element is summed and updated
    exactly once, then copied).  With the layer-segmented backward (train_step.TrainStep) slice k's reduce-scatter runs on the
    collective stream while the compute stream is in the backward of layer k-1; its shard update follows THAT segment on the
    compute stream and its all-gather runs under layer k-2 (see reduce_update); only the last slice's chain is exposed.  After the last slice, ``finish()`` waits and the caller refreshes the compute-dtype copies (cast +
    transposes of everything: 10 B/param, local).

    ``update(off, n)`` applies the optimiser to flat elements [off, off+n) (HIP: mtn_adam_step on the sub-buffers; the CPU
    tests pass a torch reference).  Slices whose length is not a multiple of 4*world leave a tail of < 4*world elements that
    is all-reduced and updated redundantly on every rank.

    Compute-dtype gather (round 4; ``lp_fn`` + ``update_lp`` given and a slice passed with ``mat_hi``).  The forward and backward
    passes read the weight MATRICES through their bf16 copy only, so a rank needs the fp32 master of its own shard and nothing
    else: for a slice laid out [matrices | vectors] = [lo, mat_hi) + [mat_hi, hi) the matrices are reduce-scattered, the shard's
    update also writes its bf16 copy (``update_lp``), and THAT is all-gathered — 2 B per parameter on the links instead of 4, and
    no cast pass over every weight after the exchange; the few vectors (biases, LayerNorm gains: read in fp32 by the kernels) are
    all-reduced and updated on every rank.  fp32 masters of foreign shards go stale and are brought back by ``gather()`` for
    checkpoints, like the Adam moments.
    Collectives: RCCL in-place reduce_scatter_tensor / all_gather_into_tensor; other backends (gloo on CUDA tensors in the
    single-GPU tests) take an all-reduce + per-shard broadcasts with the same results."""

    def __init__(self, flat_fn, grad_fn, update, group=None, force: Optional[bool] = None, lp_fn=None, update_lp=None):
        self.flat_fn, self.grad_fn, self.update, self.group = flat_fn, grad_fn, update, group
        self.lp_fn, self.update_lp = lp_fn, update_lp          # compute-dtype copy of the flat buffer + the update that also writes it
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.native = dist.is_initialized() and dist.get_backend(group) == "nccl"
        # A one-rank RCCL group (MTN_FORCE_DIST=1 on a 1-GPU box) takes the SAME collective chain as N ranks — in-place
        # reduce_scatter_tensor / all_gather_into_tensor, pipelined update, finish() — with per = the whole slice and own = lo, so
        # that the code an N-GPU job executes has run (and is tested bit for bit against the fused one-rank step) before it.
        if force is None:
            force = os.environ.get("MTN_FORCE_DIST") == "1"
        self.collective = self.world > 1 or (dist.is_initialized() and bool(force))
        self._stamp, self._stamps = None, False      # (timeline measurement: the stream the collectives' ends are stamped on)
        self._pending = None                         # the previous slice's wait -> update -> gather, issued behind the next segment
        self._works = []
        self.slices = set()             # every (lo, hi) this object split into shards (for gather())
        self.lp_slices = set()          # ... of them, the matrix ranges whose foreign fp32 masters are stale (gathered in the compute dtype)
        self.masters_stale = False      # a step ran the compute-dtype gather since the last gather() of the fp32 masters
        # measurement (bench.py secondary.exchange.timeline): when a list, every reduce_update() appends HIP events of its chain —
        # gradient slice ready / reduced / update begins / shard updated / gathered (compute stream; collectives' ends on a stamp stream)
        self.timeline = None
        self.calls = {"reduce_scatter": 0, "all_reduce": 0, "all_gather": 0, "broadcast": 0}   # collectives issued (tests, bench line)

    def lp_mode(self) -> bool:
        """Is the compute-dtype gather available?  (a separate low-precision copy exists and the caller gave the update for it)"""
        if self.lp_fn is None or self.update_lp is None:
            return False
        lp = self.lp_fn()
        return lp is not None and lp.dtype != torch.float32

    def gather(self, buf: torch.Tensor):
        """Make a per-element buffer complete on every rank — the Adam moments, and (compute-dtype gather) the fp32 master
        weights: each rank only ever updates its shards.  Blocking; checkpoints only."""
        if self.world == 1:
            return buf
        for lo, hi in sorted(self.slices):
            per, own, tail = self.split(lo, hi)
            for r in range(self.world if per > 0 else 0):
                dist.broadcast(buf[lo + r * per: lo + (r + 1) * per], src=r, group=self.group)
        return buf

    def split(self, lo: int, hi: int):
        """(elements per rank, first element of this rank's shard, first element of the replicated tail)"""
        per = ((hi - lo) // (self.world * 4)) * 4
        return per, lo + self.rank * per, lo + per * self.world

    def reduce_update(self, lo: int, hi: int, mat_hi: Optional[int] = None):
        """Issue the chain for flat[lo:hi] behind everything queued on the current stream; does not block the current stream.
        ``mat_hi``: the slice is [matrices | vectors] with the boundary there -> compute-dtype gather of the matrices (see the
        class docstring) when lp_mode(); None (or no lp_mode): fp32 all-gather of the whole slice."""
        if hi <= lo:
            return
        lp_gather = mat_hi is not None and lo < mat_hi <= hi and self.lp_mode()
        flat, grad = self.flat_fn(), self.grad_fn()
        shard_hi = mat_hi if lp_gather else hi            # the range that is cut into per-rank shards
        self.slices.add((lo, shard_hi))
        if lp_gather:
            self.lp_slices.add((lo, shard_hi))
            self.masters_stale = self.world > 1
        per, own, tail = self.split(lo, shard_hi)
        cuda = grad.is_cuda
        upd = self.update_lp if lp_gather else self.update
        if not self.collective:
            # one rank, no process group: the plain update.  MTN_DP_EMULATE_WORLD (update 1/emu of the slice, as a rank of an
            # emu-GPU job would: the RESULT IS WRONG) is a timing probe and only honoured when the caller opted in explicitly
            # (dp.ALLOW_EMULATION, set by `bench.py --no-record`); otherwise it is refused loudly.
            emu = int(os.environ.get("MTN_DP_EMULATE_WORLD", "1"))
            if emu > 1 and not ALLOW_EMULATION:
                raise RuntimeError("MTN_DP_EMULATE_WORLD produces wrong parameters (timing probe only): run it through "
                                   "`bench.py --no-record`, never in a recorded line or a training run")
            n = hi - lo
            upd(lo, n if emu <= 1 else max(4, (n // (emu * 4)) * 4))
            return
        emu = 1
        if self.world == 1 and ALLOW_EMULATION:             # timing probe (`bench.py --no-record`): the update of a rank of `emu`, see above
            emu = int(os.environ.get("MTN_DP_EMULATE_WORLD", "1"))
        tl = None
        if self.timeline is not None and cuda:
            tl = {"lo": lo, "hi": hi, "bytes_reduced": 4 * (tail - lo) if per > 0 else 0, "bytes_gathered": 0}
            self.timeline.append(tl)

        def mark(name):
            if tl is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                tl[name] = e

        def stamp(ws, name):               # measurement only: a separate stream waits for the collectives so that their END can be stamped
            if tl is not None:
                if self._stamp is None:
                    self._stamp = torch.cuda.Stream()
                with torch.cuda.stream(self._stamp):
                    for w in ws:
                        w.wait()
                    mark(name)
                self._stamps = True

        mark("ready")                          # the slice's gradients: everything the compute stream has queued so far
        works = []
        if per > 0:
            if self.native:
                w = dist.reduce_scatter_tensor(grad[own:own + per], grad[lo:tail], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self.calls["reduce_scatter"] += 1
            else:
                w = dist.all_reduce(grad[lo:tail], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self.calls["all_reduce"] += 1
            works.append(w)
        if tail < hi:                                      # the replicated remainder: < 4 * world matrix elements (+ the slice's vectors)
            works.append(dist.all_reduce(grad[tail:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.calls["all_reduce"] += 1
        stamp(works, "reduced")

        def complete():
            for w in works:
                w.wait()                           # the compute stream waits (the host does not): by now a whole segment was queued behind the reduce-scatter
            mark("update_begins")
            if per > 0:
                upd(own, per if emu <= 1 else max(4, (per // (emu * 4)) * 4))
            if tail < hi:
                upd(tail, hi - tail)               # replicated: identical inputs -> identical results on every rank
            mark("updated")
            if per > 0:
                buf = self.lp_fn() if lp_gather else flat          # what the other ranks need of this shard: its bf16 copy | its fp32 master
                mine = []
                if self.native:
                    mine.append(dist.all_gather_into_tensor(buf[lo:tail], buf[own:own + per], group=self.group, async_op=True))
                    self.calls["all_gather"] += 1
                else:
                    for r in range(self.world):
                        mine.append(dist.broadcast(buf[lo + r * per: lo + (r + 1) * per], src=r, group=self.group, async_op=True))
                        self.calls["broadcast"] += 1
                self._works += mine
                if tl is not None:
                    tl["bytes_gathered"] = buf.element_size() * (tail - lo)
                stamp(mine, "gathered")

        # The schedule, ONE compute stream + the collective stream:
        #     segment k | reduce-scatter k issued | segment k+1 | reduce-scatter k+1 issued | wait RS k, Adam on shard k, all-gather k issued | ...
        # The update of slice k runs on the COMPUTE stream, after the next segment: by then its reduce-scatter has had a whole segment
        # (~0.4 ms) to finish, so the wait is free; the shard is 1/N of the slice (9 us of Adam at N = 8), and nothing hops between
        # streams on the critical path.  Rounds 2-4 ran the update on a side stream under the next segment: on one rank, where the
        # "shard" is the whole slice, the two HBM-bound streams stretched each other by more than the update's own length and every
        # stream hop cost 60-100 us (profiles/r05_dp_timeline_serial.txt: 4.61 ms; serialised by rocprofv3 the same launches took 3.94).
        # Collective-stream order: RS 0, RS 1, AG 0, RS 2, AG 1, ...: a reduce-scatter never queues behind a gather that waits for an update.
        self._complete_pending()
        self._pending = complete

    def _complete_pending(self):
        c, self._pending = self._pending, None
        if c is not None:
            c()

    def abort(self):
        """A step raised between reduce_update() and finish(): drop the deferred wait -> update -> gather of its last slice (it would
        otherwise run at the NEXT step's first reduce_update(), against that step's gradients — possibly on this rank only) and let the
        collectives already in flight complete."""
        self._pending = None
        for w in self._works:
            try:
                w.wait()
            except Exception:
                pass
        self._works = []
        self._stamps = False

    def finish(self):
        """The current stream waits for every chain issued since the last finish()."""
        self._complete_pending()
        for w in self._works:
            w.wait()
        self._works = []
        if self._stamps:
            torch.cuda.current_stream().wait_stream(self._stamp)
            self._stamps = False
