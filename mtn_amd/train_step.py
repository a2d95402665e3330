"""One training step of the reference's batch loop (train.py:29-40 + data_utils.py:123-156) as a replayable unit:
zero glue grads -> forward -> generator + label-smoothed loss -> backward -> [gradient all-reduce] -> fused Noam/Adam.

On one GPU the whole step is captured into a single hipGraph (≈460 short kernels: launch-bound if driven from
Python).  With data parallelism the backward pass is cut at the decoder-layer boundaries (model.forward_segmented) and
the step becomes N+3 hipGraphs — [forward + loss + top layer] [layer N-2] … [layer 0] [encoder side] [Adam] — with the RCCL
all-reduce of each finished gradient slice (one layer = 17 M floats = 67 MB at cfg2) launched eagerly, asynchronously,
right after the graph that produced it: the exchange of layer k runs on RCCL's stream while layer k-1's backward runs.
Collectives stay out of graph capture on purpose.  ``overlap=False`` (or MTN_DP_OVERLAP=0) keeps
the simpler schedule: one graph for forward+backward, the whole-buffer all-reduce, one graph for Adam.
"""
from __future__ import annotations

import os
from typing import Callable, Optional

import torch

from .data_utils import FusedAdam, LabelSmoothing, NoamOpt, SimpleLossCompute

# Stream capture checks "potentially unsafe" runtime calls; in the default (global) mode a call from ANY thread invalidates the
# capture — and ProcessGroupNCCL's watchdog thread polls its work events (hipEventQuery) for a while after every collective, so a
# step captured right after an exchange (a new batch shape in BucketedTrainer, the fallback schedule, a second TrainStep) died
# with hipErrorStreamCaptureInvalidated (seen on hardware: tools/dp_rccl_probe.py, round 3).  Only this thread's calls matter here:
# the kernels come from this thread and from autograd's device thread, both onto the capturing stream.
CAPTURE_MODE = os.environ.get("MTN_CAPTURE_MODE", "thread_local")
_CUT_GACC = True      # gradients across a layer cut meet in the memory-gradient buffer (no autograd add)


class TrainStep:
    def __init__(self, model, batch, vocab: int, pad: int = 1, warmup: int = 4000, factor: float = 1.0, lam: float = 1.0,
                 smoothing: float = 0.1, grad_sync=None, use_graph: bool = True, overlap: Optional[bool] = None, opt=None,
                 dynamic_norms: bool = False, fuse_optimizer: Optional[bool] = None):
        """``opt``: share an existing NoamOpt (several TrainSteps over one model, e.g. one per batch shape).
        ``dynamic_norms``: the batch tensors are refilled in place between steps, so the loss normalisers (token counts,
        train.py:35-39) are recomputed from them inside the step instead of once at construction.
        ``fuse_optimizer``: None = on one rank, apply Adam to the sublayer weight matrices inside their parameter-gradient
        GEMMs (FusedAdam.fuse_into_backward; those gradients are then never written to ``.grad``); False = always the
        separate optimiser pass (gradients of every parameter are left in ``.grad`` after the step)."""
        self.model, self.batch = model, batch
        self.opt = opt if opt is not None else NoamOpt(model.decoder.layers[0].size, factor, warmup, FusedAdam(model))
        self.dynamic_norms = dynamic_norms
        self.crit = LabelSmoothing(vocab, pad, smoothing)
        self.lc = SimpleLossCompute(model.generator, model.auto_encoder_generator, self.crit, opt=None, l=lam, sync=False)
        self.grad_sync = grad_sync
        self.pad = pad
        self.use_graph = use_graph
        self._g_fb = self._g_opt = None
        self._loss = None
        if overlap is None:
            overlap = os.environ.get("MTN_DP_OVERLAP", "1") != "0"
        self.overlap = bool(overlap) and grad_sync is not None
        # data parallel: reduce-scatter + optimiser on this rank's shard + all-gather of the master weights
        # (dp.ShardedOptimizerSync) instead of all-reduce + the full optimiser pass on every rank; MTN_DP_SHARDED=0 = the latter
        self.sharded = None
        if grad_sync is not None and os.environ.get("MTN_DP_SHARDED", "1") != "0" and hasattr(self.opt.optimizer, "step_range"):
            from .dp import ShardedOptimizerSync
            if getattr(grad_sync, "compress", False):
                raise ValueError("bf16 gradient compression applies to the all-reduce scheme only (MTN_DP_SHARDED=0): the sharded "
                                 "optimiser reduce-scatters the fp32 gradient in place")
            adam = self.opt.optimizer
            group = getattr(grad_sync, "group", None)
            sh = getattr(adam, "_sharded", None)          # ONE exchange object per optimiser: every TrainStep over it (one per
            if sh is None or sh.group is not group:       # batch shape, BucketedTrainer) shares the shard ownership and the
                sh = ShardedOptimizerSync(lambda: model.flat_buffers()[0], lambda: model.flat_buffers()[2], adam.step_range, group=group,
                                          force=getattr(grad_sync, "force", None),
                                          lp_fn=lambda: model.flat_buffers()[1], update_lp=getattr(adam, "step_range_lp", None))
                adam._sharded = sh                        # slice set that state_dict()'s gather walks
                # nobody may read the fp32 masters whole while foreign shards are stale (compute-dtype gather):
                # model.state_dict() and model.prepare() bring them up to date through this hook (a collective)
                model._master_sync = adam.gather_masters
            self.sharded = sh
        self._g_seg = None
        self._st = None
        self._fuse_opt = False if fuse_optimizer is False else None
        ae_y = batch.cap if model.auto_encoder_ft in ("caption", "summary") else batch.query
        self._ae_y = ae_y
        # loss normalisers (train.py:35-39).  Under DP they are all-reduced ONCE here for a static batch so that
        # N ranks x local batch == one rank x concatenated batch.
        self._norms = torch.stack([batch.ntokens, (ae_y != pad).sum()]).float()
        if grad_sync is not None:
            grad_sync.all_reduce_scalars(self._norms)

    # ---- pieces
    def local_norms(self):
        b = self.batch
        return torch.stack([(b.trg_y != self.pad).sum(), (self._ae_y != self.pad).sum()]).float()

    def _refresh_norms(self):
        """dynamic_norms on ONE rank: recomputed inside the (captured) step.  Under data parallelism the global counts need a
        collective, which stays out of graph capture: the caller refreshes them eagerly (refresh_norms_eager) before the step."""
        if self.dynamic_norms and self.grad_sync is None:
            self._norms.copy_(self.local_norms())

    def refresh_norms_eager(self):
        if self.dynamic_norms and self.grad_sync is not None:
            n = self.local_norms()
            self.grad_sync.all_reduce_scalars(n)
            self._norms.copy_(n)

    def _fwd_bwd(self, fuse: bool = False):
        m, b = self.model, self.batch
        self._refresh_norms()
        m.zero_glue_grads()
        if m._queue is not None and (m._queue.gemm or m._queue.ln or m._queue._armed):
            m._queue.reset()               # left over from a step that raised: never mix it into this one
        if getattr(self, "sharded", None) is not None and (self.sharded._pending is not None or self.sharded._works):
            self.sharded.abort()           # ... nor its deferred shard update / gather
        if fuse:
            self.opt.begin_fused_step()
        try:
            out, ae_out = m.forward(b)
            loss = self.lc.loss(out, b.trg_y, self._norms[0], ae_out, self._ae_y, self._norms[1])
            loss.backward(self._unit_grad(loss))          # (a kept ones scalar: autograd would launch a fill for the implicit one)
        except BaseException:
            if m._queue is not None:
                m._queue.reset()
            if getattr(self, "sharded", None) is not None:
                self.sharded.abort()
            raise
        return loss.detach()

    def _unit_grad(self, loss):
        g = getattr(self, "_one", None)
        if g is None or g.device != loss.device or g.dtype != loss.dtype:
            g = self._one = torch.ones((), device=loss.device, dtype=loss.dtype)
        return g

    def _optim(self):
        self.opt.step()

    def _fused(self) -> bool:
        """One rank: the optimiser rides on the parameter-gradient GEMMs (no gradient exchange to wait for)."""
        if self._fuse_opt is None:
            self._fuse_opt = self.grad_sync is None and hasattr(self.opt, "begin_fused_step") and self.opt.optimizer.can_fuse()
        return self._fuse_opt

    def _step_fused(self):
        loss = self._fwd_bwd(fuse=True)
        self.opt.finish_fused_step()
        return loss

    # ---- layer-segmented backward (data parallel, overlapped exchange)
    def _segments(self):
        """[(callable, (lo, hi) slice of the flat gradient buffer that is final once the callable has run)]"""
        m = self.model
        m.prepare()
        sl = m._layer_slices
        total = m._flat_grad.numel()
        N = len(sl)
        assert all(sl[k][1] == sl[k + 1][0] for k in range(N - 1)), "layer slices not contiguous"

        def bwd(outs, leaves):
            # An auto-encoder output is BOTH the next layer's chain input (across the cut: its gradient arrives as leaf.grad) and the
            # memory of x's attention inside this layer.  In the monolithic graph the two gradients meet in one buffer (ops.py, the
            # `_mtn_gacc` protocol: first writer = the chain input's dx, the memory-gradient GEMM accumulates into it); across a cut
            # autograd would ADD them (12 torch add launches per step, 112 us at batch 64: profiles/r04_x_*).  Here the leaf's gradient
            # is handed to that buffer slot instead of to autograd: same summation order as the monolithic step, no add launch.
            roots, grads = [], []
            for o, l in zip(outs, leaves):
                if l.grad is None:
                    continue
                ga = getattr(o, "_mtn_gacc", None)
                if (_CUT_GACC and ga is not None and ga["remaining"] > 0 and ga["buf"] is None and l.grad.is_contiguous()
                        and l.grad.dtype == torch.float32 and l.grad.shape == o.shape):
                    ga["buf"] = l.grad
                    # the next layer handed THIS tensor (its dx, through the producer's output dropout, compute dtype) to the producer
                    # of `o` as a ready-made dy, valid while the tensor is untouched — it is about to be accumulated into: withdraw
                    # it (in bf16 mode the memory-gradient GEMM's epilogue hands the full sum over again; fp32 mode casts it)
                    h = getattr(o, "_mtn_next", None)
                    if h is not None and h.get("dx_ptr") == l.grad.data_ptr():
                        h["dyl"] = h["dx_ptr"] = None
                    continue
                roots.append(o); grads.append(l.grad)
            if roots:
                torch.autograd.backward(roots, grads)

        def top():
            b = self.batch
            self._refresh_norms()
            m.zero_glue_grads()
            st = m.forward_segmented(b)
            self._st = st
            loss = self.lc.loss(st["out"], b.trg_y, self._norms[0], st["ae_out"], self._ae_y, self._norms[1])
            loss.backward(self._unit_grad(loss))                # loss head + final LayerNorms
            ins, outs = st["layers"][N - 1]
            bwd(outs, st["top_in"])                             # top decoder layer
            self._loss_t = loss.detach()

        def layer(k):
            def run():
                st = self._st
                ins, outs = st["layers"][k]
                bwd(outs, st["layers"][k + 1][0])
            return run

        def enc():
            st = self._st
            bwd(st["enc_out"], st["enc_leaf"])                  # embeddings, feature Linears, Encoder LayerNorms
            self._st = None

        # a layer's slice is its weight matrices (mtn.py _ordered_params): mat_hi = hi -> they travel in the compute dtype; the last
        # slice = glue parameters + every vector (biases, all LayerNorms): fp32 throughout, complete once the last segment has run
        assert sl[N - 1][1] == total, "the last layer's matrices end the flat buffer"
        segs = [(top, (sl[N - 1][0], total, total))]
        segs += [(layer(k), sl[k] + (sl[k][1],)) for k in range(N - 2, -1, -1)]
        segs.append((enc, (0, sl[0][0], None)))
        return segs

    def _run_segmented(self, runners):
        if self.sharded is not None:
            if self.sharded._pending is not None or self.sharded._works:
                self.sharded.abort()                    # left over from a step that raised between reduce_update() and finish()
            self.opt.begin_sharded_step()
            try:
                for run, (lo, hi, mat_hi) in runners:
                    run()
                    self.sharded.reduce_update(lo, hi, mat_hi)
            except BaseException:
                self.sharded.abort()                    # never let this step's deferred shard update run against the next step's gradients
                if self.model._queue is not None:
                    self.model._queue.reset()
                raise
            if self.sharded.timeline is not None and torch.cuda.is_available():
                e = torch.cuda.Event(enable_timing=True)
                e.record()                              # the compute chain (every segment graph) ends here; what follows is exposed exchange
                self.sharded.timeline.append({"chain_end": e})
            self.sharded.finish()
            return
        works = []
        for run, (lo, hi, _) in runners:
            run()
            works.append(self.grad_sync.reduce_range(lo, hi))
        self.grad_sync.wait(works)

    def _capture(self):
        m = self.model
        m.prepare()
        self._fused()                               # decides (and builds its device-side tables) outside the capture
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):               # allocator / autograd warm-up off the capture stream
            for _ in range(2):
                self._fwd_bwd()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if self.overlap:
            err = None
            try:
                self._capture_segmented(side)
            except Exception as e:      # never lose the run to the more elaborate schedule: fall back to the simple one
                err = e
            # the decision is COLLECTIVE: a rank that fell back alone would issue a different sequence of collectives
            failed = torch.tensor([1.0 if err is not None else 0.0], device=self.model._flat.device)
            self.grad_sync.all_reduce_scalars(failed)
            if float(failed.item()) == 0.0:
                return
            import logging
            logging.getLogger("mtn_amd").warning("layer-segmented capture failed on %d rank(s) (%s); using the two-graph schedule",
                                                 int(failed.item()), err)
            self.overlap, self._g_seg, self._st = False, None, None
            torch.cuda.synchronize()
        self._capture_simple()

    def _capture_segmented(self, side):
        segs = self._segments()
        with torch.cuda.stream(side):
            for fn, _ in segs:                      # warm-up of the segmented schedule (no exchange: gradients are discarded)
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._g_seg, pool = [], None
        for fn, rng in segs:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, capture_error_mode=CAPTURE_MODE):
                fn()
            pool = g.pool()
            self._g_seg.append((g.replay, rng))
        self._loss = self._loss_t
        self._g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_opt, pool=pool, capture_error_mode=CAPTURE_MODE):
            if self.sharded is not None:
                self._refresh_after_exchange()          # the updates ran shard by shard between the segment graphs
            else:
                self._optim()
        self._g_fb = self._g_seg[0]

    def _capture_simple(self):
        self._g_fb = torch.cuda.CUDAGraph()
        if self.grad_sync is None:
            with torch.cuda.graph(self._g_fb, capture_error_mode=CAPTURE_MODE):
                if self._fused():
                    self._loss = self._step_fused()
                else:
                    self._loss = self._fwd_bwd()
                    self._optim()
        else:
            with torch.cuda.graph(self._g_fb, capture_error_mode=CAPTURE_MODE):
                self._loss = self._fwd_bwd()
            self._g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g_opt, pool=self._g_fb.pool(), capture_error_mode=CAPTURE_MODE):
                if self.sharded is not None:
                    self._refresh_after_exchange()
                else:
                    self._optim()
        # the warm-up/capture passes did not run the optimiser outside capture: parameters are untouched

    def __call__(self) -> torch.Tensor:
        """Runs one step; returns the device tensor of the NORMALISED loss (main KL / global target tokens + lambda * sum of the
        auto-encoder KLs / global query tokens: the `loss` of data_utils.py:135-144, before the `* norm` of :156), without
        synchronising."""
        if not self.use_graph:
            if self.overlap:
                self._run_segmented(self._segments())
                loss = self._loss_t
            elif self._fused():
                return self._step_fused()
            else:
                loss = self._fwd_bwd()
                if self.sharded is not None:
                    self._whole_buffer_sharded()
                elif self.grad_sync is not None:
                    self.grad_sync()
            if self.sharded is not None:
                self._refresh_after_exchange()
            else:
                self._optim()
            return loss
        if self._g_fb is None:
            self._capture()
        if self.overlap:
            self._run_segmented(self._g_seg)
            self._g_opt.replay()
            return self._loss
        self._g_fb.replay()
        if self.sharded is not None:
            self._whole_buffer_sharded()
            self._g_opt.replay()
        elif self.grad_sync is not None:
            self.grad_sync()
            self._g_opt.replay()
        return self._loss

    def _refresh_after_exchange(self):
        """After the sharded exchange: with the compute-dtype gather the matrices' bf16 copies arrived by all-gather, only the glue
        slice (generator, feature Linears: fp32 masters gathered) is cast; otherwise every weight is (the fp32 masters were gathered)."""
        lp_gather = self.sharded is not None and self.sharded.lp_mode()
        self.opt.optimizer.refresh_copies(self.model._layer_slices[0][0] if lp_gather else None)

    def _slices(self):
        """The (lo, hi) ranges of the flat buffers the exchange works in, last-finished-first: [top layer .. end], layers
        N-2 .. 0, [glue + encoder norms].  BOTH schedules use them, so which rank owns (and keeps the Adam moments of) an element
        never depends on the schedule a step ran under."""
        m = self.model
        m.prepare()
        sl = m._layer_slices
        total, N = m._flat_grad.numel(), len(sl)
        return [(sl[N - 1][0], total, total)] + [sl[k] + (sl[k][1],) for k in range(N - 2, -1, -1)] + [(0, sl[0][0], None)]

    def _whole_buffer_sharded(self):
        """The simple schedule (MTN_DP_OVERLAP=0, or the fallback) with the sharded optimiser: same slices as the segmented one."""
        self.opt.begin_sharded_step()
        for lo, hi, mat_hi in self._slices():
            self.sharded.reduce_update(lo, hi, mat_hi)
        self.sharded.finish()


class BucketedTrainer:
    """The reference's per-batch loop body (train.py:29-40) for batches of VARYING shape, on captured graphs: batch lengths
    are rounded up to multiples of ``bucket`` (padding is masked everywhere on the path, so the real tokens see the same
    arithmetic), one static Batch + one captured TrainStep is kept per padded shape, and every step refills that Batch in
    place on the device (data_handler.make_batch(out=...)) and replays its graph.  One optimiser state for all shapes."""

    def __init__(self, model, corpus, vocab_size: int, pad: int = 1, warmup: int = 4000, lam: float = 1.0, bucket: int = 8,
                 grad_sync=None, max_shapes: int = 64):
        self.model, self.corpus, self.vocab, self.pad, self.lam = model, corpus, vocab_size, pad, lam
        self.bucket, self.grad_sync, self.max_shapes = bucket, grad_sync, max_shapes
        self.opt = NoamOpt(model.decoder.layers[0].size, 1.0, warmup, FusedAdam(model))
        self.steps = {}

    def _padded(self, index):
        up = lambda v: -(-int(v) // self.bucket) * self.bucket
        x_len, h_len, q_len, a_len, c_len, n = index[2:]
        return (index[0], index[1], [up(v) for v in x_len], up(h_len), up(q_len), up(a_len), up(c_len), n)

    def step(self, index):
        """One optimiser step on the batch described by ``index`` (an entry of data_handler.make_batch_indices with
        separate_caption=True).  Returns (device loss tensor, the static Batch that now holds this batch)."""
        from .data_handler import make_batch
        pidx = self._padded(index)
        key = (tuple(pidx[2]),) + tuple(pidx[3:])
        hit = self.steps.get(key)
        if hit is None:
            if len(self.steps) >= self.max_shapes:
                self.steps.pop(next(iter(self.steps)))
            batch = make_batch(self.corpus, pidx, self.pad, separate_caption=True)
            ts = TrainStep(self.model, batch, self.vocab, pad=self.pad, lam=self.lam, grad_sync=self.grad_sync, opt=self.opt,
                           dynamic_norms=True)
            self.steps[key] = hit = (batch, ts)
        else:
            make_batch(self.corpus, pidx, self.pad, separate_caption=True, out=hit[0])
        hit[1].refresh_norms_eager()
        hit[0]._norms_global = hit[1]._norms        # [target tokens, auto-encoder tokens] the step's loss is divided by (all ranks)
        return hit[1](), hit[0]
