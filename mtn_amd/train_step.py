"""One training step of the reference's batch loop (train.py:29-40 + data_utils.py:123-156) as a replayable unit:
zero glue grads -> forward -> generator + label-smoothed loss -> backward -> [gradient all-reduce] -> fused Noam/Adam.

On one GPU the whole step is captured into a single hipGraph (≈1300 short kernels: launch-bound if driven from
Python).  With data parallelism the step is two graphs with the RCCL all-reduce of the flat gradient buffer
launched eagerly between them (collectives are kept out of graph capture on purpose: see DESIGN.md §multi-GPU).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from .data_utils import FusedAdam, LabelSmoothing, NoamOpt, SimpleLossCompute


class TrainStep:
    def __init__(self, model, batch, vocab: int, pad: int = 1, warmup: int = 4000, factor: float = 1.0, lam: float = 1.0,
                 smoothing: float = 0.1, grad_sync=None, use_graph: bool = True):
        self.model, self.batch = model, batch
        self.opt = NoamOpt(model.decoder.layers[0].size, factor, warmup, FusedAdam(model))
        self.crit = LabelSmoothing(vocab, pad, smoothing)
        self.lc = SimpleLossCompute(model.generator, model.auto_encoder_generator, self.crit, opt=None, l=lam, sync=False)
        self.grad_sync = grad_sync
        self.pad = pad
        self.use_graph = use_graph
        self._g_fb = self._g_opt = None
        self._loss = None
        ae_y = batch.cap if model.auto_encoder_ft in ("caption", "summary") else batch.query
        self._ae_y = ae_y
        # loss normalisers (train.py:35-39).  Under DP they are all-reduced ONCE here for a static batch so that
        # N ranks x local batch == one rank x concatenated batch.
        self._norms = torch.stack([batch.ntokens, (ae_y != pad).sum()]).float()
        if grad_sync is not None:
            grad_sync.all_reduce_scalars(self._norms)

    # ---- pieces
    def _fwd_bwd(self):
        m, b = self.model, self.batch
        m.zero_glue_grads()
        out, ae_out = m.forward(b)
        loss = self.lc.loss(out, b.trg_y, self._norms[0], ae_out, self._ae_y, self._norms[1])
        loss.backward()
        return loss.detach()

    def _optim(self):
        self.opt.step()

    def _capture(self):
        m = self.model
        m.prepare()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):               # allocator / autograd warm-up off the capture stream
            for _ in range(2):
                self._fwd_bwd()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._g_fb = torch.cuda.CUDAGraph()
        if self.grad_sync is None:
            with torch.cuda.graph(self._g_fb):
                self._loss = self._fwd_bwd()
                self._optim()
        else:
            with torch.cuda.graph(self._g_fb):
                self._loss = self._fwd_bwd()
            self._g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g_opt, pool=self._g_fb.pool()):
                self._optim()
        # the warm-up/capture passes did not run the optimiser outside capture: parameters are untouched

    def __call__(self) -> torch.Tensor:
        """Runs one step; returns the (device) loss tensor of data_utils.py:156 without synchronising."""
        if not self.use_graph:
            loss = self._fwd_bwd()
            if self.grad_sync is not None:
                self.grad_sync()
            self._optim()
            return loss
        if self._g_fb is None:
            self._capture()
        self._g_fb.replay()
        if self.grad_sync is not None:
            self.grad_sync()
            self._g_opt.replay()
        return self._loss
