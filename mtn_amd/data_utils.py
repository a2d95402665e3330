"""Batch / masks / optimiser / loss harness around the model — the surface of the reference's
data_utils.py + label_smoothing.py that train.py's batch loop drives (SURVEY.md §2 "IN as harness").

Differences that matter on MI355X: nothing here forces a host sync inside the step (the reference's
``loss.item()`` at data_utils.py:156 is optional), the optimiser is one fused Adam kernel over the flat
parameter buffer with the Noam rate computed on device (so a whole step can be captured in a hipGraph), and
tensors stay on whatever device they were created on (no hard-coded ``.cuda()``).
"""
from __future__ import annotations

import os

from typing import List, Optional

import torch
import torch.nn as nn

from . import lib as L


def subsequent_mask(size: int, device=None) -> torch.Tensor:
    """(1,size,size) bool, True where a position may be attended (data_utils.py:10-14)."""
    return torch.ones(1, size, size, dtype=torch.bool, device=device).tril_()


class Batch:
    """Holds one mini-batch with its masks (data_utils.py:21-54).  ``fts`` are (B,V,F) float tensors (already
    batch-first, on the target device) or (V,B,F) numpy arrays as data_handler.make_batch produces them."""

    def __init__(self, query, his, his_st=None, fts=None, cap=None, trg=None, trg_y=None, pad=0, device=None):
        dev = device if device is not None else query.device
        self.query, self.his, self.his_st = query.to(dev), his.to(dev), his_st
        if fts is not None:
            cleaned, masks = [], []
            for ft in fts:
                if not torch.is_tensor(ft):
                    ft = torch.from_numpy(ft).float().permute(1, 0, 2)          # data_utils.py:28
                ft = ft.to(dev)
                mask = ((ft != 1).sum(dim=2) != 0).unsqueeze(-2)                # frames padded with 1.0 (data_handler.py:236)
                cleaned.append(ft * mask.squeeze(-2).unsqueeze(-1).to(ft.dtype))
                masks.append(mask)
            self.fts, self.fts_mask = cleaned, masks
        else:
            self.fts = self.fts_mask = None
        self.query_mask = (self.query != pad).unsqueeze(-2)
        self.his_mask = (self.his != pad).unsqueeze(-2)
        if cap is not None:
            self.cap = cap.to(dev)
            self.cap_mask = (self.cap != pad).unsqueeze(-2)
        else:
            self.cap = self.cap_mask = None
        if trg is not None:
            self.trg, self.trg_y = trg.to(dev), trg_y.to(dev)
            self.trg_mask = self.make_std_mask(self.trg, pad)
            self.ntokens = (self.trg_y != pad).sum()

    @staticmethod
    def make_std_mask(tgt, pad):
        """pad mask AND causal mask (data_utils.py:48-54)."""
        return (tgt != pad).unsqueeze(-2) & subsequent_mask(tgt.size(-1), tgt.device)


class LabelSmoothing(nn.Module):
    """KLDivLoss(sum) against the smoothed target distribution (label_smoothing.py:9-32), computed in closed
    form without materialising the (rows, vocab) target matrix twice.  The reference's quirk is kept: rows
    whose target is <pad> are zeroed only if the SUM OF THEIR INDICES is > 0 (label_smoothing.py:29)."""

    def __init__(self, size: int, padding_idx: int, smoothing: float = 0.0):
        super().__init__()
        self.size, self.padding_idx, self.smoothing = size, padding_idx, smoothing
        self.confidence = 1.0 - smoothing

    def forward(self, x, target):
        assert x.size(1) == self.size
        eps = self.smoothing / (self.size - 2)
        true_dist = torch.full_like(x, eps)
        true_dist.scatter_(1, target.unsqueeze(1), self.confidence)
        true_dist[:, self.padding_idx] = 0
        pad_rows = target == self.padding_idx
        idx_sum = (torch.arange(target.numel(), device=target.device) * pad_rows).sum()
        zero_rows = pad_rows & (idx_sum > 0)                      # device-side form of the reference's host test
        true_dist = true_dist * (~zero_rows).unsqueeze(1).to(x.dtype)
        pos = true_dist > 0
        safe = torch.where(pos, true_dist, torch.ones_like(true_dist))
        return (true_dist * (safe.log() - x)).sum()


class FusedAdam:
    """Adam(betas=(0.9,0.98), eps=1e-9) (train.py:190) over the model's flat buffers in ONE kernel, with the Noam
    learning rate (data_utils.py:111-117) advanced on device.  Writes the compute-dtype weight copy in the same pass."""

    def __init__(self, model, betas=(0.9, 0.98), eps=1e-9):
        self.model = model
        self.betas, self.eps = betas, eps
        flat, _, _ = model.flat_buffers()
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        self.state = torch.zeros(8, device=flat.device, dtype=torch.float32)   # step, lr, 1-b1^t, 1-b2^t
        self.param_groups = [{"lr": 0.0, "params": list(model.parameters())}]
        self.grad_scale: Optional[torch.Tensor] = None                        # device scalar multiplied into every gradient
        self._aux_cache = {}                      # host-side chunk lists of the table launch's front tiles (ops.ParamGradQueue._table_aux)

    def tick(self, factor: float, model_size: int, warmup: int):
        L.check(L.load().mtn_noam_tick(self.state.data_ptr(), factor, model_size, warmup, self.betas[0], self.betas[1], L.stream_ptr()))

    def _buffers(self):
        flat, flat_lp, grad = self.model.flat_buffers()
        self.model._ln_fold_stale = True          # every caller is about to change weights: the LayerNorm fold vectors go stale
        if self.m.data_ptr() == 0 or self.m.numel() != flat.numel() or self.m.device != flat.device:
            raise L.MtnHipError("model was re-flattened after the optimiser was built")
        return flat, flat_lp, grad, (None if flat_lp is flat else flat_lp.data_ptr())

    def step(self):
        flat, flat_lp, grad, lp_ptr = self._buffers()
        L.check(L.load().mtn_adam_step(L.dtype_code(self.model.compute_dtype), flat.numel(), flat.data_ptr(), grad.data_ptr(),
                                       self.m.data_ptr(), self.v.data_ptr(), lp_ptr, self.state.data_ptr(),
                                       L.ptr(self.grad_scale), self.betas[0], self.betas[1], self.eps, L.stream_ptr()))
        self.model.refresh_transposed()

    def step_range(self, off: int, n: int, write_lp: bool = False):
        """The update on flat elements [off, off+n) only (data parallel: this rank's shard of a gradient slice,
        dp.ShardedOptimizerSync).  write_lp = False: the compute-dtype copies are NOT written — the caller refreshes them once all
        shards of all ranks have arrived (refresh_copies); True: the shard's compute-dtype copy is written by the same pass (the
        compute-dtype gather: the other ranks receive THAT, not the fp32 master)."""
        flat, flat_lp, grad, lp_ptr = self._buffers()
        assert off % 4 == 0 and n > 0
        lp = None
        if write_lp and lp_ptr is not None:
            lp = lp_ptr + flat_lp.element_size() * off
        L.check(L.load().mtn_adam_step(L.dtype_code(self.model.compute_dtype), n, flat.data_ptr() + 4 * off, grad.data_ptr() + 4 * off,
                                       self.m.data_ptr() + 4 * off, self.v.data_ptr() + 4 * off, lp, self.state.data_ptr(),
                                       L.ptr(self.grad_scale), self.betas[0], self.betas[1], self.eps, L.stream_ptr()))

    def step_range_lp(self, off: int, n: int):
        self.step_range(off, n, write_lp=True)

    def refresh_copies(self, hi: Optional[int] = None):
        """Compute-dtype weight copy + transposed copies from the fp32 master (after a sharded update + all-gather).  ``hi``:
        only flat elements [0, hi) are cast (compute-dtype gather: the matrices' copies arrived by all-gather already; what is left
        is the glue slice — generator / feature Linears — whose masters were gathered in fp32)."""
        m = self.model
        flat, flat_lp, _, lp_ptr = self._buffers()
        if lp_ptr is not None:
            n = flat.numel() if hi is None else min(int(hi), flat.numel())
            if n > 0:
                L.check(L.load().mtn_cast_f32_to_lp(L.dtype_code(m.compute_dtype), n, flat.data_ptr(), lp_ptr, L.stream_ptr()))
        m.refresh_transposed()

    # ---- optimiser epilogue: the update of the sublayer weight matrices rides on their parameter-gradient GEMMs
    def can_fuse(self) -> bool:
        m = self.model
        m.prepare()
        ok = bool(getattr(m, "_fusable", None)) and m._flat.is_cuda and os.environ.get("MTN_NO_FUSED_ADAM") is None
        if ok:      # build the two expected "rest" tables now: they are device tensors and the step may be under graph capture later
            every = frozenset(t[0] for t in m._fusable)
            m.rest_tables(every)
            m.rest_tables(every - m._fusable_optional)
        return ok

    def fuse_into_backward(self, write_grad: bool = False):
        """Arm the model's parameter-gradient queue: its next flush (the end of the coming backward pass) applies THIS
        step's update to every sublayer weight matrix inside the dW GEMM that produces its gradient (include/mtn_hip.h
        mtn_adam_fuse) — no gradient round trip through HBM, no separate transpose pass.  The schedule must already have
        been advanced (tick) and ``step_rest()`` must follow the backward pass.  Single-rank only: with data parallelism
        the gradients are exchanged between backward and the update."""
        flat, flat_lp, grad, lp_ptr = self._buffers()
        m = self.model
        esz = flat_lp.element_size()
        self._armed = dict(fusable=m._fusable, starts=[t[0] for t in m._fusable], optional=m._fusable_optional, covered=frozenset(),
                           grad=grad.data_ptr(), p=flat.data_ptr(), m=self.m.data_ptr(), v=self.v.data_ptr(), lp=lp_ptr,
                           lpT=m._flat_lpT.data_ptr() if m._flat_lpT is not None else None,
                           t_map=dict(m._t_off), esz=esz, write_grad=write_grad,
                           state=self.state.data_ptr(), grad_scale=L.ptr(self.grad_scale), betas=self.betas, eps=self.eps, applied=False,
                           n_flat=flat.numel(), rest_done=False, aux_cache=self._aux_cache)
        m._queue.adam = self._armed

    def step_rest(self):
        """After a backward pass armed by fuse_into_backward(): the update of everything the GEMM epilogues did not touch."""
        m = self.model
        armed, m._queue.adam = m._queue.adam, None
        if armed is None or not armed["applied"]:
            raise L.MtnHipError("optimiser epilogue was armed but the backward pass did not run its parameter-gradient GEMMs")
        flat, flat_lp, grad, lp_ptr = self._buffers()
        (off, ln, n), trest = m.rest_tables(armed["covered"])
        if not armed["rest_done"]:                 # (round 5: normally the table launch's front tiles have done this — ops.ParamGradQueue._table_aux)
            L.check(L.load().mtn_adam_step_chunks(L.dtype_code(m.compute_dtype), n, off.data_ptr(), ln.data_ptr(), flat.data_ptr(),
                                                  grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), lp_ptr, self.state.data_ptr(),
                                                  L.ptr(self.grad_scale), self.betas[0], self.betas[1], self.eps, L.stream_ptr()))
        if trest is not None:
            m.refresh_transposed(trest)

    def zero_grad(self, set_to_none: bool = False):
        self.model.zero_glue_grads()

    def state_dict(self):
        """Optimiser state for checkpoints (the reference saves none: train.py:215-217 pickles the model only).  Moments are
        stored per parameter NAME, so a checkpoint survives a different flat layout; ``schedule`` is the device-side Noam /
        bias-correction state [step, lr, 1-b1^t, 1-b2^t]."""
        off = 0
        sh = getattr(self, "_sharded", None)
        if sh is not None:          # data parallel with a sharded optimiser: every rank holds the moments of its shards only
            sh.gather(self.m)
            sh.gather(self.v)
            self.gather_masters()
        out = {"schedule": self.state.detach().cpu().clone(), "exp_avg": {}, "exp_avg_sq": {}}
        names = {id(p): n for n, p in self.model.named_parameters()}
        for p, o in zip(self.model._flat_params, self.model._flat_offsets):
            n = names[id(p)]
            out["exp_avg"][n] = self.m[o:o + p.numel()].view(p.shape).detach().cpu().clone()
            out["exp_avg_sq"][n] = self.v[o:o + p.numel()].view(p.shape).detach().cpu().clone()
        return out

    def gather_masters(self):
        """Data parallel with the compute-dtype gather (dp.ShardedOptimizerSync): a rank keeps the fp32 master of its own shard of
        every weight matrix current and receives the others' bf16 copies only — bring every rank's masters up to date before
        anything reads them whole (model.state_dict()).  A collective: every rank calls it."""
        sh = getattr(self, "_sharded", None)
        if sh is not None and sh.lp_slices and sh.masters_stale:
            sh.gather(self.model._flat)
            sh.masters_stale = False

    def load_state_dict(self, sd):
        names = {id(p): n for n, p in self.model.named_parameters()}
        self.state.copy_(sd["schedule"].to(self.state.device))
        for p, o in zip(self.model._flat_params, self.model._flat_offsets):
            n = names[id(p)]
            self.m[o:o + p.numel()].copy_(sd["exp_avg"][n].reshape(-1).to(self.m.device))
            self.v[o:o + p.numel()].copy_(sd["exp_avg_sq"][n].reshape(-1).to(self.v.device))


class NoamOpt:
    """Optimiser wrapper with the reference's interface (data_utils.py:92-117): ``step()``, ``rate()``,
    ``.optimizer.zero_grad()``.  The schedule itself runs on device inside FusedAdam.tick()."""

    def __init__(self, model_size, factor, warmup, optimizer: FusedAdam):
        self.optimizer = optimizer
        self._step = 0
        self.warmup, self.factor, self.model_size = warmup, factor, model_size
        self._rate = 0

    def step(self):
        self._step += 1
        self.optimizer.tick(self.factor, self.model_size, self.warmup)
        self.optimizer.step()

    def begin_fused_step(self, write_grad: bool = False):
        """step() split around the backward pass (FusedAdam.fuse_into_backward): advance the schedule and arm the epilogue
        before it, finish_fused_step() after it.  Same arithmetic as step(), element for element."""
        self._step += 1
        self.optimizer.tick(self.factor, self.model_size, self.warmup)
        self.optimizer.fuse_into_backward(write_grad)

    def finish_fused_step(self):
        self.optimizer.step_rest()

    def begin_sharded_step(self):
        """Data parallel with a sharded optimiser (dp.ShardedOptimizerSync): advance the schedule now; the per-shard updates
        (FusedAdam.step_range) follow slice by slice, refresh_copies() at the end."""
        self._step += 1
        self.optimizer.tick(self.factor, self.model_size, self.warmup)

    def step_count(self) -> int:
        """Optimiser steps taken so far.  The schedule lives on the device (FusedAdam.state[0]) and advances on every replay
        of a captured step, which the host-side counter does not see: read it there (one small D2H copy; checkpoints / logs only)."""
        st = getattr(self.optimizer, "state", None)
        if torch.is_tensor(st) and st.numel() > 0:
            self._step = int(round(float(st[0].item())))
        return self._step

    def state_dict(self):
        return {"step": self.step_count(), "optimizer": self.optimizer.state_dict()}

    def load_state_dict(self, sd):
        self._step = int(sd["step"])
        self.optimizer.load_state_dict(sd["optimizer"])

    def rate(self, step=None):
        if step is None:
            step = self.step_count()
        return self.factor * (self.model_size ** (-0.5) * min(step ** (-0.5), step * self.warmup ** (-1.5)))


class SimpleLossCompute:
    """generator -> label-smoothed KL (+ lambda * auto-encoder losses) -> backward -> optimiser step
    (data_utils.py:123-156).  ``sync=False`` returns the device tensor instead of ``loss.item()*norm`` so that the
    step contains no host synchronisation; ``grad_sync`` (optional callable) runs between backward and the
    optimiser step — that is where data-parallel gradient all-reduce goes."""

    def __init__(self, generator, ae_generator, criterion, opt=None, l=1.0, sync=True, grad_sync=None, fused=True):
        self.generator, self.ae_generator, self.criterion = generator, ae_generator, criterion
        self.opt, self.l, self.sync, self.grad_sync, self.fused = opt, l, sync, grad_sync, fused

    def _fused_loss(self, x, y, norm, ae_x, ae_y, ae_norm):
        """Generator + log-softmax + label-smoothed KL + weighted sum as the fused HIP loss head (ops.GeneratorLossFn).
        Returns None when the fused path does not apply (CPU tensors, un-flattened model, vocab not a multiple of 8,
        foreign criterion): the caller then composes the same value from PyTorch ops."""
        from . import ops
        gens = [self.generator]
        xs, ys, norms, coefs = [x], [y], [norm], [1.0]
        if ae_x is not None:
            lst = ae_x if isinstance(ae_x, (list, tuple)) else [ae_x]
            for i, a in enumerate(lst):
                if self.ae_generator is not None:
                    gens.append(self.ae_generator[i] if isinstance(ae_x, (list, tuple)) else self.ae_generator)
                else:
                    gens.append(self.generator)
                xs.append(a); ys.append(ae_y); norms.append(ae_norm); coefs.append(self.l)
        if not (isinstance(self.criterion, LabelSmoothing) and x.is_cuda and len(xs) <= 4):
            return None
        fused = [getattr(g, "_fused", None) for g in gens]
        if any(f is None for f in fused) or self.criterion.size % 8 != 0:
            return None
        if not torch.is_grad_enabled():
            pass
        spec = dict(targets=ys, norms=[n if torch.is_tensor(n) else torch.tensor(float(n), device=x.device) for n in norms], coefs=coefs,
                    gens=[(f["w_lp"], f["bias"], f["grad_w"], f["grad_b"], f.get("w_lpT")) for f in fused], vocab=self.criterion.size,
                    pad=self.criterion.padding_idx, smoothing=self.criterion.smoothing, lp_dtype=fused[0]["lp_dtype"],
                    queue=fused[0].get("queue"))
        spec["norms"] = [n.detach().float().reshape(1) for n in spec["norms"]]
        return ops.GeneratorLossFn.apply(spec, *xs)

    def loss(self, x, y, norm, ae_x=None, ae_y=None, ae_norm=None):
        fused = self._fused_loss(x, y, norm, ae_x, ae_y, ae_norm) if self.fused else None
        if fused is not None:
            return fused
        out = self.generator(x)
        loss = self.criterion(out.reshape(-1, out.size(-1)), y.reshape(-1)) / norm.float()
        if ae_x is not None:
            xs = ae_x if isinstance(ae_x, (list, tuple)) else [ae_x]
            for i, ae_in in enumerate(xs):
                if self.ae_generator is not None:
                    gen = self.ae_generator[i] if isinstance(ae_x, (list, tuple)) else self.ae_generator
                else:
                    gen = self.generator
                ae_out = gen(ae_in)
                loss = loss + self.l * self.criterion(ae_out.reshape(-1, ae_out.size(-1)), ae_y.reshape(-1)) / ae_norm.float()
        return loss

    def __call__(self, x, y, norm, ae_x=None, ae_y=None, ae_norm=None):
        loss = self.loss(x, y, norm, ae_x, ae_y, ae_norm)
        if self.opt is not None:
            loss.backward()
            if self.grad_sync is not None:
                self.grad_sync()
            self.opt.step()
            self.opt.optimizer.zero_grad()
        out = loss.detach() * norm.float()
        return out.item() if self.sync else out
