"""Operators of the MTN hot path, backed by libmtn_hip.so.

Mirrors the reference's operator surface (mtn.py): ``LayerNorm.forward`` (:111-114),
``SublayerConnection.forward`` ∘ ``MultiHeadedAttention.forward`` (:125-127, :248-267, :221-231) and
``SublayerConnection.forward`` ∘ ``PositionwiseFeedForward.forward`` (:125-127, :279-280), each as one
``torch.autograd.Function`` whose forward/backward enqueue the fused HIP kernel chains.  PyTorch is used
for device memory, streams and autograd bookkeeping only.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import lib as L


# ------------------------------------------------------------------------------------------ helpers
_XACC = True          # a tensor with several consumers collects its gradients in one buffer (no autograd add)
_HANDOFF_MEM = True   # a memory's final gradient also leaves its last dmem GEMM as the producer's masked dy


def _drop(p: float, salt: int, seed: Optional[torch.Tensor]) -> L.Dropout:
    if p > 0.0 and seed is not None:
        return L.Dropout(float(p), int(salt) & 0xFFFFFFFF, seed.data_ptr())
    return L.Dropout(0.0, 0, None)


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.MtnHipError("mtn_amd operators run on the GPU only (HIP kernels); got a CPU tensor")


def _mask_u8(mask: Optional[torch.Tensor], B: int, a: int, m: int):
    """Reference masks are bool (B|1, 1|a, m) (data_utils.py:34-53).  -> (uint8 tensor, batch stride, row stride).
    The uint8 image is cached on the mask object: the same mask serves every layer (and several streams)."""
    if mask is None:
        return None, 0, 0
    if mask.dim() != 3 or mask.size(-1) != m or mask.size(0) not in (1, B) or mask.size(1) not in (1, a):
        raise ValueError(f"mask {tuple(mask.shape)} does not broadcast to ({B},{a},{m})")
    cached = getattr(mask, "_mtn_u8", None)
    if cached is None:
        cached = mask.to(torch.uint8).contiguous()
        try:
            mask._mtn_u8 = cached
        except Exception:
            pass
    sb = 0 if cached.size(0) == 1 else cached.size(1) * m
    sq = 0 if cached.size(1) == 1 else m
    return cached, sb, sq


def prepare_masks(*masks):
    """Convert masks to their kernel form NOW (on the current stream), before other streams start using them."""
    for mk in masks:
        if isinstance(mk, (list, tuple)):
            prepare_masks(*mk)
        elif mk is not None and getattr(mk, "_mtn_u8", None) is None:
            mk._mtn_u8 = mk.to(torch.uint8).contiguous()


def gemm(dtype: int, problems):
    """problems: list of lib.GemmProblem -> one grouped launch."""
    arr = (L.GemmProblem * len(problems))(*problems)
    L.check(L.load().mtn_gemm(dtype, len(problems), arr, L.stream_ptr()))


def cast_to_lp(x: torch.Tensor, lp_dtype: torch.dtype) -> torch.Tensor:
    _require_cuda(x)
    if lp_dtype == torch.float32:
        return x.contiguous()
    x = x.contiguous()
    out = torch.empty_like(x, dtype=lp_dtype)
    L.check(L.load().mtn_cast_f32_to_lp(L.dtype_code(lp_dtype), x.numel(), x.data_ptr(), out.data_ptr(), L.stream_ptr()))
    return out


# ------------------------------------------------------------------------------------------ deferred parameter gradients
class ParamGradQueue:
    """dW/db GEMMs and LayerNorm gain/bias reductions are off the critical path of backward.  Sublayer backwards
    enqueue them here (problem structs built by the C side, nothing launched) and the queue launches them in a few
    large grouped kernels when the backward pass ends (autograd engine callback) — hundreds of workgroups per launch
    instead of ~250 small launches serialised behind the activation-gradient chain."""

    def __init__(self):
        self.gemm, self.ln, self.keep = [], [], []
        self.embed = []             # deferred embedding-table scatters (EmbedBwdDesc): every table of the step in ONE grouped launch
        self.dtype = None
        self._armed = False
        self.adam = None            # set by FusedAdam.fuse_into(): the next flush applies the optimiser in the GEMM epilogues

    def _attach_adam(self):
        """Optimiser epilogue (include/mtn_hip.h mtn_adam_fuse): every dW problem whose target is (a row block of) a fusable
        sublayer weight updates that weight, its moments and both compute-dtype copies in place instead of storing a gradient.
        Raises unless the fusable weights are covered exactly once — a silent partial update would corrupt training."""
        import bisect
        A = self.adam
        table, starts = A["fusable"], A["starts"]
        esz = A["esz"]
        keep = []
        got = {}                                     # offset of the fusable weight -> elements that received a gradient GEMM
        seen = set()
        for p in self.gemm:
            if not (p.a_trans and p.b_trans and p.out_f32):
                continue
            off = (p.out_f32 - A["grad"]) // 4
            i = bisect.bisect_right(starts, off) - 1
            if i < 0:
                continue
            o, rows, cols = table[i]
            if not (o <= off < o + rows * cols):
                continue
            rel = off - o
            ok = (not p.residual and p.ldc == cols and p.N == cols and rel % cols == 0 and rel // cols + p.M <= rows and off not in seen)
            if not ok:
                raise L.MtnHipError("optimiser epilogue: a parameter-gradient GEMM does not cover its weight block once and whole")
            seen.add(off)
            r0 = rel // cols
            f = L.AdamFuse()
            f.p, f.m, f.v = A["p"] + 4 * off, A["m"] + 4 * off, A["v"] + 4 * off
            f.p_lp = A["lp"] + esz * off if A["lp"] else None
            f.p_lpT = A["lpT"] + esz * (A["t_map"][o] + r0) if (A["lpT"] and o in A["t_map"]) else None     # (compact buffer of the kept copies)
            f.ldT = rows
            f.write_grad = 1 if A["write_grad"] else 0
            f.state, f.grad_scale = A["state"], A["grad_scale"]
            f.beta1, f.beta2, f.eps = A["betas"][0], A["betas"][1], A["eps"]
            p.adam = C.pointer(f)
            keep.append(f)
            got[o] = got.get(o, 0) + p.M * p.N
        covered = set()
        for o, rows, cols in table:
            n = got.get(o, 0)
            if n == rows * cols:
                covered.add(o)
            elif n != 0 or o not in A["optional"]:
                raise L.MtnHipError(f"optimiser epilogue: weight at offset {o} received gradient GEMMs for {n} of {rows * cols} elements")
        A["covered"] = frozenset(covered)
        A["applied"] = True
        return keep

    def _table_aux(self, tt):
        """The rest of the optimiser step as FRONT tiles of the table launch (include/mtn_hip.h mtn_tt_aux), when the optimiser
        epilogue is armed: LayerNorm gains / biases (finalize + Adam per 256 columns), the bias of every Linear whose weight
        gradient is one of the launch's problems (updated by the tile that sums it), and every other range of the flat buffers no dW
        epilogue covers — the embedding tables, whose gradient the scatter launch above has completed — as <= 4096-element chunks.
        Nothing is left behind the launch: FusedAdam.step_rest() then only refreshes transposed copies.  Returns (aux, keep-alive)
        or None (a gradient that does not live in the model's flat buffer)."""
        A = self.adam
        if A is None or not A.get("n_flat"):
            return None
        g0, n_flat = A["grad"], A["n_flat"]
        inside = lambda ptr, n: ptr is not None and g0 <= ptr and ptr + 4 * n <= g0 + 4 * n_flat and (ptr - g0) % 4 == 0
        ln_units, taken = [], []                    # taken: (offset, length) ranges updated inside the launch
        for d_ in self.ln:
            if not (inside(d_.da2, d_.d) and inside(d_.db2, d_.d)):
                return None
            a_off, b_off = (d_.da2 - g0) // 4, (d_.db2 - g0) // 4
            ln_units.append((d_.partial, d_.nparts, d_.d, a_off, b_off))
            taken += [(a_off, d_.d), (b_off, d_.d)]
        for p in tt:
            if p.rowsum_out and inside(p.rowsum_out, p.M):
                taken.append(((p.rowsum_out - g0) // 4, p.M))
            if not p.adam and inside(p.out_f32, 1):
                return None                         # a weight gradient this very launch STORES (no epilogue): not complete when the front tiles run
        for o, r, c in A["fusable"]:
            if o in A["covered"]:
                taken.append((o, r * c))
        taken.sort()
        if any(taken[i][0] + taken[i][1] > taken[i + 1][0] for i in range(len(taken) - 1)):
            return None                             # a gradient with two producers (shared module): keep the separate passes
        key = tuple(taken)
        hit = A["aux_cache"].get(key)
        if hit is None:
            offs, lens, cur = [], [], 0
            for o, n in taken + [(n_flat, 0)]:
                lo, hi = (cur + 3) // 4 * 4, o // 4 * 4            # (every parameter starts on a multiple of 8 elements; paddings are inert)
                while lo < hi:
                    k = min(4096, hi - lo)
                    offs.append(lo); lens.append(k); lo += k
                cur = o + n
            hit = ((C.c_long * max(1, len(offs)))(*offs), (C.c_int * max(1, len(lens)))(*lens), len(offs))
            A["aux_cache"][key] = hit
        ln_arr = (L.TtLnUnit * max(1, len(ln_units)))()
        for i, (partial, nparts, d, a_off, b_off) in enumerate(ln_units):
            ln_arr[i].partial, ln_arr[i].nparts, ln_arr[i].d, ln_arr[i].a_off, ln_arr[i].b_off = partial, nparts, d, a_off, b_off
        aux = L.TtAux()
        aux.p, aux.g, aux.m, aux.v, aux.lp, aux.n_flat = A["p"], g0, A["m"], A["v"], A["lp"], n_flat
        aux.n_ln, aux.ln = len(ln_units), ln_arr
        aux.n_chunks, aux.chunk_off, aux.chunk_len = hit[2], hit[0], hit[1]
        aux.bias_adam = 1
        aux.state, aux.grad_scale = A["state"], A["grad_scale"]
        aux.beta1, aux.beta2, aux.eps = A["betas"][0], A["betas"][1], A["eps"]
        return aux, (ln_arr, hit)

    def add(self, dtype, problems, ln_desc, keep):
        assert self.dtype in (None, dtype)
        self.dtype = dtype
        self.gemm.extend(problems)
        if ln_desc is not None:
            self.ln.append(ln_desc)
        self.keep.extend(keep)
        if not self._armed:
            self._armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)

    def add_embed(self, descs, keep):
        """Defer embedding-table gradient scatters (mtn_embed_bwd_group) to the flush: the text streams' table and the target
        table are separate autograd nodes, but one grouped launch serves both (one workgroup grid per table)."""
        self.embed.extend(descs)
        self.keep.extend(keep)
        if not self._armed:
            self._armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)

    def reset(self):
        """Drop everything queued and disarm.  A backward pass that raises never reaches the engine's final callbacks, so the
        queue would stay armed with stale problems and no later backward would register a flush: the step's error path and
        the start of every step call this."""
        self.gemm, self.ln, self.keep, self.embed = [], [], [], []
        self._armed = False
        self.adam = None

    def flush(self):
        self._armed = False
        if not self.gemm and not self.ln and not self.embed:
            return
        lib = L.load()
        cur = torch.cuda.current_stream()
        for i in range(0, len(self.embed), 8):            # (MTN_LN_MAX_GROUP descriptors per launch)
            chunk = self.embed[i:i + 8]
            arr = (L.EmbedBwdDesc * len(chunk))(*chunk)
            L.check(lib.mtn_embed_bwd_group(len(chunk), arr, cur.cuda_stream))
        self.embed = []
        if not self.gemm and not self.ln:
            self.keep = []
            return
        for t in self.keep:
            t.record_stream(cur)
        adam_keep = self._attach_adam() if self.adam is not None else None      # noqa: F841 (descriptors live until the launches)
        # a grouped launch lasts as long as its longest contraction: keep the long ones (memories: K = B*m rows) together
        self.gemm.sort(key=lambda p: -p.K)
        if self.dtype == L.MTN_BF16:
            # ALL parameter-gradient problems in one launch of 128x128 tiles (csrc/gemm.hip, table form), long and short
            # contractions side by side: no launch boundaries or partial rounds (296 vs 430 us at cfg2), and with the optimiser
            # epilogue the tiles' streaming phases overlap other tiles' contractions.  Without the epilogue only when the
            # problems fill 128-wide tiles (tiny test models keep the 64-wide kernels).
            tt = [p for p in self.gemm if p.a_trans and p.b_trans and p.M % 8 == 0 and p.N % 8 == 0 and not p.residual]
            big = all(p.M >= 128 and p.N >= 128 for p in tt) and sum(((p.M + 127) // 128) * ((p.N + 127) // 128) for p in tt) >= 192
            if tt and len(tt) <= 1024 and (self.adam is not None or big):
                order, lo, hi = [], 0, len(tt) - 1
                while lo <= hi:
                    order.append(tt[lo]); lo += 1
                    if lo <= hi:
                        order.append(tt[hi]); hi -= 1
                arr = (L.GemmProblem * len(order))(*order)
                rest = [p for p in self.gemm if id(p) not in set(id(q) for q in tt)]
                aux = self._table_aux(tt) if (self.adam is not None and not rest) else None
                if aux is not None:
                    L.check(lib.mtn_gemm_tt_table_aux(self.dtype, len(order), arr, C.byref(aux[0]), cur.cuda_stream))
                    self.adam["rest_done"] = True          # LayerNorm finalize + every remaining Adam update ran as front tiles
                    self.ln = []
                else:
                    L.check(lib.mtn_gemm_tt_table(self.dtype, len(order), arr, cur.cuda_stream))
                self.gemm = rest
        # ... and at most 512 of the 128x128 tiles (two resident workgroups per CU x 256 CUs) per launch: a launch that
        # spills into a second round pays for a whole extra round
        chunk, tiles = [], 0
        for p in self.gemm + [None]:
            t = 0 if p is None else ((p.M + 127) // 128) * ((p.N + 127) // 128)
            if chunk and (p is None or len(chunk) == L.GEMM_MAX_GROUP or tiles + t > 512):
                arr = (L.GemmProblem * len(chunk))(*chunk)
                L.check(lib.mtn_gemm(self.dtype, len(chunk), arr, cur.cuda_stream))
                chunk, tiles = [], 0
            if p is not None:
                chunk.append(p); tiles += t
        if self.ln:
            arr = (L.LnFinalizeDesc * len(self.ln))(*self.ln)
            L.check(lib.mtn_layernorm_bwd_finalize(len(self.ln), arr, cur.cuda_stream))
        self.gemm, self.ln, self.keep = [], [], []


# ------------------------------------------------------------------------------------------ LayerNorm
class LayerNormFn(torch.autograd.Function):
    """mtn.py:111-114.  Returns (y_f32, y_lp); y_lp (compute dtype copy for GEMM operands) is non-differentiable."""

    @staticmethod
    def forward(ctx, x, a2, b2, eps, lp_dtype, grad_a, grad_b, queue=None):
        _require_cuda(x, a2, b2)
        x = x.contiguous()
        d = x.size(-1)
        rows = x.numel() // d
        y = torch.empty_like(x)
        want_lp = lp_dtype is not None and lp_dtype != torch.float32
        y_lp = torch.empty_like(x, dtype=lp_dtype) if want_lp else None
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        code = L.dtype_code(lp_dtype) if want_lp else L.MTN_F32
        L.check(L.load().mtn_layernorm_fwd(code, rows, d, eps, x.data_ptr(), a2.data_ptr(), b2.data_ptr(), y.data_ptr(),
                                           L.ptr(y_lp), mean.data_ptr(), rstd.data_ptr(), L.stream_ptr()))
        ctx.save_for_backward(x, a2, mean, rstd)
        ctx.eps, ctx.grad_a, ctx.grad_b, ctx.queue = eps, grad_a, grad_b, queue
        if y_lp is None:
            y_lp = torch.empty(0, device=x.device, dtype=torch.float32)   # placeholder second output
        ctx.mark_non_differentiable(y_lp)
        return y, y_lp

    @staticmethod
    def backward(ctx, g, _g_lp):
        x, a2, mean, rstd = ctx.saved_tensors
        d = x.size(-1)
        rows = x.numel() // d
        g = g.contiguous()
        lib = L.load()
        dx = torch.empty_like(x)
        da = ctx.grad_a if ctx.grad_a is not None else torch.empty_like(a2)
        db = ctx.grad_b if ctx.grad_b is not None else torch.empty_like(a2)
        partial = torch.empty(lib.mtn_layernorm_bwd_partial_floats(rows, d), device=x.device, dtype=torch.float32)
        defer = ctx.queue is not None and ctx.grad_a is not None
        L.check(lib.mtn_layernorm_bwd(rows, d, ctx.eps, x.data_ptr(), a2.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                      g.data_ptr(), None, dx.data_ptr(), None if defer else da.data_ptr(),
                                      None if defer else db.data_ptr(), partial.data_ptr(), L.stream_ptr()))
        if defer:
            desc = L.LnFinalizeDesc(partial.data_ptr(), lib.mtn_layernorm_bwd_nparts(rows), d, da.data_ptr(), db.data_ptr())
            ctx.queue.add(None if ctx.queue.dtype is None else ctx.queue.dtype, [], desc, [partial])
        ret_a = None if ctx.grad_a is not None else da
        ret_b = None if ctx.grad_b is not None else db
        return dx, ret_a, ret_b, None, None, None, None, None


class LayerNormGroupFn(torch.autograd.Function):
    """Several independent LayerNorms (mtn.py:111-114; the decoder's final norms, mtn.py:164) in ONE launch each way.
    spec: norms = [(a2, b2, eps, grad_a, grad_b)] (grad_* = fp32 destinations or None), lp_dtype, queue.  With a compute dtype the
    outputs also go, in order, to consecutive row ranges of one buffer (spec['_lp_rows'], [sum rows, d]) — the operand of the
    loss head's GEMM, which then needs no cast launch.  Parameter gradients are written to grad_* (deferred through the queue)
    or, without destinations, returned through autograd as usual."""

    @staticmethod
    def forward(ctx, spec, *tensors):
        lib = L.load()
        n = len(spec["norms"])
        xs = [t.contiguous() for t in tensors[:n]]
        _require_cuda(*xs)
        dev = xs[0].device
        lp = spec["lp_dtype"]
        want_lp = lp is not None and lp != torch.float32
        code = L.dtype_code(lp) if want_lp else L.MTN_F32
        d = xs[0].size(-1)
        rows = [x.numel() // d for x in xs]
        y_lp = torch.empty(sum(rows), d, device=dev, dtype=lp) if want_lp else None
        descs = (L.LnFwdDesc * n)()
        ys, saved, off = [], [], 0
        for i, (x, (a2, b2, eps, _ga, _gb)) in enumerate(zip(xs, spec["norms"])):
            y = torch.empty_like(x)
            mean = torch.empty(rows[i], device=dev, dtype=torch.float32)
            rstd = torch.empty_like(mean)
            D = descs[i]
            D.rows, D.d, D.eps, D.x, D.a2, D.b2 = rows[i], d, eps, x.data_ptr(), a2.data_ptr(), b2.data_ptr()
            D.y_f32, D.y_lp, D.mean, D.rstd = y.data_ptr(), (y_lp[off:].data_ptr() if want_lp else None), mean.data_ptr(), rstd.data_ptr()
            ys.append(y)
            saved.append((x, mean, rstd))
            off += rows[i]
        L.check(lib.mtn_layernorm_fwd_group(code, n, descs, L.stream_ptr()))
        spec["_lp_rows"] = y_lp
        ctx.spec, ctx.saved, ctx.rows, ctx.d = spec, saved, rows, d
        # gradient hand-off (GroupMember.holder): the sublayer that produced an input wants dx once more, through ITS output dropout,
        # in the compute dtype
        ctx.feeds = [getattr(t, "_mtn_next", None) for t in tensors[:n]]
        # an input that a sublayer attends as memory too (the last layer's auto-encoder outputs): the two gradients meet in one
        # buffer (protocol of SublayerGroupFn: `_mtn_gacc`) instead of an autograd add
        ctx.xaccs = []
        for t in tensors[:n]:
            xg = getattr(t, "_mtn_gacc", None) if (_XACC and t.requires_grad) else None
            if xg is not None:
                xg["remaining"] += 1
            ctx.xaccs.append(xg)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gs):
        lib = L.load()
        spec, saved, rows, d = ctx.spec, ctx.saved, ctx.rows, ctx.d
        n = len(saved)
        queue = spec.get("queue")
        descs = (L.LnBwdDesc * n)()
        dxs, finals, keep, rets = [], [], [], []
        for i, ((x, mean, rstd), (a2, _b2, eps, ga, gb)) in enumerate(zip(saved, spec["norms"])):
            g = gs[i].contiguous() if gs[i] is not None else torch.zeros_like(x)
            dx = torch.empty_like(x)
            partial = torch.empty(lib.mtn_layernorm_bwd_partial_floats(rows[i], d), device=x.device, dtype=torch.float32)
            B_ = descs[i]
            B_.rows, B_.d, B_.eps, B_.x, B_.a2, B_.mean, B_.rstd = rows[i], d, eps, x.data_ptr(), a2.data_ptr(), mean.data_ptr(), rstd.data_ptr()
            B_.g, B_.dx, B_.partial = g.data_ptr(), dx.data_ptr(), partial.data_ptr()
            f = ctx.feeds[i]
            if f is not None and f.get("lp") is not None and f["lp"] != torch.float32 and ctx.xaccs[i] is None:   # (a shared gradient buffer is not final here)
                nxt = torch.empty(x.shape, device=x.device, dtype=f["lp"])
                f["dyl"], f["dx_ptr"], f["ver"], f["shape"] = nxt, dx.data_ptr(), dx._version, dx.shape
                B_.dx_lp, B_.dx_lp_dtype, B_.dx_lp_drop = nxt.data_ptr(), L.dtype_code(f["lp"]), _drop(f["p"], f["salt"], f["seed"])
                keep.append(nxt)
            da = ga if ga is not None else torch.empty_like(a2)
            db = gb if gb is not None else torch.empty_like(a2)
            finals.append((L.LnFinalizeDesc(partial.data_ptr(), lib.mtn_layernorm_bwd_nparts(rows[i]), d, da.data_ptr(), db.data_ptr()), partial, ga is not None))
            rets.append((None if ga is not None else da, None if gb is not None else db))
            dxs.append(dx)
            keep += [g, partial]
        L.check(lib.mtn_layernorm_bwd_group(n, descs, L.stream_ptr()))
        for i, xg in enumerate(ctx.xaccs):
            if xg is None:
                continue
            xg["remaining"] -= 1
            if xg["buf"] is None and xg["remaining"] > 0:
                xg["buf"], dxs[i] = dxs[i], None                  # first writer: the memory-role users accumulate into our dx
            elif xg["buf"] is not None:
                buf = xg["buf"]
                buf.add_(dxs[i])
                dxs[i] = None
                if xg["remaining"] == 0:
                    dxs[i], xg["buf"] = buf, None
        now = []
        for desc, partial, has_dest in finals:
            if queue is not None and has_dest:
                queue.add(queue.dtype, [], desc, [partial])
            else:
                now.append(desc)
        if now:
            L.check(lib.mtn_layernorm_bwd_finalize(len(now), (L.LnFinalizeDesc * len(now))(*now), L.stream_ptr()))
        ctx.keep = keep
        return (None,) + tuple(dxs) + tuple(r[0] for r in rets) + tuple(r[1] for r in rets)


def layer_norm_group(xs, modules):
    """-> [y_i fp32] for independent LayerNorm modules (mtn_amd.mtn.LayerNorm) in one launch; each output carries
    `_mtn_lp_rows = (buffer, first row, rows)`: its compute-dtype copy inside one [sum rows, d] buffer (None in fp32 mode)."""
    spec = dict(norms=[(m.a_2, m.b_2, m.eps) + (tuple(m._grads) if m._grads is not None else (None, None)) for m in modules],
                lp_dtype=modules[0]._lp_dtype, queue=modules[0]._queue)
    params = [m.a_2 for m in modules] + [m.b_2 for m in modules]        # autograd inputs (gradients returned when no destination is set)
    ys = LayerNormGroupFn.apply(spec, *xs, *params)
    buf, off = spec.get("_lp_rows"), 0
    for y in ys:
        r = y.numel() // y.size(-1)
        if buf is not None:
            y._mtn_lp_rows = (buf, off, r)
        off += r
    return list(ys)


def layer_norm(x, a2, b2, eps=1e-6, lp_dtype=None, grad_a=None, grad_b=None, queue=None):
    """-> (y fp32, y in the compute dtype).  grad_a/grad_b: optional fp32 destinations for da2/db2."""
    y, y_lp = LayerNormFn.apply(x, a2, b2, eps, lp_dtype, grad_a, grad_b, queue)
    if y_lp.numel() == 0 and y.numel() != 0:
        y_lp = y.detach()
    return y, y_lp


# ------------------------------------------------------------------------------------------ sublayers
@dataclass
class MhaConfig:
    heads: int
    eps: float = 1e-6
    p_attn: float = 0.0           # dropout on softmax probabilities (mtn.py:230)
    p_out: float = 0.0            # dropout on the sublayer output (mtn.py:127)
    salt: int = 0                 # unique per sublayer: sites salt*4+{0,1}
    seed: Optional[torch.Tensor] = None          # device int64[1], advanced once per step
    lp_dtype: torch.dtype = torch.bfloat16
    w_qkv_lp: Optional[torch.Tensor] = None      # compute-dtype copies of the weights ([3d,d], [d,d])
    w_o_lp: Optional[torch.Tensor] = None
    w_qkv_lpT: Optional[torch.Tensor] = None     # transposed compute-dtype copies (backward dX on the LDS-DMA GEMM path)
    w_o_lpT: Optional[torch.Tensor] = None
    grads: Optional[dict] = None                 # fp32 destinations: ln_a ln_b w_qkv b_qkv w_o b_o (flat-grad views)
    queue: Optional[ParamGradQueue] = None       # defer dW/db/LN-parameter work to the end of backward
    ln_fold: Optional[torch.Tensor] = None       # fold vectors of w_qkv behind the sublayer's LayerNorm (mtn_ln_fold), float [2K]


class MHASublayerFn(torch.autograd.Function):
    """y = x + dropout(MHA(LN(x), mem, mem, mask)).  mem=None -> self-attention (key=value=LN(x), mtn.py:183,209)."""

    @staticmethod
    def forward(ctx, x, mem, mem_lp, mask, ln_a, ln_b, w_qkv, b_qkv, w_o, b_o, cfg: MhaConfig):
        _require_cuda(x, mem, ln_a, w_qkv)
        lib = L.load()
        x = x.contiguous()
        B, a, d = x.shape
        self_attn = mem is None
        m = a if self_attn else mem.size(1)
        lp = cfg.lp_dtype
        code = L.dtype_code(lp)
        dev = x.device
        if not self_attn and mem_lp is None:
            mem_lp = cast_to_lp(mem, lp)
        w_qkv_lp = cfg.w_qkv_lp if cfg.w_qkv_lp is not None else cast_to_lp(w_qkv, lp)
        w_o_lp = cfg.w_o_lp if cfg.w_o_lp is not None else cast_to_lp(w_o, lp)
        mask_u8, sb, sq = _mask_u8(mask, B, a, m)
        y = torch.empty_like(x)
        xn = torch.empty(B * a, d, device=dev, dtype=lp)
        mean = torch.empty(B * a, device=dev, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        qkv = torch.empty(B * a, 3 * d if self_attn else d, device=dev, dtype=lp)
        kv = None if self_attn else torch.empty(B * m, 2 * d, device=dev, dtype=lp)
        o = torch.empty(B * a, d, device=dev, dtype=lp)
        lse = torch.empty(2 * B * cfg.heads * a, device=dev, dtype=torch.float32)
        args = L.MhaArgs()
        args.B, args.a, args.m, args.d, args.h = B, a, m, d, cfg.heads
        args.self_attn, args.ln_eps = int(self_attn), cfg.eps
        args.drop_attn = _drop(cfg.p_attn, cfg.salt * 4 + 0, cfg.seed)
        args.drop_out = _drop(cfg.p_out, cfg.salt * 4 + 1, cfg.seed)
        args.x, args.mem = x.data_ptr(), L.ptr(mem_lp)
        args.mask, args.mask_sb, args.mask_sq = L.ptr(mask_u8), sb, sq
        args.ln_a, args.ln_b = ln_a.data_ptr(), ln_b.data_ptr()
        args.w_qkv, args.b_qkv, args.w_o, args.b_o = w_qkv_lp.data_ptr(), b_qkv.data_ptr(), w_o_lp.data_ptr(), b_o.data_ptr()
        args.y, args.xn, args.mean, args.rstd = y.data_ptr(), xn.data_ptr(), mean.data_ptr(), rstd.data_ptr()
        args.qkv, args.kv, args.o, args.lse = qkv.data_ptr(), L.ptr(kv), o.data_ptr(), lse.data_ptr()
        L.check(lib.mtn_mha_sublayer_fwd(code, C.byref(args), L.stream_ptr()))
        ctx.save_for_backward(x, mem_lp, mask_u8, ln_a, ln_b, w_qkv_lp, b_qkv, w_o_lp, b_o, xn, mean, rstd, qkv, kv, o, lse)
        ctx.cfg, ctx.dims, ctx.mask_strides = cfg, (B, a, m, d, self_attn), (sb, sq)
        ctx.need_dmem = (not self_attn) and mem.requires_grad
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mem_lp, mask_u8, ln_a, ln_b, w_qkv_lp, b_qkv, w_o_lp, b_o, xn, mean, rstd, qkv, kv, o, lse = ctx.saved_tensors
        cfg: MhaConfig = ctx.cfg
        B, a, m, d, self_attn = ctx.dims
        lib = L.load()
        code = L.dtype_code(cfg.lp_dtype)
        dev = x.device
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dmem = torch.empty(B, m, d, device=dev, dtype=torch.float32) if ctx.need_dmem else None
        g = cfg.grads

        def dst(name, like):
            return g[name] if g is not None else torch.empty(like.shape, device=dev, dtype=torch.float32)

        d_ln_a, d_ln_b = dst("ln_a", ln_a), dst("ln_b", ln_b)
        d_w_qkv, d_b_qkv = dst("w_qkv", w_qkv_lp), dst("b_qkv", b_qkv)
        d_w_o, d_b_o = dst("w_o", w_o_lp), dst("b_o", b_o)
        ws_lp = torch.empty(lib.mtn_mha_bwd_ws_lp_elems(B, a, m, d, int(self_attn)), device=dev, dtype=cfg.lp_dtype)
        ws_f32 = torch.empty(lib.mtn_mha_bwd_ws_f32_floats(B, a, m, d), device=dev, dtype=torch.float32)
        args = L.MhaArgs()
        args.B, args.a, args.m, args.d, args.h = B, a, m, d, cfg.heads
        args.self_attn, args.ln_eps = int(self_attn), cfg.eps
        args.drop_attn = _drop(cfg.p_attn, cfg.salt * 4 + 0, cfg.seed)
        args.drop_out = _drop(cfg.p_out, cfg.salt * 4 + 1, cfg.seed)
        args.x, args.mem = x.data_ptr(), L.ptr(mem_lp)
        args.mask, args.mask_sb, args.mask_sq = L.ptr(mask_u8), ctx.mask_strides[0], ctx.mask_strides[1]
        args.ln_a, args.ln_b = ln_a.data_ptr(), ln_b.data_ptr()
        args.w_qkv, args.b_qkv, args.w_o, args.b_o = w_qkv_lp.data_ptr(), b_qkv.data_ptr(), w_o_lp.data_ptr(), b_o.data_ptr()
        args.w_qkv_t, args.w_o_t = L.ptr(cfg.w_qkv_lpT), L.ptr(cfg.w_o_lpT)
        args.xn, args.mean, args.rstd = xn.data_ptr(), mean.data_ptr(), rstd.data_ptr()
        args.qkv, args.kv, args.o, args.lse = qkv.data_ptr(), L.ptr(kv), o.data_ptr(), lse.data_ptr()
        args.dy, args.dx, args.dmem, args.dmem_accumulate = dy.data_ptr(), dx.data_ptr(), L.ptr(dmem), 0
        args.d_ln_a, args.d_ln_b = d_ln_a.data_ptr(), d_ln_b.data_ptr()
        args.d_w_qkv, args.d_b_qkv, args.d_w_o, args.d_b_o = d_w_qkv.data_ptr(), d_b_qkv.data_ptr(), d_w_o.data_ptr(), d_b_o.data_ptr()
        args.ws_lp, args.ws_f32 = ws_lp.data_ptr(), ws_f32.data_ptr()
        args.ln_fold = L.ptr(cfg.ln_fold)
        defer = g is not None and cfg.queue is not None
        args.defer_param_grads = int(defer)
        L.check(lib.mtn_mha_sublayer_bwd(code, C.byref(args), L.stream_ptr()))
        if defer:
            probs = (L.GemmProblem * 3)()
            ln = L.LnFinalizeDesc()
            n = lib.mtn_mha_param_grad_work(code, C.byref(args), probs, C.byref(ln))
            cfg.queue.add(code, [probs[i] for i in range(n)], ln, [ws_lp, ws_f32, o, xn, mem_lp] if mem_lp is not None else [ws_lp, ws_f32, o, xn])
        if g is not None:
            return dx, dmem, None, None, None, None, None, None, None, None, None
        return dx, dmem, None, None, d_ln_a, d_ln_b, d_w_qkv, d_b_qkv, d_w_o, d_b_o, None


@dataclass
class FfnConfig:
    eps: float = 1e-6
    p_hidden: float = 0.0         # mtn.py:280
    p_out: float = 0.0            # mtn.py:127
    salt: int = 0
    seed: Optional[torch.Tensor] = None
    lp_dtype: torch.dtype = torch.bfloat16
    w1_lp: Optional[torch.Tensor] = None
    w2_lp: Optional[torch.Tensor] = None
    w1_lpT: Optional[torch.Tensor] = None
    w2_lpT: Optional[torch.Tensor] = None
    grads: Optional[dict] = None  # ln_a ln_b w1 b1 w2 b2
    queue: Optional[ParamGradQueue] = None
    ln_fold: Optional[torch.Tensor] = None        # fold vectors of w1 (mtn_ln_fold), float [2 d_ff]


class FFNSublayerFn(torch.autograd.Function):
    """y = x + dropout(w_2(dropout(relu(w_1 LN(x)))))."""

    @staticmethod
    def forward(ctx, x, ln_a, ln_b, w1, b1, w2, b2, cfg: FfnConfig):
        _require_cuda(x, ln_a, w1)
        lib = L.load()
        x = x.contiguous()
        d = x.size(-1)
        rows = x.numel() // d
        ff = w1.size(0)
        lp = cfg.lp_dtype
        code = L.dtype_code(lp)
        dev = x.device
        w1_lp = cfg.w1_lp if cfg.w1_lp is not None else cast_to_lp(w1, lp)
        w2_lp = cfg.w2_lp if cfg.w2_lp is not None else cast_to_lp(w2, lp)
        y = torch.empty_like(x)
        xn = torch.empty(rows, d, device=dev, dtype=lp)
        mean = torch.empty(rows, device=dev, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        hid = torch.empty(rows, ff, device=dev, dtype=lp)
        args = L.FfnArgs()
        args.rows, args.d, args.d_ff, args.ln_eps = rows, d, ff, cfg.eps
        args.drop_hidden = _drop(cfg.p_hidden, cfg.salt * 4 + 2, cfg.seed)
        args.drop_out = _drop(cfg.p_out, cfg.salt * 4 + 1, cfg.seed)
        args.x, args.ln_a, args.ln_b = x.data_ptr(), ln_a.data_ptr(), ln_b.data_ptr()
        args.w1, args.b1, args.w2, args.b2 = w1_lp.data_ptr(), b1.data_ptr(), w2_lp.data_ptr(), b2.data_ptr()
        args.y, args.xn, args.mean, args.rstd, args.hid = y.data_ptr(), xn.data_ptr(), mean.data_ptr(), rstd.data_ptr(), hid.data_ptr()
        L.check(lib.mtn_ffn_sublayer_fwd(code, C.byref(args), L.stream_ptr()))
        ctx.save_for_backward(x, ln_a, ln_b, w1_lp, b1, w2_lp, b2, xn, mean, rstd, hid)
        ctx.cfg = cfg
        return y

    @staticmethod
    def backward(ctx, dy):
        x, ln_a, ln_b, w1_lp, b1, w2_lp, b2, xn, mean, rstd, hid = ctx.saved_tensors
        cfg: FfnConfig = ctx.cfg
        lib = L.load()
        code = L.dtype_code(cfg.lp_dtype)
        dev = x.device
        d = x.size(-1)
        rows = x.numel() // d
        ff = w1_lp.size(0)
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        g = cfg.grads

        def dst(name, like):
            return g[name] if g is not None else torch.empty(like.shape, device=dev, dtype=torch.float32)

        d_ln_a, d_ln_b = dst("ln_a", ln_a), dst("ln_b", ln_b)
        d_w1, d_b1, d_w2, d_b2 = dst("w1", w1_lp), dst("b1", b1), dst("w2", w2_lp), dst("b2", b2)
        ws_lp = torch.empty(rows * d + rows * ff, device=dev, dtype=cfg.lp_dtype)
        ws_f32 = torch.empty(lib.mtn_ffn_bwd_ws_f32_floats(rows, d, ff), device=dev, dtype=torch.float32)
        args = L.FfnArgs()
        args.rows, args.d, args.d_ff, args.ln_eps = rows, d, ff, cfg.eps
        args.drop_hidden = _drop(cfg.p_hidden, cfg.salt * 4 + 2, cfg.seed)
        args.drop_out = _drop(cfg.p_out, cfg.salt * 4 + 1, cfg.seed)
        args.x, args.ln_a, args.ln_b = x.data_ptr(), ln_a.data_ptr(), ln_b.data_ptr()
        args.w1, args.b1, args.w2, args.b2 = w1_lp.data_ptr(), b1.data_ptr(), w2_lp.data_ptr(), b2.data_ptr()
        args.w1_t, args.w2_t = L.ptr(cfg.w1_lpT), L.ptr(cfg.w2_lpT)
        args.xn, args.mean, args.rstd, args.hid = xn.data_ptr(), mean.data_ptr(), rstd.data_ptr(), hid.data_ptr()
        args.dy, args.dx = dy.data_ptr(), dx.data_ptr()
        args.d_ln_a, args.d_ln_b = d_ln_a.data_ptr(), d_ln_b.data_ptr()
        args.d_w1, args.d_b1, args.d_w2, args.d_b2 = d_w1.data_ptr(), d_b1.data_ptr(), d_w2.data_ptr(), d_b2.data_ptr()
        args.ws_lp, args.ws_f32 = ws_lp.data_ptr(), ws_f32.data_ptr()
        args.ln_fold = L.ptr(cfg.ln_fold)
        defer = g is not None and cfg.queue is not None
        args.defer_param_grads = int(defer)
        L.check(lib.mtn_ffn_sublayer_bwd(code, C.byref(args), L.stream_ptr()))
        if defer:
            probs = (L.GemmProblem * 2)()
            ln = L.LnFinalizeDesc()
            n = lib.mtn_ffn_param_grad_work(code, C.byref(args), probs, C.byref(ln))
            cfg.queue.add(code, [probs[i] for i in range(n)], ln, [ws_lp, ws_f32, hid, xn])
        if g is not None:
            return dx, None, None, None, None, None, None, None
        return dx, d_ln_a, d_ln_b, d_w1, d_b1, d_w2, d_b2, None


# ------------------------------------------------------------------------------------------ lockstep sublayer groups
@dataclass
class GroupMember:
    """One sublayer of a lockstep group.  kind 'mha': params = (ln_a, ln_b, b_qkv, b_o), cfg = MhaConfig with the compute-dtype
    weights and `grads` destinations set; kind 'ffn': params = (ln_a, ln_b, b1, b2), cfg = FfnConfig likewise."""
    kind: str
    cfg: object
    params: tuple
    mask: Optional[torch.Tensor] = None
    mem_lp: Optional[torch.Tensor] = None
    # gradient hand-off along a chain of sublayers (see include/mtn_hip.h, dyl_ready / next_dyl): `holder` travels with this
    # member's OUTPUT tensor to the sublayer that consumes it; `feeds` is the holder found on this member's INPUT tensor.
    holder: Optional[dict] = None
    feeds: Optional[dict] = None
    kv_ready: Optional[torch.Tensor] = None     # K|V of a constant memory, projected ahead of the layer loop (project_memories)
    want_lp: bool = False                       # 'ffn': also write the output in the compute dtype (-> out_lp): it is attended as raw memory
    out_lp: Optional[torch.Tensor] = None


class SublayerGroupFn(torch.autograd.Function):
    """Independent sublayers of one DecoderLayer executed in lockstep (one grouped launch per stage, see csrc/sublayer.hip).
    apply(members, x_0, mem_0, x_1, mem_1, ...) -> (y_0, y_1, ...); mem_i is None for self-attention / FFN members.
    Parameter gradients go straight to cfg.grads (flat gradient buffer) through cfg.queue (always deferred)."""

    @staticmethod
    def forward(ctx, members, *tensors):
        lib = L.load()
        xs = [t.contiguous() for t in tensors[0::2]]
        mems = list(tensors[1::2])
        _require_cuda(*xs)
        dev = xs[0].device
        lp = members[0].cfg.lp_dtype
        code = L.dtype_code(lp)
        n_mha = sum(1 for m in members if m.kind == "mha")
        n_ffn = len(members) - n_mha
        mha_args = (L.MhaArgs * max(1, n_mha))()
        ffn_args = (L.FfnArgs * max(1, n_ffn))()
        saved, ys = [], []
        im = jf = 0
        xaccs = []
        for x_orig in tensors[0::2]:
            # an input that an EARLIER sublayer already attends as memory (an auto-encoder output feeds the next layer's chain
            # and is the memory of x's attention): both gradients meet in one buffer instead of an autograd add
            xg = getattr(x_orig, "_mtn_gacc", None) if (_XACC and x_orig.requires_grad) else None     # (grad mode is off inside forward)
            if xg is not None:
                xg["remaining"] += 1
            xaccs.append(xg)
        for mem_t, x, mb in zip(mems, xs, members):
            cfg = mb.cfg
            ln_a, ln_b = mb.params[0], mb.params[1]
            y = torch.empty_like(x)
            d = x.size(-1)
            if mb.kind == "mha":
                B, a, _ = x.shape
                self_attn = mem_t is None
                m = a if self_attn else mem_t.size(1)
                mem_lp = mb.mem_lp
                if not self_attn and mem_lp is None:
                    mem_lp = cast_to_lp(mem_t, lp)
                mask_u8, sb, sq = _mask_u8(mb.mask, B, a, m)
                xn = torch.empty(B * a, d, device=dev, dtype=lp)
                mean = torch.empty(B * a, device=dev, dtype=torch.float32)
                rstd = torch.empty_like(mean)
                qkv = torch.empty(B * a, 3 * d if self_attn else d, device=dev, dtype=lp)
                kv = None if self_attn else (mb.kv_ready if mb.kv_ready is not None else torch.empty(B * m, 2 * d, device=dev, dtype=lp))
                o = torch.empty(B * a, d, device=dev, dtype=lp)
                lse = torch.empty(2 * B * cfg.heads * a, device=dev, dtype=torch.float32)
                A = mha_args[im]; im += 1
                A.kv_ready = int((not self_attn) and mb.kv_ready is not None)
                A.B, A.a, A.m, A.d, A.h = B, a, m, d, cfg.heads
                A.self_attn, A.ln_eps = int(self_attn), cfg.eps
                A.drop_attn = _drop(cfg.p_attn, cfg.salt * 4 + 0, cfg.seed)
                A.drop_out = _drop(cfg.p_out, cfg.salt * 4 + 1, cfg.seed)
                A.x, A.mem = x.data_ptr(), L.ptr(mem_lp)
                A.mask, A.mask_sb, A.mask_sq = L.ptr(mask_u8), sb, sq
                A.ln_a, A.ln_b = ln_a.data_ptr(), ln_b.data_ptr()
                A.w_qkv, A.b_qkv, A.w_o, A.b_o = cfg.w_qkv_lp.data_ptr(), mb.params[2].data_ptr(), cfg.w_o_lp.data_ptr(), mb.params[3].data_ptr()
                A.w_qkv_t, A.w_o_t = L.ptr(cfg.w_qkv_lpT), L.ptr(cfg.w_o_lpT)
                A.y, A.xn, A.mean, A.rstd = y.data_ptr(), xn.data_ptr(), mean.data_ptr(), rstd.data_ptr()
                A.qkv, A.kv, A.o, A.lse = qkv.data_ptr(), L.ptr(kv), o.data_ptr(), lse.data_ptr()
                A.ln_fold = L.ptr(cfg.ln_fold)
                need_dmem = (not self_attn) and mem_t.requires_grad
                gacc = None
                if need_dmem:          # a memory serves every layer: its gradient is accumulated in place by the dmem GEMMs
                    gacc = getattr(mem_t, "_mtn_gacc", None)
                    if gacc is None:
                        gacc = {"buf": None, "remaining": 0}
                        mem_t._mtn_gacc = gacc
                    gacc["remaining"] += 1
                saved.append(dict(x=x, mem_lp=mem_lp, mask=mask_u8, xn=xn, mean=mean, rstd=rstd, qkv=qkv, kv=kv, o=o, lse=lse,
                                  need_dmem=need_dmem, gacc=gacc, m=m,
                                  mem_next=getattr(mem_t, "_mtn_next", None) if need_dmem else None))   # the memory's producer (hand-off holder)
            else:
                rows = x.numel() // d
                ff = cfg.w1_lp.size(0)
                xn = torch.empty(rows, d, device=dev, dtype=lp)
                mean = torch.empty(rows, device=dev, dtype=torch.float32)
                rstd = torch.empty_like(mean)
                hid = torch.empty(rows, ff, device=dev, dtype=lp)
                A = ffn_args[jf]; jf += 1
                A.rows, A.d, A.d_ff, A.ln_eps = rows, d, ff, cfg.eps
                A.drop_hidden = _drop(cfg.p_hidden, cfg.salt * 4 + 2, cfg.seed)
                A.drop_out = _drop(cfg.p_out, cfg.salt * 4 + 1, cfg.seed)
                A.x, A.ln_a, A.ln_b = x.data_ptr(), ln_a.data_ptr(), ln_b.data_ptr()
                A.w1, A.b1, A.w2, A.b2 = cfg.w1_lp.data_ptr(), mb.params[2].data_ptr(), cfg.w2_lp.data_ptr(), mb.params[3].data_ptr()
                A.w1_t, A.w2_t = L.ptr(cfg.w1_lpT), L.ptr(cfg.w2_lpT)
                A.y, A.xn, A.mean, A.rstd, A.hid = y.data_ptr(), xn.data_ptr(), mean.data_ptr(), rstd.data_ptr(), hid.data_ptr()
                A.ln_fold = L.ptr(cfg.ln_fold)
                if mb.want_lp and lp != torch.float32:
                    mb.out_lp = torch.empty(y.shape, device=dev, dtype=lp)
                    A.y_lp = mb.out_lp.data_ptr()
                saved.append(dict(x=x, xn=xn, mean=mean, rstd=rstd, hid=hid))
            ys.append(y)
        L.check(lib.mtn_sublayer_group_fwd(code, n_mha, mha_args, n_ffn, ffn_args, L.stream_ptr()))
        ctx.members, ctx.saved_bufs, ctx.code = members, saved, code
        ctx.xaccs = xaccs
        ctx.mha_args, ctx.ffn_args, ctx.n_mha, ctx.n_ffn = mha_args, ffn_args, n_mha, n_ffn
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        lib = L.load()
        members, saved, code = ctx.members, ctx.saved_bufs, ctx.code
        mha_args, ffn_args = ctx.mha_args, ctx.ffn_args      # forward pointers are still valid (buffers are kept alive in `saved`)
        grads_out, keep_all = [], []
        im = jf = 0
        work = []
        post = []                    # (buffer, dx) pairs to sum after the launch: input-role accumulators that were not first
        for dy, mb, sv, xg in zip(dys, members, saved, ctx.xaccs):
            cfg, g = mb.cfg, mb.cfg.grads
            x = sv["x"]
            dev = x.device
            dy = dy.contiguous() if dy is not None else torch.zeros_like(x)
            dx = torch.empty_like(x)
            dx_ret = dx
            if xg is not None:           # see forward: the gradient of this input is collected in a shared buffer
                xg["remaining"] -= 1
                if xg["buf"] is None and xg["remaining"] > 0:
                    xg["buf"], dx_ret = dx, None                      # first writer: the memory-role users accumulate into our dx
                elif xg["buf"] is not None:
                    buf = xg["buf"]
                    post.append((buf, dx))
                    dx_ret = None
                    if xg["remaining"] == 0:
                        dx_ret, xg["buf"] = buf, None
            # hand-off, consumer side: the sublayer that ran before us in backward already wrote dy through OUR output dropout
            ready = None
            h = mb.holder
            if h is not None and h.get("dyl") is not None:
                # valid only if dy IS the dx that was handed off, untouched: same storage address, same shape, and the version
                # counter it had (autograd sums gradients in place with add_, which bumps it; any tensor derived from dx — a sum,
                # a clone — is allocated while dx is still alive, so it cannot sit at dx's address).  The dx tensor itself is not
                # held: a leaf's .grad may then take it over without a copy (layer-segmented backward).
                if h["dx_ptr"] == dy.data_ptr() and dy._version == h["ver"] and tuple(h["shape"]) == tuple(dy.shape):
                    ready = h["dyl"]
            if h is not None:
                h["dyl"] = h["dx_ptr"] = None
            # producer side: write our dx also as the next sublayer's masked compute-dtype dy
            nxt = None
            f = mb.feeds
            if f is not None and f.get("lp") == cfg.lp_dtype and xg is None:
                nxt = torch.empty(x.shape, device=dev, dtype=cfg.lp_dtype)
                f["dyl"], f["dx_ptr"], f["ver"], f["shape"] = nxt, dx.data_ptr(), dx._version, dx.shape
            if mb.kind == "mha":
                A = mha_args[im]; im += 1
                B, a, d, m = A.B, A.a, A.d, sv["m"]
                dmem, dmem_ret, accumulate = None, None, 0
                A.dmem_lp = None
                if sv["need_dmem"]:
                    gacc = sv["gacc"]
                    if gacc["buf"] is None:
                        gacc["buf"] = torch.empty(B, m, d, device=dev, dtype=torch.float32)
                    else:
                        accumulate = 1
                    dmem = gacc["buf"]
                    gacc["remaining"] -= 1
                    if gacc["remaining"] == 0:       # last user (first in forward order): hand the sum to autograd once
                        dmem_ret, gacc["buf"] = dmem, None
                        # ... and, through the dmem GEMM's epilogue, to the sublayer that produced the memory: the sum once
                        # more, through that sublayer's output dropout, in the compute dtype (its backward then needs no cast)
                        fm = sv.get("mem_next")
                        if fm is not None and fm.get("lp") == cfg.lp_dtype and cfg.lp_dtype != torch.float32 and _HANDOFF_MEM:
                            dmem_lp = torch.empty(dmem.shape, device=dev, dtype=cfg.lp_dtype)
                            fm["dyl"], fm["dx_ptr"], fm["ver"], fm["shape"] = dmem_lp, dmem.data_ptr(), dmem._version, dmem.shape
                            A.dmem_lp, A.dmem_lp_drop = dmem_lp.data_ptr(), _drop(fm["p"], fm["salt"], fm["seed"])
                ws_lp = torch.empty(lib.mtn_mha_bwd_ws_lp_elems(B, a, m, d, A.self_attn), device=dev, dtype=cfg.lp_dtype)
                ws_f32 = torch.empty(lib.mtn_mha_bwd_ws_f32_floats(B, a, m, d), device=dev, dtype=torch.float32)
                A.dy, A.dx, A.dmem, A.dmem_accumulate = dy.data_ptr(), dx.data_ptr(), L.ptr(dmem), accumulate
                A.d_ln_a, A.d_ln_b = g["ln_a"].data_ptr(), g["ln_b"].data_ptr()
                A.d_w_qkv, A.d_b_qkv, A.d_w_o, A.d_b_o = g["w_qkv"].data_ptr(), g["b_qkv"].data_ptr(), g["w_o"].data_ptr(), g["b_o"].data_ptr()
                A.ws_lp, A.ws_f32, A.defer_param_grads = ws_lp.data_ptr(), ws_f32.data_ptr(), 1
                A.ln_fold = L.ptr(cfg.ln_fold)
                A.dyl_ready, A.next_dyl = L.ptr(ready), L.ptr(nxt)
                if nxt is not None:
                    A.next_drop = _drop(f["p"], f["salt"], f["seed"])
                grads_out += [dx_ret, dmem_ret]
                keep = [dy, ws_lp, ws_f32, sv["o"], sv["xn"]] + ([sv["mem_lp"]] if sv["mem_lp"] is not None else []) + ([ready] if ready is not None else [])
                work.append(("mha", A, keep))
            else:
                A = ffn_args[jf]; jf += 1
                rows, d, ff = A.rows, A.d, A.d_ff
                ws_lp = torch.empty(rows * d + rows * ff, device=dev, dtype=cfg.lp_dtype)
                ws_f32 = torch.empty(lib.mtn_ffn_bwd_ws_f32_floats(rows, d, ff), device=dev, dtype=torch.float32)
                A.dy, A.dx = dy.data_ptr(), dx.data_ptr()
                A.d_ln_a, A.d_ln_b = g["ln_a"].data_ptr(), g["ln_b"].data_ptr()
                A.d_w1, A.d_b1, A.d_w2, A.d_b2 = g["w1"].data_ptr(), g["b1"].data_ptr(), g["w2"].data_ptr(), g["b2"].data_ptr()
                A.ws_lp, A.ws_f32, A.defer_param_grads = ws_lp.data_ptr(), ws_f32.data_ptr(), 1
                A.ln_fold = L.ptr(cfg.ln_fold)
                A.dyl_ready, A.next_dyl = L.ptr(ready), L.ptr(nxt)
                if nxt is not None:
                    A.next_drop = _drop(f["p"], f["salt"], f["seed"])
                grads_out += [dx_ret, None]
                keep = [dy, ws_lp, ws_f32, sv["hid"], sv["xn"]] + ([ready] if ready is not None else [])
                work.append(("ffn", A, keep))
            keep_all.append(dy)
        L.check(lib.mtn_sublayer_group_bwd(code, ctx.n_mha, mha_args, ctx.n_ffn, ffn_args, L.stream_ptr()))
        for buf, dxp in post:
            buf.add_(dxp)
        for kind, A, keep in work:
            probs = (L.GemmProblem * 3)()
            ln = L.LnFinalizeDesc()
            if kind == "mha":
                n = lib.mtn_mha_param_grad_work(code, C.byref(A), probs, C.byref(ln))
            else:
                n = lib.mtn_ffn_param_grad_work(code, C.byref(A), probs, C.byref(ln))
            members[0].cfg.queue.add(code, [probs[i] for i in range(n)], ln, keep)
        return (None, *grads_out)


def generator_log_probs(x, w_lp, bias):
    """Generator.forward at inference (mtn.py:68-69: log_softmax(proj(x))) on the HIP path: x (..., d) fp32 or compute dtype ->
    (..., V) fp32 log-probabilities.  One grouped-GEMM launch (x W^T + b, fp32 logits) and one row kernel (csrc/select.hip) —
    no vendor BLAS / softmax kernels in a decode step.  No autograd: training goes through GeneratorLossFn (fused loss head)."""
    _require_cuda(x, w_lp)
    lib = L.load()
    lp = w_lp.dtype
    V, d = w_lp.shape
    a = x.reshape(-1, d)
    a = a.contiguous() if a.dtype == lp else cast_to_lp(a.contiguous(), lp)
    rows = a.size(0)
    out = torch.empty(rows, V, device=x.device, dtype=torch.float32)
    pr = L.GemmProblem()
    pr.A, pr.B, pr.lda, pr.ldb, pr.M, pr.N, pr.K = a.data_ptr(), w_lp.data_ptr(), d, d, rows, V, d
    pr.bias, pr.gate_scale, pr.out_f32, pr.ldc = bias.data_ptr(), 1.0, out.data_ptr(), V
    gemm(L.dtype_code(lp), [pr])
    L.check(lib.mtn_log_softmax_rows(out.data_ptr(), rows, V, V, out.data_ptr(), V, L.stream_ptr()))
    return out.view(*x.shape[:-1], V)


def topk_rows(x, k, extra_col=-1):
    """x (rows, V) fp32 -> (rows, 2k+1) fp32: per row its k largest entries in descending order, their column indices (as floats)
    and x[row, extra_col] — the part of a log-probability row the beam search looks at (data_utils.py:219), csrc/select.hip."""
    _require_cuda(x)
    x = x.contiguous()
    out = torch.empty(x.size(0), 2 * k + 1, device=x.device, dtype=torch.float32)
    L.check(L.load().mtn_topk_rows(x.data_ptr(), x.size(0), x.size(1), x.stride(0), k, extra_col, out.data_ptr(), L.stream_ptr()))
    return out


# ------------------------------------------------------------------------------------------ memory K/V, ahead of the layers
def project_memories(items, lp_dtype, outs=None):
    """K|V projections (mtn.py:257-258) of CONSTANT memories for many sublayers at once: items = [(mem_lp (B,m,d) compute dtype,
    w_qkv_lp (3d,d), b_qkv (3d,) fp32)] -> list of (B*m, 2d) compute-dtype tensors.  The encoder-side memories (history,
    caption, query, video) are the same tensors in every decoder layer, so their N x (3+F) projections do not depend on
    anything computed inside the layer loop: they run here as a few large grouped GEMMs instead of riding in N x (3+F)
    small per-sublayer launches (and, in the decode path, once per dialogue instead of once per token)."""
    if not items:
        return []
    code = L.dtype_code(lp_dtype)
    reuse, outs, probs = outs, [], []          # `outs`: write into these buffers (a captured graph keeps reading them)
    for k, (mem_lp, w, b) in enumerate(items):
        Bm = mem_lp.size(0) * mem_lp.size(1)
        d = mem_lp.size(-1)
        kv = reuse[k] if reuse is not None else torch.empty(Bm, 2 * d, device=mem_lp.device, dtype=lp_dtype)
        p = L.GemmProblem()
        p.A, p.B, p.lda, p.ldb, p.M, p.N, p.K = mem_lp.data_ptr(), w.data_ptr() + d * d * w.element_size(), d, d, Bm, 2 * d, d
        p.bias, p.gate_scale, p.out_lp, p.ldc = b.data_ptr() + d * b.element_size(), 1.0, kv.data_ptr(), 2 * d
        outs.append(kv)
        probs.append(p)
    for i in range(0, len(probs), L.GEMM_MAX_GROUP):
        gemm(code, probs[i:i + L.GEMM_MAX_GROUP])
    return outs


# ------------------------------------------------------------------------------------------ fused embeddings
class EmbedNormFn(torch.autograd.Function):
    """Embeddings (lut[tok] * sqrt(d), mtn.py:288-289) + PositionalEncoding (+pe, dropout, mtn.py:307-309) + the Encoder's
    LayerNorm for that stream (mtn.py:83-101), for several token streams in ONE grouped launch; backward = grouped LayerNorm
    backward + one grouped scatter-add into the embedding-table gradients.
    apply(spec, *luts) -> one fp32 tensor per stream.  spec["streams"]: dicts with tokens (B,L) int64, lut (index into luts),
    pe (max_len,d), scale, p (dropout), salt, ln = None | (a2, b2, eps, grad_a, grad_b).  Table gradients are accumulated into
    lut.grad when it exists (flat gradient buffer), else returned."""

    @staticmethod
    def forward(ctx, spec, *luts):
        lib = L.load()
        streams = spec["streams"]
        lp = spec["lp_dtype"]
        code = L.dtype_code(lp) if lp is not None else L.MTN_F32
        dev = luts[0].device
        descs = (L.LnFwdDesc * len(streams))()
        outs, saved = [], []
        for i, st in enumerate(streams):
            tok = st["tokens"].contiguous()
            B, Ls = tok.shape
            d = luts[st["lut"]].size(1)
            rows = B * Ls
            y = torch.empty(B, Ls, d, device=dev, dtype=torch.float32)
            D = descs[i]
            D.rows, D.d, D.tokens, D.lut, D.emb_scale = rows, d, tok.data_ptr(), luts[st["lut"]].data_ptr(), float(st["scale"])
            D.pe, D.seq_len = st["pe"].data_ptr(), Ls
            D.drop = _drop(st["p"], st["salt"], spec["seed"])
            sv = dict(tok=tok, rows=rows, d=d)
            if st["ln"] is not None:
                a2, b2, eps = st["ln"][0], st["ln"][1], st["ln"][2]
                x_save = torch.empty(rows, d, device=dev, dtype=torch.float32)
                mean = torch.empty(rows, device=dev, dtype=torch.float32)
                rstd = torch.empty_like(mean)
                y_lp = torch.empty(B, Ls, d, device=dev, dtype=lp) if (lp is not None and lp != torch.float32) else None
                D.eps, D.a2, D.b2, D.x_out, D.mean, D.rstd = eps, a2.data_ptr(), b2.data_ptr(), x_save.data_ptr(), mean.data_ptr(), rstd.data_ptr()
                D.y_f32, D.y_lp, D.no_ln = y.data_ptr(), L.ptr(y_lp), 0
                sv.update(x=x_save, mean=mean, rstd=rstd, a2=a2, y_lp=y_lp)
            else:
                D.y_f32, D.no_ln = y.data_ptr(), 1
            outs.append(y)
            saved.append(sv)
        L.check(lib.mtn_layernorm_fwd_group(code, len(streams), descs, L.stream_ptr()))
        ctx.spec, ctx.saved_streams, ctx.luts = spec, saved, luts
        ctx.lp_copies = [sv.get("y_lp") for sv in saved]
        spec["_lp_out"] = ctx.lp_copies
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        lib = L.load()
        spec, saved, luts = ctx.spec, ctx.saved_streams, ctx.luts
        streams = spec["streams"]
        dev = luts[0].device
        queue = spec.get("queue")
        ln_descs, ln_final, emb = [], [], (L.EmbedBwdDesc * len(streams))()
        keep = []
        ret = [None] * len(luts)
        dluts = []
        for j, lut in enumerate(luts):
            if lut.grad is not None:
                dluts.append(lut.grad)
            else:
                g = torch.zeros_like(lut)
                dluts.append(g)
                ret[j] = g
        for i, (st, sv, dy) in enumerate(zip(streams, saved, dys)):
            dy = dy.contiguous() if dy is not None else torch.zeros(sv["rows"], sv["d"], device=dev)
            dx = dy
            if st["ln"] is not None:
                dx = torch.empty(sv["rows"], sv["d"], device=dev, dtype=torch.float32)
                partial = torch.empty(lib.mtn_layernorm_bwd_partial_floats(sv["rows"], sv["d"]), device=dev, dtype=torch.float32)
                ln_descs.append(L.LnBwdDesc(sv["rows"], sv["d"], st["ln"][2], sv["x"].data_ptr(), sv["a2"].data_ptr(), sv["mean"].data_ptr(),
                                            sv["rstd"].data_ptr(), dy.data_ptr(), None, dx.data_ptr(), partial.data_ptr()))
                ga, gb = st["ln"][3], st["ln"][4]
                ln_final.append((L.LnFinalizeDesc(partial.data_ptr(), lib.mtn_layernorm_bwd_nparts(sv["rows"]), sv["d"], ga.data_ptr(), gb.data_ptr()), partial))
            keep += [dy, dx]
            E = emb[i]
            E.rows, E.d, E.tokens, E.dx, E.emb_scale = sv["rows"], sv["d"], sv["tok"].data_ptr(), dx.data_ptr(), float(st["scale"])
            E.drop = _drop(st["p"], st["salt"], spec["seed"])
            E.dlut = dluts[st["lut"]].data_ptr()
            E.lut_rows = luts[st["lut"]].size(0)
        if ln_descs:
            arr = (L.LnBwdDesc * len(ln_descs))(*ln_descs)
            L.check(lib.mtn_layernorm_bwd_group(len(ln_descs), arr, L.stream_ptr()))
            if queue is not None:
                for desc, partial in ln_final:
                    queue.add(queue.dtype, [], desc, [partial])
            else:
                farr = (L.LnFinalizeDesc * len(ln_final))(*[f[0] for f in ln_final])
                L.check(lib.mtn_layernorm_bwd_finalize(len(ln_final), farr, L.stream_ptr()))
        if queue is not None and all(r is None for r in ret):
            # every table's gradient lives in the flat gradient buffer: the scatter can wait for the end of backward and share its
            # launch with the other tables' (the queue's flush); dx / dy stay alive in the queue until then
            queue.add_embed([emb[i] for i in range(len(streams))], keep)
        else:
            L.check(lib.mtn_embed_bwd_group(len(streams), emb, L.stream_ptr()))
        return (None, *ret)


# ------------------------------------------------------------------------------------------ feature streams
class FeatureEncodeFn(torch.autograd.Function):
    """vid_encoder of mtn.py:378 for ALL feature streams at once: Linear(ft_i, d) -> ReLU -> + positional encoding -> dropout,
    followed by that stream's Encoder LayerNorm (mtn.py:83-101).  Forward = one grouped cast of the features, one grouped
    MFMA GEMM (bias + ReLU in the epilogue), one grouped LayerNorm launch (PE, dropout, LN, compute-dtype copy); backward =
    grouped LayerNorm backward, one grouped mask+cast (dropout, ReLU), dW/db GEMMs on the deferred queue.
    apply(spec, *weights) -> one fp32 (B, V_i, d) tensor per stream.  spec["streams"]: dicts with x (B,V,ft) fp32, w (index),
    w_lp, bias, grad_w, grad_b, pe, p, salt, ln = (a2, b2, eps, grad_a, grad_b).  Features get no gradient (inputs)."""

    @staticmethod
    def forward(ctx, spec, *weights):
        lib = L.load()
        streams, lp = spec["streams"], spec["lp_dtype"]
        code = L.dtype_code(lp)
        dev = weights[0].device
        n = len(streams)
        casts = (L.CastDesc * n)()
        probs, lns, saved, outs = [], (L.LnFwdDesc * n)(), [], []
        for i, st in enumerate(streams):
            x = st["x"].contiguous()
            B, V, F = x.shape
            d = st["w_lp"].size(0)
            rows = B * V
            x_lp = torch.empty(rows, F, device=dev, dtype=lp)
            casts[i].n, casts[i].src, casts[i].dst = rows * F, x.data_ptr(), x_lp.data_ptr()
            h = torch.empty(rows, d, device=dev, dtype=torch.float32)
            p = L.GemmProblem()
            p.A, p.B, p.lda, p.ldb, p.M, p.N, p.K = x_lp.data_ptr(), st["w_lp"].data_ptr(), F, F, rows, d, F
            p.bias, p.relu, p.gate_scale, p.out_f32, p.ldc = st["bias"].data_ptr(), 1, 1.0, h.data_ptr(), d
            probs.append(p)
            a2, b2, eps = st["ln"][0], st["ln"][1], st["ln"][2]
            y = torch.empty(B, V, d, device=dev, dtype=torch.float32)
            y_lp = torch.empty(B, V, d, device=dev, dtype=lp) if lp != torch.float32 else None
            xs = torch.empty(rows, d, device=dev, dtype=torch.float32)
            mean = torch.empty(rows, device=dev, dtype=torch.float32)
            rstd = torch.empty_like(mean)
            D = lns[i]
            D.rows, D.d, D.eps, D.x, D.a2, D.b2 = rows, d, eps, h.data_ptr(), a2.data_ptr(), b2.data_ptr()
            D.pe, D.seq_len, D.drop = st["pe"].data_ptr(), V, _drop(st["p"], st["salt"], spec["seed"])
            D.x_out, D.mean, D.rstd, D.y_f32, D.y_lp = xs.data_ptr(), mean.data_ptr(), rstd.data_ptr(), y.data_ptr(), L.ptr(y_lp)
            outs.append(y)
            saved.append(dict(x_lp=x_lp, h=h, xs=xs, mean=mean, rstd=rstd, rows=rows, d=d, F=F, y_lp=y_lp, a2=a2))
        if lp != torch.float32:
            L.check(lib.mtn_cast_group(code, n, casts, L.stream_ptr()))
        else:
            for sv, st in zip(saved, streams):
                sv["x_lp"].copy_(st["x"].reshape(sv["rows"], sv["F"]))
        gemm(code, probs)
        L.check(lib.mtn_layernorm_fwd_group(code, n, lns, L.stream_ptr()))
        ctx.spec, ctx.saved_streams, ctx.n_w = spec, saved, len(weights)
        spec["_lp_out"] = [sv["y_lp"] for sv in saved]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        lib = L.load()
        spec, saved = ctx.spec, ctx.saved_streams
        streams, lp = spec["streams"], spec["lp_dtype"]
        code = L.dtype_code(lp)
        queue = spec.get("queue")
        n = len(streams)
        dev = saved[0]["h"].device
        lnb = (L.LnBwdDesc * n)()
        casts = (L.CastDesc * n)()
        probs, finals, keep = [], [], []
        for i, (st, sv, dy) in enumerate(zip(streams, saved, dys)):
            rows, d, F = sv["rows"], sv["d"], sv["F"]
            dy = dy.contiguous() if dy is not None else torch.zeros(rows, d, device=dev)
            dxs = torch.empty(rows, d, device=dev, dtype=torch.float32)
            partial = torch.empty(lib.mtn_layernorm_bwd_partial_floats(rows, d), device=dev, dtype=torch.float32)
            B_ = lnb[i]
            B_.rows, B_.d, B_.eps, B_.x, B_.a2, B_.mean, B_.rstd = rows, d, st["ln"][2], sv["xs"].data_ptr(), sv["a2"].data_ptr(), sv["mean"].data_ptr(), sv["rstd"].data_ptr()
            B_.g, B_.dx, B_.partial = dy.data_ptr(), dxs.data_ptr(), partial.data_ptr()
            finals.append((L.LnFinalizeDesc(partial.data_ptr(), lib.mtn_layernorm_bwd_nparts(rows), d, st["ln"][3].data_ptr(), st["ln"][4].data_ptr()), partial))
            dh = torch.empty(rows, d, device=dev, dtype=lp)
            casts[i].n, casts[i].src, casts[i].dst = rows * d, dxs.data_ptr(), dh.data_ptr()
            casts[i].drop, casts[i].gate = _drop(st["p"], st["salt"], spec["seed"]), sv["h"].data_ptr()
            p = L.GemmProblem()      # dW = dh^T x, db = column sums of dh
            p.A, p.B, p.lda, p.ldb, p.M, p.N, p.K, p.a_trans, p.b_trans = dh.data_ptr(), sv["x_lp"].data_ptr(), d, F, d, F, rows, 1, 1
            p.gate_scale, p.out_f32, p.ldc, p.rowsum_out = 1.0, st["grad_w"].data_ptr(), F, st["grad_b"].data_ptr()
            probs.append(p)
            keep += [dy, dxs, dh, sv["x_lp"], sv["h"]]
        L.check(lib.mtn_layernorm_bwd_group(n, lnb, L.stream_ptr()))
        L.check(lib.mtn_cast_group(code, n, casts, L.stream_ptr()))
        if queue is not None:
            for k, (desc, partial) in enumerate(finals):
                queue.add(code, [probs[k]], desc, [partial] + (keep if k == 0 else []))
        else:
            gemm(code, probs)
            farr = (L.LnFinalizeDesc * n)(*[f[0] for f in finals])
            L.check(lib.mtn_layernorm_bwd_finalize(n, farr, L.stream_ptr()))
        return (None,) + (None,) * ctx.n_w


# ------------------------------------------------------------------------------------------ loss head
class GeneratorLossFn(torch.autograd.Function):
    """sum_i coef_i * KLDiv(log_softmax(x_i W_i^T + b_i), smooth(y_i)) / norm_i  — Generator (mtn.py:62-69) +
    LabelSmoothing (label_smoothing.py) + SimpleLossCompute's weighted sum (data_utils.py:133-144) as: one grouped cast,
    one grouped GEMM (logits), one row kernel (log-sum-exp + closed-form KL); backward: one row kernel (dlogits), two
    grouped GEMMs (dX, dW+db).  `spec`: targets, norms (device float scalars), coefs, gens [(w_lp, bias, grad_w, grad_b)],
    vocab, pad, smoothing, lp_dtype.  Generator gradients are written to the flat gradient buffer (not returned)."""

    @staticmethod
    def forward(ctx, spec, *xs):
        lib = L.load()
        lp = spec["lp_dtype"]
        code = L.dtype_code(lp)
        dev = xs[0].device
        d, V = xs[0].size(-1), spec["vocab"]
        rows = [x.numel() // d for x in xs]
        R = sum(rows)
        # inputs that are consecutive row ranges of one compute-dtype buffer already (layer_norm_group): no cast
        ready, off = None, 0
        for i, x in enumerate(xs):
            tag = getattr(x, "_mtn_lp_rows", None)
            if tag is None or tag[1] != off or tag[2] != rows[i] or tag[0].dtype != lp or (ready is not None and tag[0] is not ready):
                ready = None
                break
            ready, off = tag[0], off + rows[i]
        if ready is not None and ready.size(0) != R:
            ready = None
        x_lp = ready if ready is not None else torch.empty(R, d, device=dev, dtype=lp)
        logits = torch.empty(R, V, device=dev, dtype=torch.float32)
        casts = (L.CastDesc * len(xs))()
        off = 0
        offs = []
        for i, x in enumerate(xs):
            xc = x.contiguous()
            casts[i].n, casts[i].src, casts[i].dst = xc.numel(), xc.data_ptr(), x_lp[off:off + rows[i]].data_ptr()
            casts[i].drop = L.Dropout(0.0, 0, None)
            offs.append(off)
            off += rows[i]
            ctx_keep = xc
        if ready is not None:
            pass
        elif lp == torch.float32:
            for i, x in enumerate(xs):
                x_lp[offs[i]:offs[i] + rows[i]].copy_(x.reshape(rows[i], d))
        else:
            L.check(lib.mtn_cast_group(code, len(xs), casts, L.stream_ptr()))
        gens = spec["gens"]
        shared = all(g[0].data_ptr() == gens[0][0].data_ptr() for g in gens)
        segs = [(0, R, gens[0])] if shared else [(offs[i], rows[i], gens[i]) for i in range(len(xs))]
        probs = []
        for o, r, (w_lp, bias, _gw, _gb, *_rest) in segs:
            p = L.GemmProblem()
            p.A, p.B, p.lda, p.ldb, p.M, p.N, p.K = x_lp[o:].data_ptr(), w_lp.data_ptr(), d, d, r, V, d
            p.bias, p.gate_scale, p.out_f32, p.ldc = bias.data_ptr(), 1.0, logits[o:].data_ptr(), V
            probs.append(p)
        gemm(code, probs)
        A = L.LossHeadArgs()
        A.n_seg = len(xs)
        norms = [n.float().reshape(1) if n.dtype != torch.float32 or n.dim() == 0 else n for n in spec["norms"]]
        targets = [t.contiguous().view(-1) for t in spec["targets"]]
        for i in range(len(xs)):
            A.rows[i], A.target[i], A.norm[i], A.coef[i] = rows[i], targets[i].data_ptr(), norms[i].data_ptr(), float(spec["coefs"][i])
        A.V, A.ldz, A.pad, A.smoothing = V, V, spec["pad"], float(spec["smoothing"])
        lse = torch.empty(R, device=dev, dtype=torch.float32)
        rowloss = torch.empty(R, device=dev, dtype=torch.float32)
        A.logits, A.lse, A.rowloss = logits.data_ptr(), lse.data_ptr(), rowloss.data_ptr()
        L.check(lib.mtn_losshead_fwd(C.byref(A), L.stream_ptr()))
        ctx.args, ctx.keep = A, (x_lp, logits, lse, norms, targets)
        ctx.meta = (spec, segs, rows, offs, d, V, R, code, [x.shape for x in xs])
        return rowloss.sum()

    @staticmethod
    def backward(ctx, g):
        lib = L.load()
        spec, segs, rows, offs, d, V, R, code, shapes = ctx.meta
        x_lp, logits, lse, norms, targets = ctx.keep
        lp = spec["lp_dtype"]
        dev = x_lp.device
        A = ctx.args
        gl = g.contiguous().float().reshape(1)
        dlogits = torch.empty(R, V, device=dev, dtype=lp)
        A.gloss, A.dlogits, A.ldd = gl.data_ptr(), dlogits.data_ptr(), V
        L.check(lib.mtn_losshead_bwd(code, C.byref(A), L.stream_ptr()))
        dx = torch.empty(R, d, device=dev, dtype=torch.float32)
        p_dx, p_dw = [], []
        for o, r, (w_lp, bias, gw, gb, *rest) in segs:
            p = L.GemmProblem()      # dX = dlogits W  (through the transposed weight copy when the model keeps one)
            w_lpT = rest[0] if rest else None
            if w_lpT is not None:
                p.A, p.B, p.lda, p.ldb, p.M, p.N, p.K, p.b_trans = dlogits[o:].data_ptr(), w_lpT.data_ptr(), V, V, r, d, V, 0
            else:
                p.A, p.B, p.lda, p.ldb, p.M, p.N, p.K, p.b_trans = dlogits[o:].data_ptr(), w_lp.data_ptr(), V, d, r, d, V, 1
            p.gate_scale, p.out_f32, p.ldc = 1.0, dx[o:].data_ptr(), d
            p_dx.append(p)
            q = L.GemmProblem()      # dW = dlogits^T x, db = column sums of dlogits
            q.A, q.B, q.lda, q.ldb, q.M, q.N, q.K, q.a_trans, q.b_trans = dlogits[o:].data_ptr(), x_lp[o:].data_ptr(), V, d, V, d, r, 1, 1
            q.gate_scale, q.out_f32, q.ldc, q.rowsum_out = 1.0, gw.data_ptr(), d, gb.data_ptr()
            p_dw.append(q)
        gemm(code, p_dx)
        queue = spec.get("queue")
        if queue is not None and queue.adam is not None:
            queue.add(code, p_dw, None, [dlogits, x_lp])      # with the optimiser epilogue: one of the problems of the table launch
        else:
            gemm(code, p_dw)
        return (None,) + tuple(dx[offs[i]:offs[i] + rows[i]].view(shapes[i]) for i in range(len(rows)))


# ------------------------------------------------------------------------------------------ attention core (tests / decode)
def attention(q, k, v, mask, heads: int, p_drop: float = 0.0, seed=None, salt: int = 0):
    """mtn.py:221-231 on packed-head layouts: q (B,a,d), k/v (B,m,d) of the compute dtype.  Returns (o, lse)."""
    _require_cuda(q, k, v)
    B, a, d = q.shape
    m = k.size(1)
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    o = torch.empty_like(q)
    lse = torch.empty(2 * B * heads * a, device=q.device, dtype=torch.float32)
    mask_u8, sb, sq = _mask_u8(mask, B, a, m)
    args = L.AttnArgs()
    args.B, args.h, args.a, args.m, args.dk = B, heads, a, m, d // heads
    args.q, args.k, args.v, args.ldq, args.ldkv = q.data_ptr(), k.data_ptr(), v.data_ptr(), d, d
    args.mask, args.mask_sb, args.mask_sq = L.ptr(mask_u8), sb, sq
    args.drop = _drop(p_drop, salt, seed)
    args.o, args.ldo, args.lse = o.data_ptr(), d, lse.data_ptr()
    L.check(L.load().mtn_attention_fwd(L.dtype_code(q.dtype), C.byref(args), L.stream_ptr()))
    return o, lse


def attention_bwd(q, k, v, o, lse, d_o, mask, heads: int, p_drop: float = 0.0, seed=None, salt: int = 0):
    B, a, d = q.shape
    m = k.size(1)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    mask_u8, sb, sq = _mask_u8(mask, B, a, m)
    args = L.AttnArgs()
    args.B, args.h, args.a, args.m, args.dk = B, heads, a, m, d // heads
    args.q, args.k, args.v, args.ldq, args.ldkv = q.data_ptr(), k.data_ptr(), v.data_ptr(), d, d
    args.mask, args.mask_sb, args.mask_sq = L.ptr(mask_u8), sb, sq
    args.drop = _drop(p_drop, salt, seed)
    args.o, args.ldo, args.lse = o.data_ptr(), d, lse.data_ptr()
    args.d_o, args.dq, args.dk_out, args.dv_out = d_o.contiguous().data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    kv_acc = None
    if a > 32 and q.dtype != torch.float32:      # several query-block passes: their dK / dV sums stay in fp32 until the last one
        kv_acc = torch.empty(2, B * m, d, device=q.device, dtype=torch.float32)
        args.kv_acc = kv_acc.data_ptr()
    L.check(L.load().mtn_attention_bwd(L.dtype_code(q.dtype), C.byref(args), L.stream_ptr()))
    return dq, dk, dv


# ------------------------------------------------------------------------------------------ un-fused operator forms
class AttentionCoreFn(torch.autograd.Function):
    """attention() of mtn.py:221-231 with autograd, on (B,L,d) packed-head tensors of the compute dtype."""

    @staticmethod
    def forward(ctx, q, k, v, mask, heads, p_drop, seed, salt):
        o, lse = attention(q, k, v, mask, heads, p_drop, seed, salt)
        ctx.save_for_backward(q.contiguous(), k.contiguous(), v.contiguous(), o, lse)
        ctx.meta = (mask, heads, p_drop, seed, salt)
        return o

    @staticmethod
    def backward(ctx, d_o):
        q, k, v, o, lse = ctx.saved_tensors
        mask, heads, p_drop, seed, salt = ctx.meta
        dq, dk, dv = attention_bwd(q, k, v, o, lse, d_o.to(q.dtype), mask, heads, p_drop, seed, salt)
        return dq, dk, dv, None, None, None, None, None


class LinearFn(torch.autograd.Function):
    """y = [relu](x W^T + b) on the MFMA GEMM (nn.Linear as used at mtn.py:243-244,273-276,378)."""

    @staticmethod
    def forward(ctx, x, w, b, lp_dtype, relu, out_f32):
        _require_cuda(x, w)
        code = L.dtype_code(lp_dtype)
        K, N = w.size(1), w.size(0)
        x2 = x.reshape(-1, K)
        x_lp = x2.contiguous() if x2.dtype == lp_dtype else cast_to_lp(x2.float(), lp_dtype)
        w_lp = w.contiguous() if w.dtype == lp_dtype else cast_to_lp(w.float(), lp_dtype)
        M = x_lp.size(0)
        out = torch.empty(M, N, device=x.device, dtype=torch.float32 if out_f32 else lp_dtype)
        p = L.GemmProblem()
        p.A, p.B, p.lda, p.ldb, p.M, p.N, p.K = x_lp.data_ptr(), w_lp.data_ptr(), K, K, M, N, K
        p.bias, p.relu, p.gate_scale, p.ldc = L.ptr(b), int(relu), 1.0, N
        if out.dtype == torch.float32:
            p.out_f32 = out.data_ptr()
        else:
            p.out_lp = out.data_ptr()
        gemm(code, [p])
        ctx.save_for_backward(x_lp, w_lp, out if relu else None)
        ctx.meta = (lp_dtype, relu, x.shape, x.dtype, b is not None)
        return out.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x_lp, w_lp, out = ctx.saved_tensors
        lp_dtype, relu, xshape, xdtype, has_b = ctx.meta
        code = L.dtype_code(lp_dtype)
        M, K = x_lp.shape
        N = w_lp.size(0)
        dy2 = dy.reshape(M, N)
        if relu:
            dy2 = dy2 * (out > 0).to(dy2.dtype)
        dy_lp = dy2.contiguous() if dy2.dtype == lp_dtype else cast_to_lp(dy2.float(), lp_dtype)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, device=dy.device, dtype=torch.float32)
            p = L.GemmProblem()
            p.A, p.B, p.lda, p.ldb, p.M, p.N, p.K, p.b_trans = dy_lp.data_ptr(), w_lp.data_ptr(), N, K, M, K, N, 1
            p.gate_scale, p.out_f32, p.ldc = 1.0, dx.data_ptr(), K
            gemm(code, [p])
            dx = dx.view(xshape).to(xdtype)
        dw = torch.empty(N, K, device=dy.device, dtype=torch.float32)
        db = torch.empty(N, device=dy.device, dtype=torch.float32)
        p = L.GemmProblem()
        p.A, p.B, p.lda, p.ldb, p.M, p.N, p.K, p.a_trans, p.b_trans = dy_lp.data_ptr(), x_lp.data_ptr(), N, K, N, K, M, 1, 1
        p.gate_scale, p.out_f32, p.ldc, p.rowsum_out = 1.0, dw.data_ptr(), K, db.data_ptr()
        gemm(code, [p])
        return dx, dw, (db if has_b else None), None, None, None


def linear(x, w, b, lp_dtype=torch.bfloat16, relu=False, out_f32=False):
    return LinearFn.apply(x, w, b, lp_dtype, relu, out_f32)
