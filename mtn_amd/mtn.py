"""MTN model on MI355X: same ``make_model()`` / module-tree / ``state_dict`` surface as the reference's
mtn.py, with the decoder hot path executed by fused HIP sublayer kernels (mtn_amd.ops).

What is kept from the reference (so that run.sh / train.py / generate.py-style callers drop in):
  * ``make_model(src_vocab, tgt_vocab, N, d_model, d_ff, h, dropout, separate_his_embed, separate_cap_embed,
    ft_sizes, diff_encoder, diff_embed, diff_gen, auto_encoder_ft, auto_encoder_attn)``  (mtn.py:332-337)
  * ``EncoderDecoder.forward(b) / encode(...) / decode(...) / vid_encode(...)``, ``.generator``,
    ``.auto_encoder_generator``                                                         (mtn.py:28-60)
  * the state_dict key schema, so reference checkpoints load with ``load_state_dict``.
What is different: parameters live in ONE flat fp32 buffer (plus a flat compute-dtype copy and a flat
gradient buffer) laid out [glue | every vector | per-layer weight matrices] so that the q/k/v Linears of every
attention are one [3d,d] matrix; sublayers run as lockstep groups of fused HIP launches (LayerNorm + head-slice
projections + attention in one kernel, output projection / w_2 + bias + dropout + residual in a grouped GEMM;
backward: fused dO + attention backward, then the dLN-out GEMM with LayerNorm backward in its epilogue); backward
writes parameter gradients straight into the flat gradient buffer — or, on one rank, applies Adam inside the
parameter-gradient launch.  Embeddings + positional encoding + Encoder LayerNorm, the feature Linears, the loss
head and (at inference) the generator run on library kernels too: a captured train step launches nothing but
libmtn_hip kernels apart from one fill, one scalar sum and one counter add.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import lib as L
from . import ops

ATTN_DROPOUT_DEFAULT = 0.1   # mtn.py:339 builds MultiHeadedAttention(h, d_model) without forwarding `dropout`


_HANDOFF = True       # gradient hand-off along a chain: the consumer's backward writes the producer's masked compute-dtype dy (no cast launches)


# ------------------------------------------------------------------------------------------ leaf modules
class LayerNorm(nn.Module):
    """a_2 * (x - mean) / (std_unbiased + eps) + b_2  (mtn.py:103-114) on the HIP kernel."""

    def __init__(self, features: int, eps: float = 1e-6):
        super().__init__()
        self.a_2 = nn.Parameter(torch.ones(features))
        self.b_2 = nn.Parameter(torch.zeros(features))
        self.eps = eps
        self._grads = None          # (flat-grad views) once the owner model is flattened
        self._lp_dtype = None
        self._queue = None

    def forward_lp(self, x):
        ga, gb = self._grads if self._grads is not None else (None, None)
        return ops.layer_norm(x, self.a_2, self.b_2, self.eps, self._lp_dtype, ga, gb, self._queue)

    def forward(self, x):
        return self.forward_lp(x)[0]


class MultiHeadedAttention(nn.Module):
    """Parameter container + standalone operator for mtn.py:233-267.  ``linears[0..2]`` = q,k,v input
    projections, ``linears[3]`` = output projection."""

    def __init__(self, h: int, d_model: int, dropout: float = ATTN_DROPOUT_DEFAULT):
        super().__init__()
        assert d_model % h == 0
        self.d_k = d_model // h
        self.h = h
        self.linears = nn.ModuleList([nn.Linear(d_model, d_model) for _ in range(4)])
        self.attn = None
        self.p = dropout
        self._fused = None          # dict of fused views, set by EncoderDecoder._flatten

    def fused(self):
        if self._fused is None:     # stand-alone use (tests): build packed weights on the fly, grads via autograd
            w = torch.cat([self.linears[i].weight for i in range(3)], 0)
            b = torch.cat([self.linears[i].bias for i in range(3)], 0)
            return dict(w_qkv=w, b_qkv=b, w_o=self.linears[3].weight, b_o=self.linears[3].bias,
                        w_qkv_lp=None, w_o_lp=None, grads=None)
        return self._fused

    def forward(self, query, key, value, mask=None):
        """Un-fused operator form (projections + attention core + output projection), no residual/LayerNorm."""
        lp = torch.bfloat16 if self._fused is None else self._fused["lp_dtype"]
        f = self.fused()
        B, a, d = query.shape
        wq, wk, wv = f["w_qkv"][:d], f["w_qkv"][d:2 * d], f["w_qkv"][2 * d:]
        bq, bk, bv = f["b_qkv"][:d], f["b_qkv"][d:2 * d], f["b_qkv"][2 * d:]
        q = ops.linear(query, wq, bq, lp)
        k = ops.linear(key, wk, bk, lp)
        v = ops.linear(value, wv, bv, lp)
        o = ops.AttentionCoreFn.apply(q, k, v, mask, self.h, self.p if self.training else 0.0, None, 0)
        return ops.linear(o, f["w_o"], f["b_o"], lp, out_f32=True)


class PositionwiseFeedForward(nn.Module):
    """Parameter container for mtn.py:269-280."""

    def __init__(self, d_model: int, d_ff: int, dropout: float = 0.1):
        super().__init__()
        self.w_1 = nn.Linear(d_model, d_ff)
        self.w_2 = nn.Linear(d_ff, d_model)
        self.p = dropout
        self._fused = None

    def fused(self):
        if self._fused is None:
            return dict(w1=self.w_1.weight, b1=self.w_1.bias, w2=self.w_2.weight, b2=self.w_2.bias,
                        w1_lp=None, w2_lp=None, grads=None)
        return self._fused

    def forward(self, x):
        lp = torch.bfloat16 if self._fused is None else self._fused["lp_dtype"]
        f = self.fused()
        hdn = ops.linear(x, f["w1"], f["b1"], lp, relu=True)
        if self.training and self.p > 0:
            hdn = F.dropout(hdn, self.p)
        return ops.linear(hdn, f["w2"], f["b2"], lp, out_f32=True)


class SublayerConnection(nn.Module):
    """x + dropout(sublayer(norm(x)))  (mtn.py:116-127).  ``attend`` / ``feed`` are the fused fast paths the
    DecoderLayer uses; ``forward(x, sublayer)`` keeps the reference's generic callable form."""

    def __init__(self, size: int, dropout: float):
        super().__init__()
        self.norm = LayerNorm(size)
        self.p = dropout
        self.salt = 0
        self._owner = None          # EncoderDecoder (for compute dtype / dropout seed); set at flatten
        self._ln_fold = None        # (id(module), float [2K] view): fold vectors of the Linear behind this LayerNorm (EncoderDecoder._build_ln_fold)

    def _fold_for(self, module):
        f = self._ln_fold
        if f is None or f[0] != id(module):
            return None
        o = self._owner
        if o is not None and o._ln_fold_stale:      # the weights changed since the vectors were computed (a caller that skips encode())
            o.fold_layer_norms()
        return f[1]

    def _ctx(self):
        o = self._owner
        if o is None:
            return torch.bfloat16, None, None
        return o.compute_dtype, (o._seed if self.training else None), o._queue

    def attend(self, x, attn: MultiHeadedAttention, mem, mask):
        lp, seed, queue = self._ctx()
        f = attn.fused()
        g = None
        if f["grads"] is not None and self.norm._grads is not None:
            g = dict(f["grads"], ln_a=self.norm._grads[0], ln_b=self.norm._grads[1])
        cfg = ops.MhaConfig(heads=attn.h, eps=self.norm.eps,
                            p_attn=attn.p if self.training else 0.0, p_out=self.p if self.training else 0.0,
                            salt=self.salt, seed=seed, lp_dtype=lp, w_qkv_lp=f["w_qkv_lp"], w_o_lp=f["w_o_lp"],
                            w_qkv_lpT=f.get("w_qkv_lpT"), w_o_lpT=f.get("w_o_lpT"), grads=g,
                            queue=queue if g is not None else None, ln_fold=self._fold_for(attn))
        mem_lp = getattr(mem, "_mtn_lp", None) if mem is not None else None
        if mem_lp is not None and mem_lp.dtype != lp:
            mem_lp = None
        return ops.MHASublayerFn.apply(x, mem, mem_lp, mask, self.norm.a_2, self.norm.b_2,
                                       f["w_qkv"], f["b_qkv"], f["w_o"], f["b_o"], cfg)

    def member(self, sublayer, mem=None, mask=None):
        """This sublayer as a member of a lockstep group (ops.SublayerGroupFn); needs the flattened model."""
        lp, seed, queue = self._ctx()
        f = sublayer.fused()
        g = dict(f["grads"], ln_a=self.norm._grads[0], ln_b=self.norm._grads[1])
        if isinstance(sublayer, MultiHeadedAttention):
            cfg = ops.MhaConfig(heads=sublayer.h, eps=self.norm.eps, p_attn=sublayer.p if self.training else 0.0,
                                p_out=self.p if self.training else 0.0, salt=self.salt, seed=seed, lp_dtype=lp,
                                w_qkv_lp=f["w_qkv_lp"], w_o_lp=f["w_o_lp"], w_qkv_lpT=f.get("w_qkv_lpT"), w_o_lpT=f.get("w_o_lpT"),
                                grads=g, queue=queue, ln_fold=self._fold_for(sublayer))
            mem_lp = getattr(mem, "_mtn_lp", None) if mem is not None else None
            if mem_lp is not None and mem_lp.dtype != lp:
                mem_lp = None
            return ops.GroupMember("mha", cfg, (self.norm.a_2, self.norm.b_2, f["b_qkv"], f["b_o"]), mask, mem_lp)
        cfg = ops.FfnConfig(eps=self.norm.eps, p_hidden=sublayer.p if self.training else 0.0, p_out=self.p if self.training else 0.0,
                            salt=self.salt, seed=seed, lp_dtype=lp, w1_lp=f["w1_lp"], w2_lp=f["w2_lp"], w1_lpT=f.get("w1_lpT"),
                            w2_lpT=f.get("w2_lpT"), grads=g, queue=queue, ln_fold=self._fold_for(sublayer))
        return ops.GroupMember("ffn", cfg, (self.norm.a_2, self.norm.b_2, f["b1"], f["b2"]))

    def feed(self, x, ff: PositionwiseFeedForward):
        lp, seed, queue = self._ctx()
        f = ff.fused()
        g = None
        if f["grads"] is not None and self.norm._grads is not None:
            g = dict(f["grads"], ln_a=self.norm._grads[0], ln_b=self.norm._grads[1])
        cfg = ops.FfnConfig(eps=self.norm.eps, p_hidden=ff.p if self.training else 0.0,
                            p_out=self.p if self.training else 0.0, salt=self.salt, seed=seed, lp_dtype=lp,
                            w1_lp=f["w1_lp"], w2_lp=f["w2_lp"], w1_lpT=f.get("w1_lpT"), w2_lpT=f.get("w2_lpT"), grads=g,
                            queue=queue if g is not None else None, ln_fold=self._fold_for(ff))
        return ops.FFNSublayerFn.apply(x, self.norm.a_2, self.norm.b_2, f["w1"], f["b1"], f["w2"], f["b2"], cfg)

    def forward(self, x, sublayer, mem=None, mask=None, self_attention=False):
        """``sublayer`` is a MultiHeadedAttention (with ``mem``/``mask``; ``self_attention`` -> key=value=norm(x)), a
        PositionwiseFeedForward (as the reference passes it, mtn.py:213,218) or any callable (generic, un-fused)."""
        if isinstance(sublayer, MultiHeadedAttention):
            return self.attend(x, sublayer, None if self_attention else mem, mask)
        if isinstance(sublayer, PositionwiseFeedForward):
            return self.feed(x, sublayer)
        y = sublayer(self.norm(x))
        if self.training and self.p > 0:
            y = F.dropout(y, self.p)
        return x + y


class DecoderLayer(nn.Module):
    """One MTN block (mtn.py:166-218): 4 text attentions, per-modality query-aware auto-encoder
    (self-attn, attend-to-video, FFN) + attend-to-auto-encoder, final FFN: 5+4F fused sublayers."""

    def __init__(self, size, self_attn, cap_attn, his_attn, q_attn, auto_encoder_self_attn, auto_encoder_vid_attn,
                 auto_encoder_attn, feed_forward, auto_encoder_feed_forward, dropout):
        super().__init__()
        self.size = size
        self.self_attn = self_attn
        self.src_attn = q_attn
        self.feed_forward = feed_forward
        self.his_attn = his_attn
        self.cap_attn = cap_attn
        self.auto_encoder_attn = auto_encoder_attn
        self.auto_encoder_self_attn = auto_encoder_self_attn
        self.auto_encoder_vid_attn = auto_encoder_vid_attn
        self.auto_encoder_feed_forward = auto_encoder_feed_forward
        self.sublayer = nn.ModuleList([SublayerConnection(size, dropout) for _ in range(5 + 4 * len(auto_encoder_vid_attn))])

    @staticmethod
    def _run_group(items, raw_memory_outputs=False):
        """items: [(sublayer_connection, module, mem, mask, input)] that do not depend on each other -> their outputs,
        computed by one lockstep group (ops.SublayerGroupFn: shared launches).  raw_memory_outputs: the feed-forward outputs are
        attended as un-projected memory later (the auto-encoder streams, mtn.py:215): they also leave in the compute dtype."""
        members, tensors = [], []
        for it in items:
            sc, mod, mem, mask, inp = it[:5]      # (a sixth element names the next reader of the output: LayerNorm FORWARD by linearity used
            mb = sc.member(mod, mem, mask)        #  it — measured in round 4, no net gain, removed in round 5: profiles/r04_k_ln_forward_by_linearity.txt)
            mb.want_lp = raw_memory_outputs and mb.kind == "ffn"
            kv = getattr(sc, "_kv_ready", None)        # K|V of a constant memory projected ahead of the layer loop
            if kv is not None and mb.kind == "mha" and mem is not None and kv.size(0) == mem.size(0) * mem.size(1):
                mb.kv_ready = kv
            if torch.is_grad_enabled() and _HANDOFF:   # gradient hand-off along the chain (ops.GroupMember.holder / .feeds)
                mb.feeds = getattr(inp, "_mtn_next", None)
                mb.holder = dict(p=mb.cfg.p_out, salt=mb.cfg.salt * 4 + 1, seed=mb.cfg.seed, lp=mb.cfg.lp_dtype, dyl=None, dx_ptr=None, ver=None, shape=None)
            members.append(mb)
            tensors += [inp, mem if isinstance(mod, MultiHeadedAttention) else None]
        # attention members first, then FFN members (the C side takes two arrays)
        order = sorted(range(len(members)), key=lambda k: members[k].kind != "mha")
        outs = ops.SublayerGroupFn.apply([members[k] for k in order], *[t for k in order for t in tensors[2 * k:2 * k + 2]])
        res = [None] * len(members)
        for pos, k in enumerate(order):
            res[k] = outs[pos]
            if members[k].holder is not None:
                outs[pos]._mtn_next = members[k].holder
            if members[k].out_lp is not None:
                outs[pos]._mtn_lp = members[k].out_lp
        return res

    def _plan(self, cap_memory, cap_mask, his_memory, his_mask, q_memory, q_mask, tgt_mask, vid_fts, vid_mask, ae_features):
        """Sublayer schedule of mtn.py:183-213: (the four text attentions of x, per-modality auto-encoder chains, the
        auto-encoder seed when the caller passes none, the auto-encoder mask)."""
        sl = self.sublayer
        if ae_features in ("caption", "summary"):
            text = [(sl[0], self.self_attn, None, tgt_mask), (sl[1], self.his_attn, his_memory, his_mask),
                    (sl[2], self.src_attn, q_memory, q_mask), (sl[3], self.cap_attn, cap_memory, cap_mask)]
            seed_ae, ae_mask = cap_memory, cap_mask
        elif ae_features == "query":
            text = [(sl[0], self.self_attn, None, tgt_mask), (sl[1], self.his_attn, his_memory, his_mask),
                    (sl[2], self.cap_attn, cap_memory, cap_mask), (sl[3], self.src_attn, q_memory, q_mask)]
            seed_ae, ae_mask = q_memory, q_mask
        else:
            raise ValueError("auto_encoder_ft must be 'query', 'caption' or 'summary' (reference: mtn.py:187-202)")
        nF = len(vid_fts)
        chains = [[(sl[4 + 4 * i], self.auto_encoder_self_attn[i], None, ae_mask),
                   (sl[5 + 4 * i], self.auto_encoder_vid_attn[i], vid_fts[i], vid_mask[i]),
                   (sl[6 + 4 * i], self.auto_encoder_feed_forward[i], None, None)] for i in range(nF)]
        return text, chains, seed_ae, ae_mask

    def _forward_lockstep(self, x, cap_memory, cap_mask, his_memory, his_mask, q_memory, q_mask, tgt_mask, vid_fts, vid_mask,
                          ae_fts, ae_features):
        """Same arithmetic as forward(), scheduled as lockstep groups: the first three text attentions of x run in the same
        launches as the three sublayers of each auto-encoder chain (they do not depend on each other, mtn.py:183-213)."""
        sl, run = self.sublayer, self._run_group
        text, chains, seed_ae, ae_mask = self._plan(cap_memory, cap_mask, his_memory, his_mask, q_memory, q_mask, tgt_mask,
                                                    vid_fts, vid_mask, ae_features)
        if ae_fts is None:
            ae_fts = seed_ae
        nF = len(vid_fts)
        aes = [ae_fts[i] if isinstance(ae_fts, (list, tuple)) else ae_fts for i in range(nF)]
        # who reads each output next (its LayerNorm's gains ride into the producer's epilogue): the chain's next sublayer, across the
        # layer boundary the next layer's first sublayer of the same stream
        nl = getattr(self, "_next_layer", None)
        nl_ok = nl is not None and len(nl.sublayer) == len(sl)
        ffn_sc = sl[4 + 4 * nF]
        for j in range(3):
            ae_next = [chains[i][j + 1][0] if j < 2 else (nl.sublayer[4 + 4 * i] if nl_ok else None) for i in range(nF)]
            items = [text[j] + (x, text[j + 1][0])] + [chains[i][j] + (aes[i], ae_next[i]) for i in range(nF)]
            outs = run(items, raw_memory_outputs=(j == 2))
            x, aes = outs[0], list(outs[1:])
        x = run([text[3] + (x, sl[7] if nF > 0 else ffn_sc)])[0]
        for i in range(nF):
            x = run([(sl[7 + 4 * i], self.auto_encoder_attn[i], aes[i], ae_mask, x, sl[7 + 4 * (i + 1)] if i + 1 < nF else ffn_sc)])[0]
        x = run([(ffn_sc, self.feed_forward, None, None, x, nl.sublayer[0] if nl_ok else None)])[0]
        return x, aes

    def forward_ae_chains(self, cap_memory, cap_mask, q_memory, q_mask, vid_fts, vid_mask, ae_fts, ae_features):
        """Only the query-aware auto-encoder chains of this layer (self-attn -> attend-to-video -> FFN per modality,
        mtn.py:204-209).  They do not depend on the target stream, so the decode path runs them once per dialogue."""
        _, chains, seed_ae, _ = self._plan(cap_memory, cap_mask, None, None, q_memory, q_mask, None, vid_fts, vid_mask, ae_features)
        if ae_fts is None:
            ae_fts = seed_ae
        nF = len(vid_fts)
        aes = [ae_fts[i] if isinstance(ae_fts, (list, tuple)) else ae_fts for i in range(nF)]
        for j in range(3):
            aes = self._run_group([chains[i][j] + (aes[i],) for i in range(nF)], raw_memory_outputs=(j == 2))
        return aes

    def forward_target(self, x, cap_memory, cap_mask, his_memory, his_mask, q_memory, q_mask, tgt_mask, aes, ae_features):
        """Only the target stream of this layer, given the layer's auto-encoder outputs ``aes`` (forward_ae_chains)."""
        sl = self.sublayer
        nF = len(aes)
        text, _, _, ae_mask = self._plan(cap_memory, cap_mask, his_memory, his_mask, q_memory, q_mask, tgt_mask, [None] * nF,
                                         [None] * nF, ae_features)
        for j in range(4):
            x = self._run_group([text[j] + (x,)])[0]
        for i in range(nF):
            x = self._run_group([(sl[7 + 4 * i], self.auto_encoder_attn[i], aes[i], ae_mask, x)])[0]
        return self._run_group([(sl[4 + 4 * nF], self.feed_forward, None, None, x)])[0]

    def self_kv_of_new_rows(self, x):
        """K|V projection of the target self-attention (sublayer 0) for NEW target rows x (W, n, d): LayerNorm(x) w_k|w_v^T + b
        -> (W, n, 2d) in the compute dtype (decode with a prefix K/V cache: mtn.py:183 applied to one position)."""
        sc, attn = self.sublayer[0], self.self_attn
        lp = sc._ctx()[0]
        f = attn.fused()
        d = x.size(-1)
        _, xn_lp = ops.layer_norm(x, sc.norm.a_2, sc.norm.b_2, sc.norm.eps, lp)
        w = f["w_qkv_lp"] if f["w_qkv_lp"] is not None else ops.cast_to_lp(f["w_qkv"], lp)
        rows = x.numel() // d
        out = torch.empty(rows, 2 * d, device=x.device, dtype=lp)
        pr = ops.L.GemmProblem()
        pr.A, pr.B, pr.lda, pr.ldb, pr.M, pr.N, pr.K = xn_lp.data_ptr(), w.data_ptr() + d * d * w.element_size(), d, d, rows, 2 * d, d
        pr.bias, pr.gate_scale, pr.out_lp, pr.ldc = f["b_qkv"].data_ptr() + d * f["b_qkv"].element_size(), 1.0, out.data_ptr(), 2 * d
        ops.gemm(ops.L.dtype_code(lp), [pr])
        return out.view(x.size(0), x.size(1), 2 * d)

    def forward_target_cached(self, x, cap_memory, cap_mask, his_memory, his_mask, q_memory, q_mask, aes, ae_features,
                              self_kv, self_mem, self_mask):
        """forward_target for the NEWEST target position only (x: (W, 1, d)) against a prefix K/V cache of this layer's target
        self-attention: ``self_kv`` (W * L, 2d) already holds K|V of positions 0..l (the new row included), ``self_mask``
        (W, 1, L) enables keys 0..l, ``self_mem`` is a (W, L, d) placeholder that only carries the shape.  The self-attention
        of mtn.py:183 on the last row is then a cross-attention of that row over the cache — same weights, same arithmetic."""
        sl = self.sublayer
        nF = len(aes)
        text, _, _, ae_mask = self._plan(cap_memory, cap_mask, his_memory, his_mask, q_memory, q_mask, None, [None] * nF,
                                         [None] * nF, ae_features)
        prev = getattr(sl[0], "_kv_ready", None)
        object.__setattr__(sl[0], "_kv_ready", self_kv)
        try:
            x = self._run_group([(sl[0], self.self_attn, self_mem, self_mask, x)])[0]
        finally:
            object.__setattr__(sl[0], "_kv_ready", prev)
        for j in range(1, 4):
            x = self._run_group([text[j] + (x,)])[0]
        for i in range(nF):
            x = self._run_group([(sl[7 + 4 * i], self.auto_encoder_attn[i], aes[i], ae_mask, x)])[0]
        return self._run_group([(sl[4 + 4 * nF], self.feed_forward, None, None, x)])[0]

    def forward(self, x, cap_memory, cap_mask, his_memory, his_mask, q_memory, q_mask, tgt_mask, vid_fts, vid_mask,
                ae_fts, ae_features):
        owner = getattr(self, "_owner", None)
        if (owner is not None and owner.lockstep and x.is_cuda and owner._flat is not None and len(vid_fts) + 1 <= 4
                and not self._forward_hooks_on_sublayers()):
            return self._forward_lockstep(x, cap_memory, cap_mask, his_memory, his_mask, q_memory, q_mask, tgt_mask, vid_fts,
                                          vid_mask, ae_fts, ae_features)
        sl = self.sublayer
        x = sl[0](x, self.self_attn, None, tgt_mask, True)
        x = sl[1](x, self.his_attn, his_memory, his_mask)
        if ae_features in ("caption", "summary"):
            x = sl[2](x, self.src_attn, q_memory, q_mask)
            x = sl[3](x, self.cap_attn, cap_memory, cap_mask)
            if ae_fts is None:
                ae_fts = cap_memory
            ae_mask = cap_mask
        elif ae_features == "query":
            x = sl[2](x, self.cap_attn, cap_memory, cap_mask)
            x = sl[3](x, self.src_attn, q_memory, q_mask)
            if ae_fts is None:
                ae_fts = q_memory
            ae_mask = q_mask
        else:
            raise ValueError("auto_encoder_ft must be 'query', 'caption' or 'summary' (reference: mtn.py:187-202)")
        # The query-aware auto-encoder chains (self-attn -> attend-to-video -> FFN, one per modality) depend only on the previous
        # layer's auto-encoder outputs, not on x: the lockstep schedule above runs them in the same launches as x's text attentions
        # (per-chain HIP streams were the round-1 form: hipGraph serialised them); this is the one-sublayer-at-a-time form.
        nF = len(vid_fts)
        aes = []
        for i, vid_ft in enumerate(vid_fts):
            ae = ae_fts[i] if isinstance(ae_fts, (list, tuple)) else ae_fts
            k = 4 + 4 * i
            ae = sl[k](ae, self.auto_encoder_self_attn[i], None, ae_mask, True)
            ae = sl[k + 1](ae, self.auto_encoder_vid_attn[i], vid_ft, vid_mask[i])
            ae = sl[k + 2](ae, self.auto_encoder_feed_forward[i])
            aes.append(ae)
        out_ae = []
        for i in range(nF):
            x = sl[4 + 4 * i + 3](x, self.auto_encoder_attn[i], aes[i], ae_mask)
            out_ae.append(aes[i])
        k = 4 + 4 * nF
        return sl[k](x, self.feed_forward), out_ae


    def _forward_hooks_on_sublayers(self):
        """Per-sublayer forward hooks (used to tap intermediate outputs) need the one-sublayer-at-a-time schedule."""
        return any(len(s._forward_hooks) > 0 for s in self.sublayer)


class Decoder(nn.Module):
    """N DecoderLayers + final LayerNorms (mtn.py:149-164)."""

    def __init__(self, make_layer, N: int, ft_sizes: Sequence[int]):
        super().__init__()
        self.layers = nn.ModuleList([make_layer() for _ in range(N)])
        size = self.layers[0].size
        self.norm = LayerNorm(size)
        self.ae_norm = nn.ModuleList([LayerNorm(size) for _ in ft_sizes])

    def forward(self, vid_ft, vid_mask, x, his_memory, his_mask, cap_memory, cap_mask, query_memory, query_mask, tgt_mask,
                auto_encoded_ft, auto_encoded_features):
        ops.prepare_masks(tgt_mask, his_mask, cap_mask, query_mask, vid_mask)
        for layer in self.layers:
            x, auto_encoded_ft = layer(x, cap_memory, cap_mask, his_memory, his_mask, query_memory, query_mask, tgt_mask,
                                       vid_ft, vid_mask, auto_encoded_ft, auto_encoded_features)
        if x.is_cuda and len(auto_encoded_ft) == len(self.ae_norm) and all(t.size(-1) == x.size(-1) for t in auto_encoded_ft):
            ys = ops.layer_norm_group([x] + list(auto_encoded_ft), [self.norm] + list(self.ae_norm))      # one launch each way
            return ys[0], ys[1:]
        return self.norm(x), [self.ae_norm[i](ft) for i, ft in enumerate(auto_encoded_ft)]


class Encoder(nn.Module):
    """Bank of LayerNorms applied to the (possibly nested) input streams in order (mtn.py:75-101).
    Each output carries its compute-dtype copy (``._mtn_lp``) for use as attention memory."""

    def __init__(self, size: int, nb_layers: int):
        super().__init__()
        self.norm = nn.ModuleList([LayerNorm(size) for _ in range(nb_layers)])
        self.nb_layers = nb_layers

    def _apply_norm(self, i, x):
        y, y_lp = self.norm[i].forward_lp(x)
        y._mtn_lp = y_lp
        return y

    def forward(self, *seqs):
        out, i = [], 0
        for s in seqs:
            if i >= self.nb_layers:
                break
            if isinstance(s, (list, tuple)):
                ys = []
                for t in s:
                    ys.append(self._apply_norm(i, t)); i += 1
                out.append(ys)
            else:
                out.append(self._apply_norm(i, s)); i += 1
        return out


class Embeddings(nn.Module):
    def __init__(self, d_model: int, vocab: int):
        super().__init__()
        self.lut = nn.Embedding(vocab, d_model)
        self.d_model = d_model

    def forward(self, x):                      # mtn.py:288-289
        return self.lut(x) * math.sqrt(self.d_model)


class PositionalEncoding(nn.Module):
    def __init__(self, d_model: int, dropout: float, max_len: int = 5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pos = torch.arange(0.0, max_len).unsqueeze(1)
        div = torch.exp(torch.arange(0.0, d_model, 2) * -(math.log(10000.0) / d_model))
        pe = torch.zeros(max_len, d_model)
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        self.register_buffer("pe", pe.unsqueeze(0))

    def forward(self, x):                      # mtn.py:307-309
        return self.dropout(x + self.pe[:, : x.size(1)])


class Generator(nn.Module):
    def __init__(self, d_model: int, vocab: int):
        super().__init__()
        self.proj = nn.Linear(d_model, vocab)
        self._fused = None          # compute-dtype weight / flat-gradient views (fused loss head), set at flatten

    def forward(self, x):                      # mtn.py:68-69
        f = self._fused
        differentiated = torch.is_grad_enabled() and (x.requires_grad or self.proj.weight.requires_grad)
        if f is not None and x.is_cuda and not self.training and not differentiated and f["w_lp"].device == x.device and x.size(-1) % 8 == 0:
            # inference — eval() mode and nothing to differentiate (decode, validation): the library's GEMM + row log-softmax
            # (ops.generator_log_probs), no vendor BLAS / softmax kernels in a decode step.  NUMERICS: x and the weight enter the GEMM in
            # the compute dtype (bf16: 8 mantissa bits; fp32 accumulation, fp32 log-softmax), where the reference's generator is fp32
            # throughout — log-probabilities differ by ~1e-3 relative (tests/test_decode_gpu.py pins the bound).  The path depends on
            # the module's MODE, not on the grad mode alone: train() always takes the composed fp32 form below, and so does a caller
            # that differentiates through an eval() module.
            f["prepare"]()                     # the compute-dtype weight copy follows the fp32 master (load_state_dict, torch optimisers)
            return ops.generator_log_probs(x, f["w_lp"], f["bias"])
        return F.log_softmax(self.proj(x), dim=-1)


# ------------------------------------------------------------------------------------------ top level
# which transposed weight copies (besides W_o^T) the flat layout keeps: see EncoderDecoder._flatten


class EncoderDecoder(nn.Module):
    def __init__(self, query_encoder, his_encoder, cap_encoder, vid_encoder, decoder, query_embed, his_embed, cap_embed,
                 tgt_embed, generator, diff_encoder=False, auto_encoder_embed=None, auto_encoder_ft=None,
                 auto_encoder_generator=None, compute_dtype=torch.bfloat16):
        super().__init__()
        self.query_encoder = query_encoder
        self.his_encoder = his_encoder
        self.cap_encoder = cap_encoder
        self.vid_encoder = vid_encoder
        self.decoder = decoder
        self.query_embed = query_embed
        self.his_embed = his_embed
        self.cap_embed = cap_embed
        self.tgt_embed = tgt_embed
        self.generator = generator
        self.diff_encoder = diff_encoder
        self.auto_encoder_embed = auto_encoder_embed
        self.auto_encoder_ft = auto_encoder_ft
        self.auto_encoder_generator = auto_encoder_generator
        self.compute_dtype = compute_dtype
        self._flat = self._flat_lp = self._flat_grad = None
        self._flat_version = -1
        self._master_sync = None               # data parallel, compute-dtype gather: brings stale fp32 masters back (collective)
        self._glue_numel = 0
        self._layer_slices = []
        self._seed = None
        self._queue = ops.ParamGradQueue()     # dW/db/LN-parameter work batched at the end of backward
        self.lockstep = True                   # independent sublayers of a layer share launches (ops.SublayerGroupFn)
        self.fused_embed = True                # Embeddings + PositionalEncoding + Encoder LayerNorm in one grouped launch
        self.hoist_kv = True                   # K|V of the constant memories projected ahead of the layer loop
        self._embed_calls = 0
        self._ln_fold_stale = True
        self._ln_fold_buf = self._ln_fold_table = None

    # ---- flat parameter storage ------------------------------------------------------------------
    def _ordered_params(self):
        """(glue params, path params) in flat-buffer order.  q/k/v weights (then biases) of every attention adjacent."""
        seen, glue, path = set(), [], []

        def add(lst, p):
            if p is not None and id(p) not in seen:
                seen.add(id(p)); lst.append(p)

        # Path parameters: all the VECTORS first (Encoder LayerNorm bank, then per layer the biases and the sublayers' LayerNorm gains
        # / biases, then the decoder's final LayerNorms), then per layer its weight MATRICES.  The kernels read the matrices through
        # their compute-dtype copy only, so under data parallelism a layer's slice — pure matrices — travels in the compute dtype
        # (dp.ShardedOptimizerSync, `mat_hi`), while the vectors (read in fp32, 0.25 % of the parameters) sit next to the glue
        # parameters and are exchanged with them in fp32, in the step's last slice.
        def layer_modules(layer):
            mods = [layer.self_attn, layer.his_attn, layer.cap_attn, layer.src_attn]
            for i in range(len(layer.auto_encoder_vid_attn)):
                mods += [layer.auto_encoder_self_attn[i], layer.auto_encoder_vid_attn[i], layer.auto_encoder_feed_forward[i], layer.auto_encoder_attn[i]]
            return mods + [layer.feed_forward]

        for n in self.query_encoder.norm:
            add(path, n.a_2); add(path, n.b_2)
        for layer in self.decoder.layers:
            for m in layer_modules(layer):
                if isinstance(m, MultiHeadedAttention):
                    for i in range(3):
                        add(path, m.linears[i].bias)
                    add(path, m.linears[3].bias)
                else:
                    add(path, m.w_1.bias); add(path, m.w_2.bias)
            for sc in layer.sublayer:
                add(path, sc.norm.a_2); add(path, sc.norm.b_2)
        for n in [self.decoder.norm] + list(self.decoder.ae_norm):
            add(path, n.a_2); add(path, n.b_2)
        layer_marks = []
        for layer in self.decoder.layers:
            start = len(path)
            for m in layer_modules(layer):
                if isinstance(m, MultiHeadedAttention):
                    for i in range(3):
                        add(path, m.linears[i].weight)
                    add(path, m.linears[3].weight)
                else:
                    add(path, m.w_1.weight); add(path, m.w_2.weight)
            layer_marks.append((start, len(path)))
        for p in self.parameters():            # everything else = glue (embeddings, feature Linear, generator)
            if id(p) not in seen:
                add(glue, p)
        return glue, path, layer_marks

    def _flatten(self):
        """(Re)build the flat fp32 / compute-dtype / gradient buffers on the parameters' current device."""
        glue, path, layer_marks = self._ordered_params()
        params = glue + path
        dev = params[0].device
        pad = lambda n: (n + 7) // 8 * 8
        offs, total = [], 0
        for p in params:
            offs.append(total); total += pad(p.numel())
        flat = torch.zeros(total, device=dev, dtype=torch.float32)
        grad = torch.zeros(total, device=dev, dtype=torch.float32)
        for p, o in zip(params, offs):
            flat[o:o + p.numel()].copy_(p.data.reshape(-1).float())
            p.data = flat[o:o + p.numel()].view(p.shape)
            p.grad = grad[o:o + p.numel()].view(p.shape)
        self._flat, self._flat_grad = flat, grad
        self._flat_params = params
        self._flat_offsets = offs
        self._glue_numel = offs[len(glue)] if path else total
        path_off = {id(p): o for p, o in zip(params, offs)}
        self._layer_slices = [(path_off[id(path[s])], path_off[id(path[e - 1])] + pad(path[e - 1].numel())) for s, e in layer_marks]
        lp = self.compute_dtype
        self._flat_lp = torch.empty(total, device=dev, dtype=lp) if lp != torch.float32 else flat
        self._flat_version = -1
        # the transposed copies live in a compact buffer of their own (allocated below, once the kept copies are known):
        # _t_off maps a weight's offset in the flat buffers to its copy's offset there
        want_t = dev.type == "cuda"
        self._flat_lpT = None
        self._t_off = {}
        t_views = []         # (dict, key, flat offset, rows, cols) of every kept transposed copy
        # (offset, rows, cols) of every 2-D path weight that gets a transposed copy.  Only W_o does (the fused head backward
        # reads W_o^T rows, csrc/fused_bwd.hip): every other dX = dY W runs on the LDS-DMA GEMM with W as it lies (b_trans = 1:
        # [k][n] tiles + transposing LDS reads, csrc/gemm.hip), so the optimiser epilogue writes no transposed copy for them —
        # those scattered 128-byte runs were 16 % of the launch for 7 % of its bytes (profiles/r03_tt_ablation.txt).
        # (rounds 1-2 kept one for every 2-D weight; the A/B switch is gone with round 5: profiles/r03_m_keep_wt_ab.txt)
        keep = set()
        tdescs = []
        fusable = []         # ... of those, the weights whose gradient is ONE deferred GEMM (optimiser epilogue)
        optional = set()     # ... offsets of fusable weights that may legitimately get no deferred GEMM in a step
        if dev.type == "cuda":
            if self._seed is None or self._seed.device != dev:
                self._seed = torch.full((1,), torch.initial_seed() & 0x7FFFFFFFFFFFFFFF, device=dev, dtype=torch.int64)

        def views(p):
            o = path_off[id(p)]
            return flat[o:o + p.numel()].view(p.shape), self._flat_lp[o:o + p.numel()].view(p.shape), grad[o:o + p.numel()].view(p.shape)

        def span(ps, shape):
            o = path_off[id(ps[0])]
            n = sum(p.numel() for p in ps)
            assert all(path_off[id(ps[i + 1])] == path_off[id(ps[i])] + ps[i].numel() for i in range(len(ps) - 1)), "fused span not contiguous"
            return flat[o:o + n].view(shape), self._flat_lp[o:o + n].view(shape), grad[o:o + n].view(shape)

        for m in self.modules():
            if isinstance(m, MultiHeadedAttention) and id(m.linears[0].weight) in path_off:
                d = m.linears[0].weight.size(1)
                w, wl, gw = span([m.linears[i].weight for i in range(3)], (3 * d, d))
                b, _, gb = span([m.linears[i].bias for i in range(3)], (3 * d,))
                wo, wol, gwo = views(m.linears[3].weight)
                bo, _, gbo = views(m.linears[3].bias)
                m._fused = dict(w_qkv=w, b_qkv=b, w_o=wo, b_o=bo, w_qkv_lp=wl, w_o_lp=wol, lp_dtype=lp,
                                grads=dict(w_qkv=gw, b_qkv=gb, w_o=gwo, b_o=gbo))
                if want_t:
                    oq, oo = path_off[id(m.linears[0].weight)], path_off[id(m.linears[3].weight)]
                    t_views.append((m._fused, "w_o_lpT", oo, d, d))
                    tdescs += [(oo, d, d)]
                    if "qkv" in keep:
                        t_views.append((m._fused, "w_qkv_lpT", oq, 3 * d, d))
                        tdescs += [(oq, 3 * d, d)]
                    fusable += [(oq, 3 * d, d), (oo, d, d)]
            elif isinstance(m, PositionwiseFeedForward) and id(m.w_1.weight) in path_off:
                w1, w1l, g1 = views(m.w_1.weight); b1, _, gb1 = views(m.w_1.bias)
                w2, w2l, g2 = views(m.w_2.weight); b2, _, gb2 = views(m.w_2.bias)
                m._fused = dict(w1=w1, b1=b1, w2=w2, b2=b2, w1_lp=w1l, w2_lp=w2l, lp_dtype=lp,
                                grads=dict(w1=g1, b1=gb1, w2=g2, b2=gb2))
                if want_t:
                    o1, o2 = path_off[id(m.w_1.weight)], path_off[id(m.w_2.weight)]
                    ffd, dm = m.w_1.weight.shape
                    if "w1" in keep:
                        t_views.append((m._fused, "w1_lpT", o1, ffd, dm))
                        tdescs += [(o1, ffd, dm)]
                    if "w2" in keep:
                        t_views.append((m._fused, "w2_lpT", o2, dm, ffd))
                        tdescs += [(o2, dm, ffd)]
                    fusable += [(o1, ffd, dm), (o2, dm, ffd)]
            elif isinstance(m, Generator):
                o_w, o_b = path_off[id(m.proj.weight)], path_off[id(m.proj.bias)]
                nw, nb = m.proj.weight.numel(), m.proj.bias.numel()
                m._fused = dict(w_lp=self._flat_lp[o_w:o_w + nw].view(m.proj.weight.shape), bias=flat[o_b:o_b + nb],
                                grad_w=grad[o_w:o_w + nw].view(m.proj.weight.shape), grad_b=grad[o_b:o_b + nb], lp_dtype=lp, w_lpT=None)
                vocab, dm = m.proj.weight.shape
                if want_t and vocab % 8 == 0 and not any(t[0] == o_w for t in fusable):
                    if "gen" in keep:
                        t_views.append((m._fused, "w_lpT", o_w, vocab, dm))
                        tdescs.append((o_w, vocab, dm))
                    fusable.append((o_w, vocab, dm))
                    optional.add(o_w)            # its dW reaches the queue only through the fused loss head
                m._fused["queue"] = self._queue
                m._fused["prepare"] = self.prepare
            elif isinstance(m, LayerNorm) and id(m.a_2) in path_off:
                m._grads = (views(m.a_2)[2], views(m.b_2)[2])
                m._lp_dtype = lp
                m._queue = self._queue
        # the feature Linears (glue parameters, mtn.py:378): their dW is ONE deferred GEMM of the table launch too (ops.FeatureEncodeFn)
        # — optional, because feature widths the GEMM cannot take go through PyTorch's Linear instead
        if dev.type == "cuda":
            for seq in self.vid_encoder:
                lin = seq[0]
                o_w = path_off[id(lin.weight)]
                dm, fs = lin.weight.shape
                if fs % 8 == 0 and dm % 8 == 0 and not any(t[0] == o_w for t in fusable):
                    fusable.append((o_w, dm, fs))
                    optional.add(o_w)
        if t_views:
            t_total = 0
            for _, _, o, r, c in t_views:
                self._t_off[o] = t_total
                t_total += pad(r * c)
            self._flat_lpT = torch.zeros(t_total, device=dev, dtype=lp)
            for dct, key, o, r, c in t_views:
                dct[key] = self._flat_lpT[self._t_off[o]:self._t_off[o] + r * c].view(c, r)
        self._tdesc = self._tdesc_table(tdescs)
        # optimiser-epilogue bookkeeping (ops.ParamGradQueue / data_utils.FusedAdam): the fusable weights sorted by offset,
        # the transposed copies that still need the transpose pass, and the rest of the flat buffer as <= 4096-element chunks
        self._fusable = sorted(set(fusable))
        self._fusable_optional = frozenset(optional)
        self._tdescs_all = list(tdescs)
        self._rest_cache = {}
        for n, layer in enumerate(self.decoder.layers):
            object.__setattr__(layer, "_owner", self)
            object.__setattr__(layer, "_next_layer", self.decoder.layers[n + 1] if n + 1 < len(self.decoder.layers) else None)
            layer._layer_index = n
            for k, s in enumerate(layer.sublayer):
                object.__setattr__(s, "_owner", self)      # plain attribute: not a registered submodule
                s.salt = n * 64 + k + 1
        self._build_ln_fold()

    def _build_ln_fold(self):
        """Fold vectors u = W a2, c = b + W b2 of every Linear that follows a sublayer's LayerNorm (include/mtn_hip.h,
        mtn_ln_epilogue): one flat fp32 buffer with a [2K] slot per (sublayer connection, module) pair of the layer schedule
        (mtn.py:183-218), the device-side descriptor table of mtn_ln_fold, and the views the sublayers hand to the kernels.
        bf16 on the GPU only (the kernels that use them are the fused bf16 ones)."""
        self._ln_fold_buf = self._ln_fold_table = None
        self._ln_fold_stale = True
        for layer in self.decoder.layers:
            for sc in layer.sublayer:
                sc._ln_fold = None
        dev = self._flat.device
        if dev.type != "cuda" or self.compute_dtype != torch.bfloat16 or self.auto_encoder_ft not in ("caption", "summary", "query"):
            return
        pairs = []                   # (sublayer connection, module, W [K, d] compute dtype, bias [K])
        for layer in self.decoder.layers:
            sl, nF = layer.sublayer, len(layer.auto_encoder_vid_attn)
            if len(sl) != 5 + 4 * nF:
                continue
            text = [layer.self_attn, layer.his_attn] + ([layer.src_attn, layer.cap_attn] if self.auto_encoder_ft in ("caption", "summary")
                                                        else [layer.cap_attn, layer.src_attn])
            sched = [(sl[j], m, j == 0) for j, m in enumerate(text)]
            for i in range(nF):
                sched += [(sl[4 + 4 * i], layer.auto_encoder_self_attn[i], True), (sl[5 + 4 * i], layer.auto_encoder_vid_attn[i], False),
                          (sl[6 + 4 * i], layer.auto_encoder_feed_forward[i], None), (sl[7 + 4 * i], layer.auto_encoder_attn[i], False)]
            sched.append((sl[4 + 4 * nF], layer.feed_forward, None))
            for sc, mod, self_attn in sched:
                f = mod._fused
                if f is None or sc.norm._grads is None:
                    continue
                if isinstance(mod, MultiHeadedAttention):
                    d = f["w_o"].size(0)
                    K = 3 * d if self_attn else d        # a cross-attention's k | v blocks see the memory, not LayerNorm(x)
                    pairs.append((sc, mod, f["w_qkv_lp"][:K], f["b_qkv"][:K]))
                else:
                    pairs.append((sc, mod, f["w1_lp"], f["b1"]))
        if not pairs:
            return
        d = pairs[0][2].size(1)
        if any(w.size(1) != d for _, _, w, _ in pairs) or d % 8 != 0 or d > 2048:
            return
        total = sum(2 * w.size(0) for _, _, w, _ in pairs)
        buf = torch.zeros(total, device=dev, dtype=torch.float32)
        descs = (L.LnFoldDesc * len(pairs))()
        block_desc, off, blocks = [], 0, 0
        for i, (sc, mod, w, b) in enumerate(pairs):
            K = w.size(0)
            view = buf[off:off + 2 * K]
            descs[i].w, descs[i].bias, descs[i].a2, descs[i].b2 = w.data_ptr(), b.data_ptr(), sc.norm.a_2.data_ptr(), sc.norm.b_2.data_ptr()
            descs[i].out, descs[i].K, descs[i].block_start = view.data_ptr(), K, blocks
            nb = (K + 63) // 64
            block_desc += [i] * nb
            blocks += nb
            off += 2 * K
            sc._ln_fold = (id(mod), view)
        table = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
        self._ln_fold_buf = buf
        self._ln_fold_table = (table, torch.tensor(block_desc, dtype=torch.int32).to(dev), blocks, d)

    def fold_layer_norms(self):
        """Recompute the fold vectors from the current weights (one launch, mtn_ln_fold): once per forward that will be
        differentiated — the weights change every step."""
        t = self._ln_fold_table
        if t is not None:
            L.check(L.load().mtn_ln_fold(L.MTN_BF16, t[0].data_ptr(), t[1].data_ptr(), t[2], t[3], L.stream_ptr()))
        self._ln_fold_stale = False

    def state_dict(self, *a, **kw):
        """The reference's key schema.  Data parallel with the compute-dtype gather: a rank holds current fp32
        masters of its own shards only, so the masters are gathered first — a COLLECTIVE: every rank calls state_dict(), rank 0
        writes the file."""
        if self._master_sync is not None:
            self._master_sync()
        return super().state_dict(*a, **kw)

    def _apply(self, fn, *a, **kw):             # .cuda()/.to(): parameters are re-created -> re-flatten
        super()._apply(fn, *a, **kw)
        self._flat = None
        return self

    def set_compute_dtype(self, dtype):
        self.compute_dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}.get(dtype, dtype)
        self._flat = None
        return self

    def prepare(self):
        """Make flat buffers current: flatten if needed, refresh the compute-dtype weight copy if the fp32
        master was modified by anything other than the fused optimiser (load_state_dict, a torch optimiser)."""
        if self._flat is None:
            self._flatten()
        ver = sum(p._version for p in self._flat_params)     # `.data = view` gives every parameter its own counter
        if self._flat_lp is not self._flat and ver != self._flat_version and self._flat_version != -1 and self._master_sync is not None:
            # something other than the fused optimiser touched a parameter while (data parallel, compute-dtype gather) this rank's
            # fp32 masters of the other ranks' shards are stale: recasting them would revert (world-1)/world of every matrix.  Every
            # rank sees the same modification (replicas), so every rank arrives here: gather the masters first (a collective).
            self._master_sync()
        if self._flat_lp is not self._flat and ver != self._flat_version:
            L.check(L.load().mtn_cast_f32_to_lp(L.dtype_code(self.compute_dtype), self._flat.numel(), self._flat.data_ptr(),
                                                self._flat_lp.data_ptr(), L.stream_ptr()))
            self._flat_version = ver
            self._ln_fold_stale = True
            self.refresh_transposed()
        elif self._flat_lp is self._flat and ver != self._flat_version:
            self._flat_version = ver
            self.refresh_transposed()
        return self

    def _tdesc_table(self, descs):
        """Device-side descriptor table of mtn_transpose_group for the (offset, rows, cols) weights in ``descs``."""
        if not descs:
            return None
        arr = (L.TransposeDesc * len(descs))()
        tiles = 0
        for i, (o, r, c) in enumerate(descs):
            arr[i].off, arr[i].dst_off, arr[i].rows, arr[i].cols, arr[i].tile_start = o, self._t_off[o], r, c, tiles
            tiles += ((r + 63) // 64) * ((c + 63) // 64)
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self._flat.device)
        return (raw, len(descs), tiles)

    def rest_tables(self, covered: frozenset):
        """What is left for the separate optimiser pass when the fusable weights at offsets ``covered`` were updated by the
        epilogue of their parameter-gradient GEMM: (chunks of <= 4096 elements of the flat buffers as device arrays
        (offsets, lengths, count), the transposed copies that still need the transpose pass).  Cached per set."""
        hit = self._rest_cache.get(covered)
        if hit is None:
            dev = self._flat.device
            total = self._flat.numel()
            done = [t for t in self._fusable if t[0] in covered]
            offs_c, lens_c, cur = [], [], 0
            for o, r, c in done + [(total, 0, 0)]:
                assert o % 4 == 0 and (r * c) % 4 == 0
                while cur < o:
                    n = min(4096, o - cur)
                    offs_c.append(cur); lens_c.append(n); cur += n
                cur = o + r * c
            chunks = (torch.tensor(offs_c, dtype=torch.int64).to(dev), torch.tensor(lens_c, dtype=torch.int32).to(dev), len(offs_c))
            dset = set(done)
            hit = (chunks, self._tdesc_table([t for t in self._tdescs_all if t not in dset]))
            self._rest_cache[covered] = hit
        return hit

    def refresh_transposed(self, table=None):
        """Rewrite the transposed compute-dtype weight copies from the current weights (one grouped kernel).
        ``table``: a subset (rest_tables()[1]) instead of all of them."""
        table = self._tdesc if table is None else table
        if table is None:
            return
        raw, n, tiles = table
        L.check(L.load().mtn_transpose_group(L.dtype_code(self.compute_dtype), self._flat_lp.data_ptr(), self._flat_lpT.data_ptr(),
                                             raw.data_ptr(), n, tiles, L.stream_ptr()))

    def flat_buffers(self):
        self.prepare()
        return self._flat, self._flat_lp, self._flat_grad

    def zero_glue_grads(self):
        """Gradients of path parameters are overwritten by the kernels every step; only the autograd-accumulated
        glue parameters (embeddings, feature Linear, generator) need zeroing."""
        self._flat_grad[: self._glue_numel].zero_()

    def advance_dropout_seed(self):
        if self._seed is not None:
            self._seed.add_(0x9E3779B97F4A7C15 & 0x7FFFFFFFFFFFFFFF)

    # ---- reference API ---------------------------------------------------------------------------
    def forward(self, b):                      # mtn.py:28-30
        q, v, cp, hs, ae = self.encode(b.query, b.query_mask, b.his, b.his_mask, b.cap, b.cap_mask, b.fts, b.fts_mask)
        return self.decode(v, hs, cp, q, b.fts_mask, b.his_mask, b.cap_mask, b.query_mask, b.trg, b.trg_mask, ae)

    def vid_encode(self, video_features, video_features_mask=None, encoded_query=None):   # mtn.py:32-36
        return [self.vid_encoder[i](ft) for i, ft in enumerate(video_features)]

    def _embed_fused(self, items):
        """items: [(tokens, Sequential(Embeddings, PositionalEncoding), LayerNorm | None)] -> list of fp32 tensors, each
        carrying its compute-dtype copy (._mtn_lp) when normalised.  One grouped HIP launch (ops.EmbedNormFn)."""
        luts, streams = [], []
        for k, (tok, emb, ln) in enumerate(items):
            lut = emb[0].lut.weight
            idx = next((j for j, t in enumerate(luts) if t is lut), None)
            if idx is None:
                luts.append(lut); idx = len(luts) - 1
            lninfo = None
            if ln is not None:
                ga, gb = ln._grads if ln._grads is not None else (None, None)
                lninfo = (ln.a_2, ln.b_2, ln.eps, ga, gb)
            streams.append(dict(tokens=tok, lut=idx, pe=emb[1].pe[0], scale=math.sqrt(emb[0].d_model),
                                p=emb[1].dropout.p if self.training else 0.0, salt=900000 + self._embed_calls * 16 + k, ln=lninfo))
        self._embed_calls += 1
        spec = dict(streams=streams, lp_dtype=self.compute_dtype, seed=self._seed if self.training else None, queue=self._queue)
        outs = ops.EmbedNormFn.apply(spec, *luts)
        for y, y_lp, (_, _, ln) in zip(outs, spec["_lp_out"], items):
            if ln is not None:
                y._mtn_lp = y_lp if y_lp is not None else y.detach()
        return list(outs)

    def _features_fused(self, vid):
        """vid_encoder (Linear -> ReLU -> PE -> dropout, mtn.py:378) + the Encoder LayerNorm of every feature stream on the
        HIP path (ops.FeatureEncodeFn): grouped cast, grouped GEMM, grouped LayerNorm launch."""
        streams, weights = [], []
        off = {id(p): o for p, o in zip(self._flat_params, self._flat_offsets)}
        for i, x in enumerate(vid):
            lin, pos = self.vid_encoder[i][0], self.vid_encoder[i][2]
            ln = self.query_encoder.norm[1 + i]
            o = off[id(lin.weight)]
            w_lp = self._flat_lp[o:o + lin.weight.numel()].view(lin.weight.shape)
            streams.append(dict(x=x, w_lp=w_lp, bias=lin.bias, grad_w=lin.weight.grad, grad_b=lin.bias.grad, pe=pos.pe[0],
                                p=pos.dropout.p if self.training else 0.0, salt=910000 + i,
                                ln=(ln.a_2, ln.b_2, ln.eps, ln._grads[0], ln._grads[1])))
            weights.append(lin.weight)
        spec = dict(streams=streams, lp_dtype=self.compute_dtype, seed=self._seed if self.training else None, queue=self._queue)
        outs = ops.FeatureEncodeFn.apply(spec, *weights)
        for y, y_lp in zip(outs, spec["_lp_out"]):
            y._mtn_lp = y_lp if y_lp is not None else y.detach()
        return list(outs)

    def _fused_embed_ok(self, tokens):
        return self.fused_embed and tokens.is_cuda and self._flat is not None and all(n._grads is not None for n in self.query_encoder.norm)

    def encode(self, query, query_mask, his=None, his_mask=None, cap=None, cap_mask=None, vid=None, vid_mask=None):
        """mtn.py:38-56 — every text stream goes through ``query_embed``; returns
        [q_mem, [vid_mem], cap_mem, his_mem, ae] with ae = list of auto-encoder seeds or None."""
        self.prepare()
        self.fold_layer_norms()                # (the forward kernels use the fold vectors too: LayerNorm as a rank-1 correction)
        if self.training:
            self.advance_dropout_seed()
        self._embed_calls = 0
        if self._fused_embed_ok(query):
            nF = len(vid)
            norm = self.query_encoder.norm
            items = [(query, self.query_embed, norm[0]), (cap, self.query_embed, norm[nF + 1]), (his, self.query_embed, norm[nF + 2])]
            if self.diff_encoder:
                ft = cap if self.auto_encoder_ft in ("caption", "summary") else query
                for i in range(nF):
                    emb = self.auto_encoder_embed[i] if self.auto_encoder_embed is not None else self.query_embed
                    items.append((ft, emb, norm[nF + 3 + i]))
            outs = self._embed_fused(items)
            if all(v.size(-1) % 8 == 0 for v in vid):
                vids = self._features_fused(vid)
            else:           # feature widths the GEMM's 16-byte row alignment cannot take: PyTorch Linear, HIP LayerNorm
                vids = [self.query_encoder._apply_norm(1 + i, v) for i, v in enumerate(self.vid_encode(vid, vid_mask))]
            return [outs[0], vids, outs[1], outs[2], outs[3:] if self.diff_encoder else None]
        streams = [self.query_embed(query), self.vid_encode(vid, vid_mask), self.query_embed(cap), self.query_embed(his)]
        if not self.diff_encoder:
            out = self.query_encoder(*streams)
            out.append(None)
            return out
        ft = cap if self.auto_encoder_ft in ("caption", "summary") else query
        if self.auto_encoder_embed is not None:
            ae = [self.auto_encoder_embed[i](ft) for i in range(len(vid))]
        else:
            ae = [self.query_embed(ft) for _ in range(len(vid))]
        return self.query_encoder(*streams, ae)

    def hoist_memory_kv(self, cap_memory, his_memory, q_memory, vid_fts, outs=None):
        """Project K|V of the encoder-side memories for every decoder layer that attends them (3 text cross-attentions + F
        video attentions per layer) in a few grouped GEMMs ahead of the layer loop (ops.project_memories).  The projections
        are left on the sublayer connections (``_kv_ready``) for the lockstep groups to pick up; clear_memory_kv() removes
        them.  Returns the list of K|V buffers (pass it back as ``outs`` to refresh them in place)."""
        if not (self.hoist_kv and self.lockstep and self._flat is not None and q_memory.is_cuda):
            return None
        nF = len(vid_fts)
        items, targets = [], []
        for layer in self.decoder.layers:
            if layer._forward_hooks_on_sublayers() or nF + 1 > 4:
                continue
            text, chains, _, _ = layer._plan(cap_memory, None, his_memory, None, q_memory, None, None, vid_fts, [None] * nF,
                                             self.auto_encoder_ft)
            for sc, mod, mem, _ in list(text[1:]) + [chains[i][1] for i in range(nF)]:
                mem_lp = getattr(mem, "_mtn_lp", None)
                f = mod.fused()
                if mem_lp is None or mem_lp.dtype != self.compute_dtype or f.get("w_qkv_lp") is None:
                    continue
                items.append((mem_lp, f["w_qkv_lp"], f["b_qkv"]))
                targets.append(sc)
        kvs = ops.project_memories(items, self.compute_dtype, outs)
        self._kv_targets = list(zip(targets, kvs))
        self.attach_memory_kv(self._kv_targets)
        return kvs

    @staticmethod
    def attach_memory_kv(pairs):
        for sc, kv in pairs:
            object.__setattr__(sc, "_kv_ready", kv)

    def clear_memory_kv(self):
        for layer in self.decoder.layers:
            for sc in layer.sublayer:
                if getattr(sc, "_kv_ready", None) is not None:
                    object.__setattr__(sc, "_kv_ready", None)

    def embed_target(self, tgt):
        """tgt_embed of mtn.py:59 (lookup * sqrt(d) + positional encoding + dropout)."""
        if self._fused_embed_ok(tgt):
            return self._embed_fused([(tgt, self.tgt_embed, None)])[0]
        return self.tgt_embed(tgt)

    def forward_segmented(self, b):
        """forward(b) with the autograd graph CUT at every decoder-layer boundary, for a backward that runs one layer at a
        time (train_step.TrainStep under data parallelism: layer k's gradient slice is all-reduced while layer k-1's
        backward runs).  Same arithmetic and launches as forward().  Returns a dict:
          enc_out / enc_leaf : encoder-side outputs (memories, auto-encoder seeds, embedded target) and their detached twins
          layers             : per layer (inputs [x, *ae] as leaves, outputs [x, *ae])
          top_in             : leaves feeding the final LayerNorms;  out, ae_out : as forward()."""
        q, v, cp, hs, ae = self.encode(b.query, b.query_mask, b.his, b.his_mask, b.cap, b.cap_mask, b.fts, b.fts_mask)
        x = self.embed_target(b.trg)
        enc_out = [q, cp, hs, x] + list(v) + (list(ae) if ae is not None else [])

        def cut(t):
            u = t.detach().requires_grad_()
            for attr in ("_mtn_lp", "_mtn_next"):      # compute-dtype copy; gradient hand-off to the producer across the cut
                if hasattr(t, attr):
                    setattr(u, attr, getattr(t, attr))
            return u

        enc_leaf = [cut(t) for t in enc_out]
        nF = len(v)
        q_l, cp_l, hs_l, x_l = enc_leaf[:4]
        v_l = enc_leaf[4:4 + nF]
        ae_l = enc_leaf[4 + nF:] if ae is not None else None
        ops.prepare_masks(b.trg_mask, b.his_mask, b.cap_mask, b.query_mask, b.fts_mask)
        self.hoist_memory_kv(cp_l, hs_l, q_l, v_l)
        layers = []
        x_in, ae_in = x_l, ae_l
        for layer in self.decoder.layers:
            x_out, ae_out = layer(x_in, cp_l, b.cap_mask, hs_l, b.his_mask, q_l, b.query_mask, b.trg_mask, v_l, b.fts_mask,
                                  ae_in, self.auto_encoder_ft)
            ins = [x_in] + (list(ae_in) if isinstance(ae_in, (list, tuple)) else [])
            layers.append((ins, [x_out] + list(ae_out)))
            x_in, ae_in = cut(x_out), [cut(a) for a in ae_out]
        self.clear_memory_kv()
        top_in = [x_in] + ae_in
        if x_in.is_cuda and len(ae_in) == len(self.decoder.ae_norm):
            ys = ops.layer_norm_group([x_in] + list(ae_in), [self.decoder.norm] + list(self.decoder.ae_norm))    # as Decoder.forward
            out, ae_fin = ys[0], ys[1:]
        else:
            out = self.decoder.norm(x_in)
            ae_fin = [self.decoder.ae_norm[i](a) for i, a in enumerate(ae_in)]
        return dict(enc_out=enc_out, enc_leaf=enc_leaf, layers=layers, top_in=top_in, out=out, ae_out=ae_fin)

    def decode(self, encoded_vid_features, his_memory, cap_memory, query_memory, vid_features_mask, his_mask, cap_mask,
               query_mask, tgt, tgt_mask, auto_encoded_ft):          # mtn.py:58-60
        self.prepare()
        x0 = self.embed_target(tgt)
        self.hoist_memory_kv(cap_memory, his_memory, query_memory, encoded_vid_features)
        try:
            return self._decode_layers(encoded_vid_features, vid_features_mask, x0, his_memory, his_mask, cap_memory, cap_mask,
                                       query_memory, query_mask, tgt_mask, auto_encoded_ft)
        finally:
            self.clear_memory_kv()

    def _decode_layers(self, encoded_vid_features, vid_features_mask, x0, his_memory, his_mask, cap_memory, cap_mask, query_memory,
                       query_mask, tgt_mask, auto_encoded_ft):
        return self.decoder(encoded_vid_features, vid_features_mask, x0, his_memory, his_mask, cap_memory,
                            cap_mask, query_memory, query_mask, tgt_mask, auto_encoded_ft, self.auto_encoder_ft)


def make_model(src_vocab, tgt_vocab, N=6, d_model=512, d_ff=2048, h=8, dropout=0.1, separate_his_embed=False,
               separate_cap_embed=False, ft_sizes=None, diff_encoder=False, diff_embed=False, diff_gen=False,
               auto_encoder_ft=None, auto_encoder_attn=False, compute_dtype="bf16", attn_dropout=ATTN_DROPOUT_DEFAULT):
    """Same signature and defaults as mtn.py:332-337 (+ ``compute_dtype``: 'bf16' | 'fp32', and
    ``attn_dropout`` exposing the reference's hard-wired attention-probability dropout of 0.1).
    Xavier-uniform on every parameter with dim > 1, nn.Linear defaults for biases (mtn.py:410-412)."""
    ft_sizes = list(ft_sizes or [])
    F_ = len(ft_sizes)
    attn = lambda: MultiHeadedAttention(h, d_model, attn_dropout)
    ff = lambda: PositionwiseFeedForward(d_model, d_ff, dropout)
    pos = lambda: PositionalEncoding(d_model, dropout)
    embed = lambda vocab: nn.Sequential(Embeddings(d_model, vocab), pos())
    query_embed, tgt_embed = embed(src_vocab), embed(tgt_vocab)
    his_embed = embed(src_vocab) if separate_his_embed else None
    cap_embed = embed(src_vocab) if separate_cap_embed else None
    auto_encoder_embed = nn.ModuleList([embed(src_vocab) for _ in ft_sizes]) if diff_embed else None
    query_encoder = Encoder(d_model, nb_layers=3 + (2 * F_ if diff_encoder else F_))
    vid_encoder = nn.ModuleList([nn.Sequential(nn.Linear(fs, d_model), nn.ReLU(), pos()) for fs in ft_sizes])
    generator = Generator(d_model, tgt_vocab)
    auto_encoder_generator = nn.ModuleList([Generator(d_model, tgt_vocab) for _ in ft_sizes]) if diff_gen else None

    def make_layer():
        return DecoderLayer(d_model, attn(), attn(), attn(), attn(),
                            nn.ModuleList([attn() for _ in ft_sizes]), nn.ModuleList([attn() for _ in ft_sizes]),
                            nn.ModuleList([attn() for _ in ft_sizes]), ff(), nn.ModuleList([ff() for _ in ft_sizes]), dropout)

    decoder = Decoder(make_layer, N, ft_sizes)
    cd = {"bf16": torch.bfloat16, "fp32": torch.float32}.get(compute_dtype, compute_dtype)
    model = EncoderDecoder(query_encoder=query_encoder, his_encoder=None, cap_encoder=None, vid_encoder=vid_encoder,
                           decoder=decoder, query_embed=query_embed, his_embed=his_embed, cap_embed=cap_embed,
                           tgt_embed=tgt_embed, generator=generator, auto_encoder_generator=auto_encoder_generator,
                           auto_encoder_embed=auto_encoder_embed, diff_encoder=diff_encoder, auto_encoder_ft=auto_encoder_ft,
                           compute_dtype=cd)
    for p in model.parameters():
        if p.dim() > 1:
            nn.init.xavier_uniform_(p)
    return model
