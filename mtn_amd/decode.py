"""Decode path (BASELINE configs[4]): greedy / beam search over the MTN decoder for ONE dialogue, as generate.py drives it
through data_utils.py:159-242, on the HIP kernels.

What the reference does per generated token (data_utils.py:200-205): for EVERY live hypothesis, one full ``model.decode`` of
its prefix — which re-runs all N decoder layers *including the auto-encoder chains*, although those chains depend only on
the dialogue (query/caption, video), never on the target prefix.  Here:

* the encoder side and the N x F auto-encoder chains run ONCE per dialogue (``DecoderLayer.forward_ae_chains``);
* all live hypotheses are decoded together, as the batch dimension of one target-stream pass
  (``DecoderLayer.forward_target``: 4 text attentions + F attend-to-auto-encoder + FFN per layer);
* the pass has a FIXED shape — (width, max_len) tokens under the causal mask, log-probabilities read at position l — so the
  whole pass is ONE hipGraph replayed per token (launch-latency-bound at the reference's max_len = 20);
* for longer searches (``kv_cache``, default above KV_CACHE_FROM tokens) the pass covers only the NEWEST position: every layer
  keeps K|V of the target prefix per hypothesis, the new row's K|V are projected into it, and the self-attention of that row is
  a cross-attention over the cache (keys 0..l enabled); a beam step re-orders hypotheses, so the cache rows follow their
  parents by one gather per token.  Same n-best lists as the full-prefix pass (tested), O(l) instead of O(l^2) work per token;
* hypothesis bookkeeping stays on the host exactly as in the reference (same candidate order, same tie behaviour).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import os

import torch

from . import ops
from .data_utils import subsequent_mask


class DecodeSession:
    """Everything about one dialogue that does not depend on the target prefix, plus the replayable target-stream pass.
    A session is reusable for further dialogues of the same shapes (``load``): buffers and the captured graph persist."""

    def __init__(self, model, batch, max_len: int, width: int, pad: int = 1, use_graph: bool = True, kv_cache: bool = False,
                 select=None):
        self.model, self.width, self.max_len, self.pad = model, width, max_len, pad
        self.kv_cache = bool(kv_cache)
        # select = (k, column): the pass also leaves, per row, its k largest log-probabilities, their columns and that column's
        # value in self.top (ops.topk_rows, inside the captured graph): the beam search reads only that
        self.select = select if (select is not None and batch.query.is_cuda) else None
        self.top = None
        self.use_graph = use_graph and batch.query.is_cuda
        dev = batch.query.device
        self.q = self.cp = self.hs = self.aes = self.masks = None
        self._kvs, self._kv_pairs = None, []
        self.D = batch.query.size(0)
        self.tokens = torch.full((self.D * width, max_len), pad, dtype=torch.long, device=dev)
        self.pos = torch.zeros(1, dtype=torch.long, device=dev)
        self.trg_mask = subsequent_mask(max_len, device=dev)       # (1, L, L): data_utils.py:204 uses the causal mask only
        ops.prepare_masks(self.trg_mask)
        self.logp = None
        self._graph = None
        self._static = None                   # load(): a private copy of the dialogue's inputs that a captured encoder-side pass reads
        self._load_graph = None
        self._loads = 0
        self.load(batch)
        if self.kv_cache:
            # Prefix K/V cache of the target self-attention (the reference recomputes the whole prefix for every hypothesis and
            # token, data_utils.py:197-205): per layer (W, L, 2d) in the compute dtype, TWO copies — a beam step re-orders the
            # hypotheses, so the rows of the cache follow their parents by one gather from the other copy per token.
            W, L, d = self.D * width, max_len, model.decoder.layers[0].size
            lp = model.compute_dtype
            nl = len(model.decoder.layers)
            self._cache = [torch.zeros(nl, W, L, 2 * d, device=dev, dtype=lp) for _ in range(2)]
            self._cur = 0
            self._parent = torch.arange(W, device=dev)
            self._self_mem = torch.zeros(W, L, d, device=dev, dtype=torch.float32)      # shape carrier (never read: K|V are ready)
            self._self_mem._mtn_lp = torch.zeros(W, L, d, device=dev, dtype=lp) if lp != torch.float32 else self._self_mem
            self._self_mask = torch.zeros(W, 1, L, dtype=torch.bool, device=dev)
            self._self_mask._mtn_u8 = torch.zeros(W, 1, L, dtype=torch.uint8, device=dev)
            self._prev = None                                                            # prefixes of the previous step, per dialogue
            self._graphs = {}

    @staticmethod
    def signature(model, batch, max_len, width):
        return (id(model), model._flat.data_ptr() if model._flat is not None else 0, model.compute_dtype, width, max_len, tuple(batch.query.shape), tuple(batch.his.shape),
                tuple(batch.cap.shape), tuple(tuple(f.shape) for f in (batch.fts or [])), str(batch.query.device))

    def load(self, batch):
        """Encoder side + the N x F auto-encoder chains of a new dialogue (target-independent: once per dialogue).  A batch of
        D dialogues is decoded side by side: every per-dialogue tensor is repeated `width` times along the batch dimension
        (rows d*width .. d*width+width-1 belong to dialogue d).
        The pass is ~100 small launches, host-bound when issued one by one (2-3 ms per dialogue — a fifth of a 20-token beam search):
        from the third dialogue of a shape on, the inputs are copied into a private static Batch and the whole pass is ONE graph replay."""
        model = self.model
        if model.training:                       # (nn.Module.eval() walks every submodule: ~1 ms of Python per dialogue)
            model.eval()
        model.prepare()
        self._loads += 1
        # what a captured pass froze: the weights' flat buffers (prepare() re-creates them when parameters were replaced) and the set of
        # hoisted projections — if either changed since the capture, the graph is stale: capture again
        ver = (getattr(model, "_flat_version", None), len(getattr(model, "_kv_targets", None) or ()))
        if getattr(self, "_load_ver", ver) != ver:
            self._load_graph, self._loads = None, 2 if self._static is not None else 1
        self._load_ver = ver
        if not (self.use_graph and batch.query.is_cuda) or self._loads == 1:
            return self._load_body(batch)
        self._stage(batch)
        if self._load_graph is None:
            if self._loads == 2:
                return self._load_body(self._static)       # warm-up on the static inputs (and this dialogue's results)
            try:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._load_body(self._static)
                self._load_graph = g
            except Exception as e:                          # never lose a decode to the faster path
                import logging
                logging.getLogger("mtn_amd").warning("decode: capturing the encoder-side pass failed (%s); it stays eager", e)
                self._load_graph = False
                torch.cuda.synchronize()
                return self._load_body(self._static)
        if self._load_graph is False:
            return self._load_body(self._static)
        self._load_graph.replay()

    def _stage(self, b):
        """Copy a dialogue's inputs into the session's static Batch (created from the first one staged)."""
        import copy
        tensors = lambda x: [x.query, x.query_mask, x.his, x.his_mask, x.cap, x.cap_mask] + list(x.fts or []) + list(x.fts_mask or [])
        if self._static is None:
            st = copy.copy(b)
            st.query, st.query_mask, st.his, st.his_mask, st.cap, st.cap_mask = (t.clone() for t in (b.query, b.query_mask, b.his, b.his_mask, b.cap, b.cap_mask))
            st.fts, st.fts_mask = [t.clone() for t in (b.fts or [])], [t.clone() for t in (b.fts_mask or [])]
            self._static = st
            return
        mine, theirs = tensors(self._static), tensors(b)
        # the captured pass is frozen to the first staged dialogue's shapes and dtypes: anything else must not be replayed into it
        if len(mine) != len(theirs) or any(d_.shape != s_.shape or d_.dtype != s_.dtype for d_, s_ in zip(mine, theirs)):
            raise ValueError("DecodeSession.load: this dialogue's tensors differ in shape / dtype from the session's (sessions are per shape: decode._session keys them)")
        for dst, src in zip(mine, theirs):
            dst.copy_(src)

    def _load_body(self, batch):
        model, width, b = self.model, self.width, batch
        self.D = D = b.query.size(0)
        lp = model.compute_dtype
        if b is self._static:                    # its masks' kernel images exist from the warm-up: refresh them (part of the captured pass)
            for mk in [b.query_mask, b.his_mask, b.cap_mask] + list(b.fts_mask):
                if getattr(mk, "_mtn_u8", None) is not None:
                    mk._mtn_u8.copy_(mk)
        with torch.no_grad():
            q, v, cp, hs, ae = model.encode(b.query, b.query_mask, b.his, b.his_mask, b.cap, b.cap_mask, b.fts, b.fts_mask)
            ops.prepare_masks(b.his_mask, b.cap_mask, b.query_mask, b.fts_mask)
            aes_per_layer = []
            for layer in model.decoder.layers:
                ae = layer.forward_ae_chains(cp, b.cap_mask, q, b.query_mask, v, b.fts_mask, ae, model.auto_encoder_ft)
                aes_per_layer.append(ae)

            def widen(t, into=None):                           # (D, m, d) -> (D*width, m, d), with its compute-dtype copy
                if into is None:
                    into = torch.empty(D * width, t.size(1), t.size(2), device=t.device, dtype=t.dtype)
                    into._mtn_lp = torch.empty_like(into, dtype=lp) if lp != torch.float32 else into
                into.view(D, width, t.size(1), t.size(2)).copy_(t.unsqueeze(1).expand(-1, width, -1, -1))
                if into._mtn_lp is not into:
                    into._mtn_lp.copy_(into)
                return into

            first = self.q is None
            self.q, self.cp, self.hs = widen(q, self.q), widen(cp, self.cp), widen(hs, self.hs)
            self.aes = [[widen(a, None if first else self.aes[k][i]) for i, a in enumerate(aes)] for k, aes in enumerate(aes_per_layer)]
            # K|V of the (widened) text memories for all layers: once per dialogue, not once per token; refreshed in place so
            # that a captured pass keeps reading the same buffers
            kvs = model.hoist_memory_kv(self.cp, self.hs, self.q, [], outs=None if first else self._kvs)
            if kvs is not None:
                self._kvs, self._kv_pairs = kvs, list(model._kv_targets)
                # ... and of the auto-encoder outputs the target stream attends un-projected (mtn.py:215): they do not change
                # with the prefix either, so their K|V leave the per-token pass as well (2 x N projections of Q rows per token)
                items, scs = [], []
                for k, layer in enumerate(model.decoder.layers):
                    for i, mem in enumerate(self.aes[k]):
                        f = layer.auto_encoder_attn[i].fused()
                        if f.get("w_qkv_lp") is None or mem._mtn_lp.dtype != lp:
                            continue
                        items.append((mem._mtn_lp, f["w_qkv_lp"], f["b_qkv"]))
                        scs.append(layer.sublayer[7 + 4 * i])
                if items:
                    self._ae_kvs = ops.project_memories(items, lp, None if first else getattr(self, "_ae_kvs", None))
                    self._kv_pairs += list(zip(scs, self._ae_kvs))
            model.clear_memory_kv()
            new_masks = (b.cap_mask, b.his_mask, b.query_mask)
            rep = lambda mk: mk.repeat_interleave(width, dim=0)
            if first:
                self.masks = tuple(rep(mk) for mk in new_masks)
                ops.prepare_masks(*self.masks)
            else:
                for mine, mk in zip(self.masks, new_masks):    # the captured pass reads the uint8 images
                    mine.copy_(rep(mk))
                    mine._mtn_u8.copy_(mine)

    def _pass(self):
        m = self.model
        cap_mask, his_mask, q_mask = self.masks
        x = m.embed_target(self.tokens)
        m.attach_memory_kv(self._kv_pairs)
        try:
            for k, layer in enumerate(m.decoder.layers):
                x = layer.forward_target(x, self.cp, cap_mask, self.hs, his_mask, self.q, q_mask, self.trg_mask, self.aes[k],
                                         m.auto_encoder_ft)
        finally:
            m.clear_memory_kv()
        x = m.decoder.norm(x)
        last = x.index_select(1, self.pos).squeeze(1)               # (width, d): the position being extended
        self.logp = m.generator(last).float()                       # (width, V) log-probabilities (mtn.py:68-69)
        if self.select is not None:
            self.top = ops.topk_rows(self.logp, min(self.select[0], self.logp.size(1)), self.select[1])

    def _pass_cached(self, cur: int):
        """One target position (self.pos) for every hypothesis, against the prefix cache: gather the cache rows of the parents
        (copy 1-cur -> cur), embed the newest token, per layer project its K|V into the cache and run the layer on that one row."""
        m = self.model
        cap_mask, his_mask, q_mask = self.masks
        src, dst = self._cache[1 - cur], self._cache[cur]
        torch.index_select(src, 1, self._parent, out=dst)
        tok = self.tokens.index_select(1, self.pos)                                  # (W, 1)
        emb, pos_enc = m.tgt_embed[0], m.tgt_embed[1]
        x = emb(tok) + pos_enc.pe[0].index_select(0, self.pos).unsqueeze(0)           # lut * sqrt(d) + PE[l]   (mtn.py:289, 308; eval: no dropout)
        ar = torch.arange(self.max_len, device=x.device)
        self._self_mask._mtn_u8.copy_((ar <= self.pos).view(1, 1, -1).expand_as(self._self_mask))
        m.attach_memory_kv(self._kv_pairs)
        try:
            for k, layer in enumerate(m.decoder.layers):
                kv_new = layer.self_kv_of_new_rows(x)                                # (W, 1, 2d)
                dst[k].index_copy_(1, self.pos, kv_new)
                x = layer.forward_target_cached(x, self.cp, cap_mask, self.hs, his_mask, self.q, q_mask, self.aes[k], m.auto_encoder_ft,
                                                dst[k].view(-1, dst.size(-1)), self._self_mem, self._self_mask)
        finally:
            m.clear_memory_kv()
        x = m.decoder.norm(x)
        self.logp = m.generator(x.squeeze(1)).float()
        if self.select is not None:
            self.top = ops.topk_rows(self.logp, min(self.select[0], self.logp.size(1)), self.select[1])

    def _step_cached(self, prefix_lists):
        l = len(prefix_lists[0][0])
        W = self.width
        host_tok = torch.full((self.D * W, self.max_len), self.pad, dtype=torch.long)
        parent = torch.arange(self.D * W)
        for d, prefixes in enumerate(prefix_lists):
            host_tok[d * W:d * W + len(prefixes), :l] = torch.tensor(prefixes, dtype=torch.long)
            if l > 1:
                prev = self._prev[d]
                for i, p in enumerate(prefixes):                    # the hypothesis this one extends (same tokens but the last)
                    parent[d * W + i] = d * W + prev.index(list(p[:-1]))
        self._prev = [[list(p) for p in prefixes] for prefixes in prefix_lists]
        self.tokens.copy_(host_tok)
        self._parent.copy_(parent)
        self.pos.fill_(l - 1)
        cur = self._cur
        with torch.no_grad():
            if not self.use_graph:
                self._pass_cached(cur)
            else:
                g = self._graphs.get(cur)
                if g is None:
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        self._pass_cached(cur)
                    torch.cuda.current_stream().wait_stream(side)
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        self._pass_cached(cur)
                    g = self._graphs[cur] = (g, self.logp, self.top)          # each graph writes its own output buffers
                g[0].replay()
                self.logp, self.top = g[1], g[2]
        self._cur = 1 - cur
        return [self.logp[d * W:d * W + len(p)] for d, p in enumerate(prefix_lists)]

    def step(self, prefixes: Sequence[Sequence[int]]) -> torch.Tensor:
        """Log-probabilities (n, V) of the next token after each prefix of ONE dialogue (all prefixes have the same length)."""
        return self.step_many([prefixes])[0]

    def step_many(self, prefix_lists: Sequence[Sequence[Sequence[int]]]):
        """One decode step for all dialogues of the session: prefix_lists[d] = live prefixes of dialogue d (same length
        everywhere, at most `width` per dialogue) -> list of (n_d, V) log-probability tensors."""
        if len(prefix_lists) != self.D:
            raise ValueError("one prefix list per dialogue of the session")
        l = len(prefix_lists[0][0])
        if max(len(p) for p in prefix_lists) > self.width or l > self.max_len:
            raise ValueError("more hypotheses / longer prefix than the session was built for")
        if self.kv_cache:
            return self._step_cached(prefix_lists)
        host = torch.full((self.D * self.width, self.max_len), self.pad, dtype=torch.long)
        for d, prefixes in enumerate(prefix_lists):
            host[d * self.width:d * self.width + len(prefixes), :l] = torch.tensor(prefixes, dtype=torch.long)
        self.tokens.copy_(host, non_blocking=False)
        self.pos.fill_(l - 1)
        with torch.no_grad():
            if not self.use_graph:
                self._pass()
            else:
                if self._graph is None:
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        self._pass()
                    torch.cuda.current_stream().wait_stream(side)
                    torch.cuda.synchronize()
                    self._graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self._graph):
                        self._pass()
                self._graph.replay()
        return [self.logp[d * self.width:d * self.width + len(p)] for d, p in enumerate(prefix_lists)]


class MegaDecodeSession(DecodeSession):
    """The per-token pass as ONE persistent launch (csrc/decode.hip, include/mtn_hip.h mtn_decode_step) + the generator's three small
    launches, for sessions of at most 16 live hypotheses on a bf16 model (8 at d_ff = 4096): the newest position of every hypothesis walks a
    device-resident stage list (per layer: self-attention over its prefix cache, the cross-attentions over the K|V hoisted by load(),
    the feed-forward) with grid barriers instead of ~90 dependent launches.  The prefix cache is never copied when a beam step
    re-orders hypotheses: position t of hypothesis j's prefix is read from cache slot anc[j][t], a (W, L) int table the host updates
    from the parents (the slot of row j at position l-1 is j itself).  Same search results as the launch-per-sublayer pass (tested).

    The launch needs ALL its workgroups resident at once (they poll each other).  `supported()` refuses devices with fewer compute units than
    the launch has workgroups; if a poll still times out (another kernel held compute units), `timed_out()` reports it, `recover()` makes the
    session's device state consistent again, and the callers below re-run the search on the launch-per-sublayer pass (FALLBACKS counts them)."""

    MAX_W = 16
    FALLBACKS = 0          # searches re-run on the launch-per-sublayer pass after a poll timeout of the persistent step

    @staticmethod
    def supported(model, batch, max_len, width) -> bool:
        if os.environ.get("MTN_DECODE_MEGA", "1") == "0" or not batch.query.is_cuda:
            return False
        try:
            layer = model.decoder.layers[0]
            d, h = layer.size, layer.self_attn.h
            dff = layer.feed_forward.w_1.weight.size(0)
        except AttributeError:
            return False
        W = batch.query.size(0) * width
        if model.compute_dtype != torch.bfloat16 or W > MegaDecodeSession.MAX_W or W * d > 8192 or d not in (128, 256, 512, 1024) or d % h or d // h not in (32, 64):
            return False
        if dff > 4096 or dff < d or dff % 32 or max_len > 1024 or model.auto_encoder_ft not in ("query", "caption", "summary"):
            return False
        if W * (2 * max(d, dff) + 16) > 66048:                      # csrc/decode.hip DEC_ACT_BYTES: the activation image of the widest Linear
            return False
        if max([batch.his.size(1), batch.cap.size(1), batch.query.size(1)] + [f.size(1) for f in (batch.fts or [])]) > 1024:
            return False
        # every workgroup of the launch must be resident at once, one per compute unit: W x h attention units + d / 16 writers of the
        # residual stream + enough "wide" workgroups that a slice of the q|k|v / first FFN projection is at most 64 features
        grid = MegaDecodeSession.grid_for(batch.query.device)
        n_mid = min(128, grid - W * h - d // 16)
        if W * h > 128 or n_mid < 16 or (-(-max(3 * d, dff) // n_mid) + 3) // 4 * 4 > 64:
            return False
        n_stages = 2 + sum(3 + 2 * (3 + len(l.auto_encoder_attn)) + 2 for l in model.decoder.layers)       # csrc/decode.hip DEC_MAX_STAGES
        return n_stages <= 160 and all(len(l.sublayer) == 5 + 4 * len(l.auto_encoder_attn) for l in model.decoder.layers)

    @staticmethod
    def grid_for(device) -> int:
        """The most workgroups the persistent launch may use on this device: one per compute unit, at most 256."""
        return min(256, int(torch.cuda.get_device_properties(device).multi_processor_count))

    def __init__(self, model, batch, max_len, width, pad=1, use_graph=True, select=None):
        super().__init__(model, batch, max_len, width, pad=pad, use_graph=use_graph, kv_cache=False, select=select)
        from . import lib as L
        dev = batch.query.device
        layers = model.decoder.layers
        d, h = layers[0].size, layers[0].self_attn.h
        dff = layers[0].feed_forward.w_1.weight.size(0)
        W, Lm, nl = self.D * width, max_len, len(layers)
        lp = torch.bfloat16
        self._W = W
        self._mcache = torch.zeros(nl, W, Lm, 2 * d, device=dev, dtype=lp)
        # granule buffers (8 bytes {data, tag}): residual stream, q|k|v of the newest row, attention output, FFN hidden — zeroed once
        self._x = torch.zeros(W, d, device=dev, dtype=torch.int64)
        self._q = torch.zeros(W, 3 * d // 2, device=dev, dtype=torch.int64)
        self._o = torch.zeros(W, d // 2, device=dev, dtype=torch.int64)
        self._hid = torch.zeros(W, dff // 2, device=dev, dtype=torch.int64)
        self._out_lp = torch.zeros(W, d, device=dev, dtype=lp)
        self._sync = torch.zeros(4, device=dev, dtype=torch.int32)
        # what the host changes every step, in ONE pinned block -> ONE copy: [W newest tokens (int64) | position (int32, padded) | anc (W x L int32)]
        self._off_pos, self._off_anc = 8 * W, 8 * W + 8
        nbytes = self._off_anc + 4 * W * Lm
        self._host = torch.zeros(nbytes, dtype=torch.uint8).pin_memory()
        self._devblk = torch.zeros(nbytes, device=dev, dtype=torch.uint8)
        import numpy as np
        hb = self._host.numpy()                       # numpy views of the pinned block: the per-step bookkeeping below is plain numpy slicing
        self._h_tok = hb[:8 * W].view(np.int64)
        self._h_pos = hb[self._off_pos:self._off_pos + 8].view(np.int32)
        self._h_anc = hb[self._off_anc:].view(np.int32).reshape(W, Lm)
        self._h_anc[:] = np.arange(W, dtype=np.int32)[:, None]
        self._prev = None
        self._top_host = None
        self._grid = self.grid_for(dev)     # the most the launch may use (one workgroup per CU); the library deals them to its three classes
        self._build_stages(L, d, h, dff)
        emb, pe = model.tgt_embed[0], model.tgt_embed[1]
        a = L.DecodeArgs()
        a.W, a.d, a.h, a.L, a.n_stages, a.d_ff, a.max_m = W, d, h, Lm, self._n_stages, dff, self._max_m
        self._dbg = torch.zeros(4 * self._n_stages, device=dev, dtype=torch.int64) if os.environ.get("MTN_DECODE_TIMELINE") == "1" else None
        a.dbg = self._dbg.data_ptr() if self._dbg is not None else None
        a.xg, a.qg, a.og, a.hg, a.out_lp = self._x.data_ptr(), self._q.data_ptr(), self._o.data_ptr(), self._hid.data_ptr(), self._out_lp.data_ptr()
        a.tokens = self._devblk.data_ptr()
        a.lut, a.emb_scale, a.pe = emb.lut.weight.data_ptr(), float(d) ** 0.5, pe.pe.data_ptr()
        a.pos, a.anc, a.sync = self._devblk.data_ptr() + self._off_pos, self._devblk.data_ptr() + self._off_anc, self._sync.data_ptr()
        self._args = a
        if pe.pe.size(1) < Lm or not emb.lut.weight.is_contiguous():
            raise ValueError("positional-encoding table shorter than max_len")

    def _build_stages(self, L, d, h, dff):
        """The stage list of one decode step (include/mtn_hip.h MTN_DEC_*): per decoder layer the schedule of mtn.py:183-218 for the
        target stream — self-attention, history, caption | query (order by auto_encoder_ft), one attention per auto-encoder stream,
        feed-forward — every attention followed by its output projection; pointers into the model's flat buffers and this session's
        hoisted K|V / masks (all refreshed IN PLACE by load(), so the table is built once)."""
        import ctypes as C
        m = self.model
        cap_mask, his_mask, q_mask = self.masks
        kvmap = {id(sc): kv for sc, kv in self._kv_pairs}
        st = []
        self._max_m = 1

        def stage(kind, **kw):
            s_ = L.DecodeStage()
            s_.kind = kind
            for k, v in kw.items():
                setattr(s_, k, v)
            st.append(s_)

        def ln(sc):
            return dict(ln_a=sc.norm.a_2.data_ptr(), ln_b=sc.norm.b_2.data_ptr(), ln_eps=float(sc.norm.eps))

        stage(L.DEC_EMBED)
        for k, layer in enumerate(m.decoder.layers):
            sl, nF = layer.sublayer, len(layer.auto_encoder_attn)
            f = layer.self_attn.fused()
            cache = self._mcache[k].data_ptr()
            stage(L.DEC_SELF_QKV, N=3 * d, K=d, w=f["w_qkv_lp"].data_ptr(), bias=f["b_qkv"].data_ptr(), cache=cache, **ln(sl[0]))
            stage(L.DEC_SELF_ATT, cache=cache)
            stage(L.DEC_OUT, N=d, K=d, w=f["w_o_lp"].data_ptr(), bias=f["b_o"].data_ptr())
            text, _, _, ae_mask = layer._plan(self.cp, cap_mask, self.hs, his_mask, self.q, q_mask, None, [None] * nF, [None] * nF, m.auto_encoder_ft)
            cross = list(text[1:]) + [(sl[7 + 4 * i], layer.auto_encoder_attn[i], self.aes[k][i], ae_mask) for i in range(nF)]
            for sc, mod, mem, mask in cross:
                f = mod.fused()
                kv = kvmap.get(id(sc))
                if kv is None or kv.dtype != torch.bfloat16:
                    raise ValueError("a memory's K|V were not hoisted")
                mk = getattr(mask, "_mtn_u8", None)
                self._max_m = max(self._max_m, int(mem.size(1)))
                stage(L.DEC_CROSS, N=d, K=d, w=f["w_qkv_lp"].data_ptr(), bias=f["b_qkv"].data_ptr(), kv=kv.data_ptr(), m=mem.size(1),
                      mask=mk.data_ptr() if mk is not None else None, mask_stride=mem.size(1), **ln(sc))
                stage(L.DEC_OUT, N=d, K=d, w=f["w_o_lp"].data_ptr(), bias=f["b_o"].data_ptr())
            ff = layer.feed_forward.fused()
            stage(L.DEC_FFN1, N=dff, K=d, w=ff["w1_lp"].data_ptr(), bias=ff["b1"].data_ptr(), **ln(sl[4 + 4 * nF]))
            stage(L.DEC_FFN2, N=d, K=dff, w=ff["w2_lp"].data_ptr(), bias=ff["b2"].data_ptr())
        nrm = m.decoder.norm
        stage(L.DEC_FINAL, ln_a=nrm.a_2.data_ptr(), ln_b=nrm.b_2.data_ptr(), ln_eps=float(nrm.eps))
        arr = (L.DecodeStage * len(st))(*st)
        self._n_stages = len(st)
        self._stages_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self._x.device)

    def _pass_mega(self):
        import ctypes as C
        from . import lib as L
        m = self.model
        # the step's host inputs and (with device-side selection) its host outputs travel INSIDE the pass — copy nodes of the captured graph —
        # so a generated token costs the host one replay and one stream synchronisation
        self._devblk.copy_(self._host, non_blocking=True)
        L.check(L.load().mtn_decode_step(C.byref(self._args), self._stages_dev.data_ptr(), self._grid, L.stream_ptr()))
        g = m.generator._fused
        self.logp = ops.generator_log_probs(self._out_lp, g["w_lp"], g["bias"])          # (W, V) fp32 log-probabilities (mtn.py:68-69)
        if self.select is not None:
            self.top = ops.topk_rows(self.logp, min(self.select[0], self.logp.size(1)), self.select[1])
            if self._top_host is None:
                self._top_host = torch.empty(self.top.shape, dtype=self.top.dtype).pin_memory()
            self._top_host.copy_(self.top, non_blocking=True)

    def top_host(self):
        """The rows' heads of the last step on the host (a pinned block the pass itself filled)."""
        torch.cuda.current_stream().synchronize()
        return self._top_host.numpy()

    def search(self, beam, k, start, unk, eos, penalty, min_len, nbest):
        log = self._search_log(beam, k, start, unk, eos, penalty, min_len, eos)
        if log is None:
            return None
        par, tok, n_old, n_new, score, done_sc, flags = log
        results = []
        for d_ in range(self.D):
            base = d_ * self.width
            outs, done = [[]], []
            for l in range(self.max_len):
                if l >= min_len:
                    done += [(outs[h], float(done_sc[l, base + h])) for h in range(int(n_old[l, d_]))]
                outs = [outs[int(par[l, base + i])] + [int(tok[l, base + i])] for i in range(int(n_new[l, d_]))]
            if done:
                results.append((sorted(done, key=lambda h: -h[1])[:nbest], max(h[1] for h in done)))
            else:
                results.append(([([], 0)], None))
        return results

    def greedy(self, start):
        """argmax decoding of the session's (one) dialogue as one graph replay: a beam of one that skips nothing and finishes nothing
        (mtn_beam_advance with beam = k = 1, no <unk> / <eos>).  Returns the max_len - 1 generated tokens, or None (tie / not applicable)."""
        log = self._search_log(1, 1, start, -1, -1, 0.0, self.max_len + 1, self.select[1] if self.select is not None else 0)
        return None if log is None else [int(t) for t in log[1][:self.max_len - 1, 0]]

    def _search_log(self, beam, k, start, unk, eos, penalty, min_len, extra_col):
        """A whole beam search (data_utils.py:188-242) for every dialogue of the session as ONE graph replay: max_len x [persistent decode
        step, generator, row heads (csrc/select.hip topk_rows), hypothesis bookkeeping on the device (mtn_beam_advance)], the step log
        copied to a pinned block at the end.  The host synchronises once per search and rebuilds the n-best lists from the log.
        Returns None when a row's head held an exact tie (the reference's visiting order then comes from the full row: the caller runs the
        search step by step) or when the device-side selection does not apply."""
        import ctypes as C
        import numpy as np
        from . import lib as L
        k_top = k + 1
        if self.select is None or self.select[0] != k_top or self.select[1] != extra_col or beam > self.width or k_top > SELECT_MAX_K or not self.use_graph:
            return None
        W, D, Lm = self._W, self.D, self.max_len
        key = (beam, k, start, unk, eos, float(penalty), min_len)
        if getattr(self, "_search_key", None) != key:
            dev = self._x.device
            # device state of the search [lp (W doubles) | n_live (D) | step (D) | flags (2)] and its initial image
            nst = 8 * W + 4 * (2 * D + 2)
            self._bstate = torch.zeros(nst, device=dev, dtype=torch.uint8)
            init = np.zeros(nst, dtype=np.uint8)
            init[8 * W:8 * W + 4 * D].view(np.int32)[:] = 1
            self._bstate_init = torch.from_numpy(init).pin_memory()
            blk = np.zeros(self._host.numel(), dtype=np.uint8)               # [tokens | pos | anc] before the first step: <sos> in every dialogue's row 0
            tok = blk[:8 * W].view(np.int64)
            tok[:] = self.pad
            tok[::self.width] = start
            blk[self._off_anc:].view(np.int32).reshape(W, Lm)[:] = np.arange(W, dtype=np.int32)[:, None]
            self._blk_init = torch.from_numpy(blk).pin_memory()
            # the step log: [parent | tok (int32 L x W each) | n_old | n_new (int32 L x D each) | score | done (double L x W each) | flags (2 x int32)]
            o_par, o_tok = 0, 4 * Lm * W
            o_nold, o_nnew = 8 * Lm * W, 8 * Lm * W + 4 * Lm * D
            o_sc = (8 * Lm * W + 8 * Lm * D + 7) // 8 * 8
            o_done = o_sc + 8 * Lm * W
            o_flags = o_done + 8 * Lm * W
            self._log = torch.zeros(o_flags + 8, device=dev, dtype=torch.uint8)
            self._log_host = torch.zeros(o_flags + 8, dtype=torch.uint8).pin_memory()
            hb = self._log_host.numpy()
            self._log_views = (hb[o_par:o_tok].view(np.int32).reshape(Lm, W), hb[o_tok:o_nold].view(np.int32).reshape(Lm, W),
                               hb[o_nold:o_nnew].view(np.int32).reshape(Lm, D), hb[o_nnew:o_nnew + 4 * Lm * D].view(np.int32).reshape(Lm, D),
                               hb[o_sc:o_done].view(np.float64).reshape(Lm, W), hb[o_done:o_flags].view(np.float64).reshape(Lm, W),
                               hb[o_flags:o_flags + 8].view(np.int32))
            a = L.BeamArgs()
            a.dialogues, a.width, a.L, a.k_top, a.k, a.beam, a.unk, a.eos, a.pad, a.min_len = D, self.width, Lm, k_top, k, beam, unk, eos, self.pad, min_len
            a.penalty = float(penalty)
            p0, s0, l0 = self._devblk.data_ptr(), self._bstate.data_ptr(), self._log.data_ptr()
            a.tokens, a.pos, a.anc = p0, p0 + self._off_pos, p0 + self._off_anc
            a.lp, a.n_live, a.step, a.flags = s0, s0 + 8 * W, s0 + 8 * W + 4 * D, l0 + o_flags
            a.log_parent, a.log_tok, a.log_n_old, a.log_n_new, a.log_score, a.log_done = l0 + o_par, l0 + o_tok, l0 + o_nold, l0 + o_nnew, l0 + o_sc, l0 + o_done
            self._beam_args = a

            def body():
                self._devblk.copy_(self._blk_init, non_blocking=True)
                self._bstate.copy_(self._bstate_init, non_blocking=True)
                self._log[o_flags:].zero_()
                g = self.model.generator._fused
                for _ in range(Lm):
                    L.check(L.load().mtn_decode_step(C.byref(self._args), self._stages_dev.data_ptr(), self._grid, L.stream_ptr()))
                    logp = ops.generator_log_probs(self._out_lp, g["w_lp"], g["bias"])
                    top = ops.topk_rows(logp, k_top, extra_col)
                    a.top = top.data_ptr()
                    L.check(L.load().mtn_beam_advance(C.byref(a), L.stream_ptr()))
                self._log_host.copy_(self._log, non_blocking=True)

            with torch.no_grad():
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    body()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                self._search_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._search_graph):
                    body()
            self._search_key = key
        self._search_graph.replay()
        self._prev = None
        torch.cuda.current_stream().synchronize()
        if self.timed_out():                 # the log is garbage: the caller sees timed_out() and falls back
            return None
        return None if self._log_views[6][0] else self._log_views

    def timed_out(self) -> bool:
        """True if a poll of any step since the session was built (or last recovered) timed out: the results since then are garbage."""
        return int(self._sync[1].item()) != 0

    def recover(self):
        """After a timeout: nothing advanced the launch generation and the granule buffers hold tags of the failed step.  Advance the
        generation past it, clear the timeout flag and the check-in counter — the next launch starts clean."""
        torch.cuda.synchronize()
        gen = int(self._sync[0].item())
        self._sync.copy_(torch.tensor([gen + 1, 0, 0, 0], dtype=torch.int32))
        self._prev = None
        torch.cuda.synchronize()

    def check(self):
        """Raises if a poll of any step since the session was built timed out (for callers that drive step() / step_extend() themselves;
        beam_search_decode / greedy_decode fall back to the launch-per-sublayer pass instead)."""
        if self.timed_out():
            raise RuntimeError("mtn_decode_step: a poll timed out (not every workgroup of the launch was resident?)")

    def step_extend(self, l: int, slots, tokens, parents):
        """One step given the live hypotheses directly: hypothesis in row ``slots[i]`` ends in ``tokens[i]`` (its l-th token) and extends
        the hypothesis that sat in row ``parents[i]`` at the previous step (ignored at l = 1).  What step_many() derives from prefix
        lists, without building or searching them: the beam search below knows every candidate's parent."""
        if l > 1:
            self._h_anc[slots, :l - 1] = self._h_anc[parents, :l - 1]          # (the right-hand side is gathered into a copy first)
        self._h_anc[slots, l - 1] = slots
        self._h_tok[:] = self.pad
        self._h_tok[slots] = tokens
        self._h_pos[0] = l - 1
        self._prev = None
        self._run()

    def _run(self):
        with torch.no_grad():
            if not self.use_graph:
                self._pass_mega()
            else:
                if self._graph is None:
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        self._pass_mega()
                    torch.cuda.current_stream().wait_stream(side)
                    torch.cuda.synchronize()
                    self._graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self._graph):
                        self._pass_mega()
                self._graph.replay()

    def step_many(self, prefix_lists):
        if len(prefix_lists) != self.D:
            raise ValueError("one prefix list per dialogue of the session")
        l = len(prefix_lists[0][0])
        W = self.width
        if max(len(p) for p in prefix_lists) > W or l > self.max_len:
            raise ValueError("more hypotheses / longer prefix than the session was built for")
        anc_old = self._h_anc.copy() if l > 1 else None
        self._h_tok[:] = self.pad
        for d_, prefixes in enumerate(prefix_lists):
            for i, p in enumerate(prefixes):
                j = d_ * W + i
                self._h_tok[j] = int(p[-1])
                if l > 1:
                    parent = d_ * W + self._prev[d_].index(list(p[:-1]))          # the hypothesis this one extends
                    self._h_anc[j, :l - 1] = anc_old[parent, :l - 1]
                self._h_anc[j, l - 1] = j
        self._prev = [[list(p) for p in prefixes] for prefixes in prefix_lists]
        self._h_pos[0] = l - 1
        self._run()
        return [self.logp[d_ * W:d_ * W + len(p)] for d_, p in enumerate(prefix_lists)]


_SESSIONS: dict = {}
SELECT_MAX_K = 16        # csrc/select.hip SEL_MAX_K
KV_CACHE_FROM = 32      # prefix K/V cache by default for searches longer than this (at the reference's max_len = 20 the full-prefix
                        # pass is launch-latency-bound and as fast; the cache makes a token cost O(l) instead of O(l^2) work beyond it)


def _session(model, batch, max_len, width, pad, use_graph, kv_cache=False, select=None, mega=False) -> DecodeSession:
    """Sessions are kept per (model, shapes): a dialogue with the shapes of an earlier one reuses its buffers and graph.
    The cache is dropped when the model's weights change (prepare() version) or it grows past a few shapes."""
    model.prepare()
    mega = bool(mega) and MegaDecodeSession.supported(model, batch, max_len, width)
    key = DecodeSession.signature(model, batch, max_len, width) + (bool(use_graph), bool(kv_cache), select, mega)
    ver = getattr(model, "_flat_version", None)
    hit = _SESSIONS.get(key)
    if hit is not None and hit[1] == ver and hit[0].model is model:
        hit[0].load(batch)
        return hit[0]
    if len(_SESSIONS) >= 8:
        _SESSIONS.clear()
    if mega:
        sess = MegaDecodeSession(model, batch, max_len, width, pad=pad, use_graph=use_graph, select=select)
    else:
        sess = DecodeSession(model, batch, max_len, width, pad=pad, use_graph=use_graph, kv_cache=bool(kv_cache), select=select)
    _SESSIONS[key] = (sess, ver)
    return sess


class _Beam:
    """Hypothesis bookkeeping of data_utils.py:196-240 for one dialogue (same candidate order and tie behaviour)."""

    def __init__(self, start_symbol, unk_symbol, end_symbol, beam, penalty, min_len):
        self.unk, self.eos, self.beam, self.penalty, self.min_len = unk_symbol, end_symbol, beam, penalty, min_len
        self.hyps = [([], 0.0, [start_symbol])]
        self.parents = [0]                        # per hypothesis: the index (in the previous step's list) of the one it extends
        self.best, self.done = None, []

    def prefixes(self):
        return [h[2] for h in self.hyps]

    def advance(self, logp, l, top=None):
        """logp: (n_hyp, V) log-probabilities on the host — or None when ``top`` = (values (n,k), indices (n,k), eos column (n,))
        carries each row's k = beam+2 best entries in descending order: at most `beam` candidates per hypothesis can enter the
        new beam and at most two (<unk>, <eos>) are skipped, so the rest of the vocabulary is never looked at."""
        import numpy as np
        new, par, argmin = [], [], 0
        for h, (out, lp, st) in enumerate(self.hyps):
            if top is None:
                lp_vec = (logp[h] + lp).astype("float32")
                eos_val = float(lp_vec[self.eos])
                order = ((int(o), float(lp_vec[o])) for o in np.argsort(lp_vec)[::-1])
            else:
                vals = (top[0][h] + lp).astype("float32")
                eos_val = float(np.float32(top[2][h] + lp))
                order = ((int(o), float(v)) for o, v in zip(top[1][h], vals))
            if l >= self.min_len:
                s = eos_val + self.penalty * (len(out) + 1)
                self.done.append((out, s))
                if self.best is None or self.best < s:
                    self.best = s
            for o, s in order:
                if o == self.unk or o == self.eos:
                    continue
                if len(new) == self.beam:
                    if new[argmin][1] < s:
                        new[argmin] = (out + [o], s, st + [o])
                        par[argmin] = h
                        argmin = min(range(len(new)), key=lambda i: new[i][1])
                    else:
                        break
                else:
                    new.append((out + [o], s, st + [o]))
                    par.append(h)
                    if len(new) == self.beam:
                        argmin = min(range(len(new)), key=lambda i: new[i][1])
        self.hyps, self.parents = new, par

    def result(self, nbest):
        if self.done:
            return sorted(self.done, key=lambda h: -h[1])[:nbest], self.best
        return [([], 0)], None


def beam_search_decode_many(model, batch, max_len, start_symbol, unk_symbol, end_symbol, pad_symbol, beam=5, penalty=1.0,
                            nbest=5, min_len=1, use_graph=True, kv_cache=None):
    """beam_search_decode for a Batch of D dialogues at once: the D x beam live hypotheses are the batch dimension of ONE
    target-stream pass per generated token (the pass is launch-latency-bound, so D dialogues cost little more than one).
    Returns a list of D (n-best list, best score) pairs, each equal to what the single-dialogue search returns."""
    auto = kv_cache is None          # the caller leaves the pass to us: one persistent launch per token where it applies (<= 16 hypotheses, bf16)
    if kv_cache is None:
        kv_cache = max_len > KV_CACHE_FROM
    args = (model, batch, max_len, start_symbol, unk_symbol, end_symbol, pad_symbol, beam, penalty, nbest, min_len, use_graph, kv_cache)
    res = _beam_search_many(*args, mega=auto)
    if res is None:                  # a poll of the persistent step timed out (compute units held by another kernel): the launch-per-sublayer pass
        res = _beam_search_many(*args, mega=False)
    return res


def _mega_failed(sess) -> bool:
    """After a search on a persistent-step session: True (and the session is made consistent again) if one of its polls timed out."""
    if not sess.timed_out():
        return False
    import logging
    logging.getLogger("mtn_amd").warning("decode: the persistent step timed out (not every workgroup was resident); re-running on the launch-per-sublayer pass")
    sess.recover()
    MegaDecodeSession.FALLBACKS += 1
    return True


def _beam_search_many(model, batch, max_len, start_symbol, unk_symbol, end_symbol, pad_symbol, beam, penalty, nbest, min_len, use_graph, kv_cache, mega):
    auto = mega
    k = beam + 2
    # device-side candidate selection (csrc/select.hip) holds at most SELECT_MAX_K entries per row: wider beams keep torch.topk
    sel = (k + 1, end_symbol) if k + 1 <= SELECT_MAX_K else None
    sess = _session(model, batch, max_len, beam, pad_symbol, use_graph, kv_cache, select=sel, mega=auto)
    mega = isinstance(sess, MegaDecodeSession)
    if mega and sel is not None:
        # the whole search as one graph replay, hypothesis bookkeeping on the device; None = a tie somewhere: step by step below
        res = sess.search(beam, k, start_symbol, unk_symbol, end_symbol, penalty, min_len, nbest)
        if _mega_failed(sess):
            return None
        if res is not None:
            return res
    beams = [_Beam(start_symbol, unk_symbol, end_symbol, beam, penalty, min_len) for _ in range(sess.D)]
    for l in range(max_len):
        if mega:
            # the persistent step takes (row, newest token, parent row) of every live hypothesis: no prefix lists are built or searched
            Wd = sess.width
            live = [d * Wd + i for d, bm in enumerate(beams) for i in range(len(bm.hyps))]
            sess.step_extend(l + 1, live, [h[2][-1] for bm in beams for h in bm.hyps],
                             [d * Wd + p for d, bm in enumerate(beams) for p in bm.parents])
            logps = [sess.logp[d * Wd:d * Wd + len(bm.hyps)] for d, bm in enumerate(beams)] if sess.top is None else None
            counts = [len(bm.hyps) for bm in beams]
        else:
            logps = sess.step_many([bm.prefixes() for bm in beams])
            counts = [lp.size(0) for lp in logps]
            live = [r for d, n in enumerate(counts) for r in range(d * sess.width, d * sess.width + n)]
        if sess.top is not None:
            # device-side selection inside the pass (csrc/select.hip): only the heads of the rows travel, in one copy
            full = (sess.top_host() if mega else sess.top.cpu().numpy()).astype("float64")
            packed = full[live]
            kk = (packed.shape[1] - 1) // 2
            allp = None
        else:
            allp = torch.cat(logps, 0)
            tv, ti = torch.topk(allp, min(k + 1, allp.size(1)), dim=-1)
            packed = torch.cat([tv.double(), ti.double(), allp[:, end_symbol:end_symbol + 1].double()], 1).cpu().numpy()
            kk = tv.size(1)
        vals, idx, eos = packed[:, :kk], packed[:, kk:2 * kk].astype("int64"), packed[:, 2 * kk]
        # exact ties inside a row's head would make the visiting order depend on the selection algorithm: the reference's
        # order (argsort, data_utils.py:219) is then taken from the full row
        tie = bool((vals[:, 1:] == vals[:, :-1]).any())
        host = (allp if allp is not None else sess.logp[live]).double().cpu().numpy() if tie else None
        o = 0
        for bm, n in zip(beams, counts):
            if tie:
                bm.advance(host[o:o + n], l)
            else:
                bm.advance(None, l, top=(vals[o:o + n, :k], idx[o:o + n, :k], eos[o:o + n]))
            o += n
    if mega and _mega_failed(sess):
        return None
    return [bm.result(nbest) for bm in beams]


def beam_search_decode(model, batch, max_len, start_symbol, unk_symbol, end_symbol, pad_symbol, beam=5, penalty=1.0,
                       nbest=5, min_len=1, use_graph=True, kv_cache=None):
    """data_utils.py:188-242, same arguments and return value: (n-best list of (token list, score) sorted by score,
    best score).  A hypothesis ending with <eos> at length k scores logp + penalty * k; <unk> and <eos> never extend
    a hypothesis; candidates are visited in descending log-probability exactly as the reference does (data_utils.py:219)."""
    if batch.query.size(0) != 1:
        raise ValueError("beam_search_decode works on one dialogue (data_utils.py:188); use beam_search_decode_many for a batch")
    return beam_search_decode_many(model, batch, max_len, start_symbol, unk_symbol, end_symbol, pad_symbol, beam, penalty, nbest,
                                   min_len, use_graph, kv_cache)[0]


def greedy_decode(model, batch, max_len, start_symbol, pad_symbol=1, use_graph=True, kv_cache=None):
    """data_utils.py:159-186 (the reference's own greedy_decode cannot run: it calls decode() with the wrong arity, SURVEY
    §8c) — pinned to: argmax of the generator's log-probabilities at every step, (1, max_len) tokens incl. <sos>."""
    auto = kv_cache is None
    ys = _greedy(model, batch, max_len, start_symbol, pad_symbol, use_graph, kv_cache, auto)
    if ys is None:                   # the persistent step timed out: the launch-per-sublayer pass
        ys = _greedy(model, batch, max_len, start_symbol, pad_symbol, use_graph, kv_cache, False)
    return torch.tensor([ys], dtype=batch.query.dtype, device=batch.query.device)


def _greedy(model, batch, max_len, start_symbol, pad_symbol, use_graph, kv_cache, auto):
    sess = _session(model, batch, max_len, 1, pad_symbol, use_graph, max_len > KV_CACHE_FROM if kv_cache is None else kv_cache,
                    select=(2, 0) if auto else None, mega=auto)
    ys = [start_symbol]
    if isinstance(sess, MegaDecodeSession) and sess.select is not None:
        toks = sess.greedy(start_symbol)                   # the whole decode as one graph replay (None: a tie in some row's head)
        if toks is None:
            # step by step: the row's head (largest log-probability, its column) arrives in a pinned block — one replay and one
            # stream synchronisation per token, no argmax launch, no .item()
            for l in range(1, max_len):
                sess.step_extend(l, [0], [ys[-1]], [0])
                ys.append(int(sess.top_host()[0, 2]))
        else:
            ys += toks
        return None if _mega_failed(sess) else ys
    for _ in range(max_len - 1):
        nxt = int(sess.step([ys]).argmax(dim=-1)[0])
        ys.append(nxt)
    return ys
