"""CPU oracle for the MTN hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (PyTorch-CPU fp32 tensor algebra,
functional style, no nn.Module) of the arithmetic the reference performs on its
transformer hot path.  It exists to *check* the HIP path; it is never the thing
shipped or measured.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product package
(``mtn_amd``) must never import anything from ``oracle/``.

Parity pin: the reference (henryhungle/MTN) has no tests and no golden vectors of
its own, so this restatement is pinned against outputs of the
reference itself, produced in the build container by ``oracle/make_golden.py``
(which imports ``/root/reference/mtn.py`` on CPU) and committed under
``tests/golden/``.  ``tests/test_oracle_golden.py`` replays them.

Every function cites the reference lines it restates (paths relative to
/root/reference).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor


# --------------------------------------------------------------------------- config
@dataclass
class OracleConfig:
    """Constructor arguments of ``make_model`` (mtn.py:332-337) that shape the math."""
    vocab: int
    n_layers: int = 6
    d_model: int = 512
    d_ff: int = 2048
    heads: int = 8
    ft_sizes: Sequence[int] = (2048, 128)
    diff_encoder: bool = True
    diff_embed: bool = False
    diff_gen: bool = False
    auto_encoder_ft: str = "query"
    ln_eps: float = 1e-6

    @property
    def n_ft(self) -> int:
        return len(self.ft_sizes)


# --------------------------------------------------------------------------- leaf ops
def layer_norm(x: Tensor, a2: Tensor, b2: Tensor, eps: float = 1e-6) -> Tensor:
    """mtn.py:111-114 — NOT nn.LayerNorm: unbiased std (÷(d-1)), eps added to std."""
    d = x.shape[-1]
    mean = x.sum(-1, keepdim=True) / d
    xc = x - mean
    var_unbiased = (xc * xc).sum(-1, keepdim=True) / (d - 1)
    std = torch.sqrt(var_unbiased)
    return a2 * xc / (std + eps) + b2


def linear(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """nn.Linear as used at mtn.py:243-244,273-276: y = x W^T + b, W is [out,in]."""
    return x @ w.t() + b


def scaled_dot_attention(q: Tensor, k: Tensor, v: Tensor, mask: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    """mtn.py:221-231 (dropout omitted: oracle runs the eval()/dropout=0 path).

    q (B,h,a,dk), k/v (B,h,m,dk), mask broadcastable to (B,h,a,m) with 0 = masked.
    Masked scores are set to -1e9 (not -inf): a fully masked row becomes uniform.
    """
    dk = q.shape[-1]
    scores = (q @ k.transpose(-2, -1)) / math.sqrt(dk)
    if mask is not None:
        scores = torch.where(mask == 0, torch.full_like(scores, -1e9), scores)
    p = torch.softmax(scores, dim=-1)
    return p @ v, p


def multi_head_attention(query: Tensor, key: Tensor, value: Tensor, mask: Optional[Tensor],
                         w: Sequence[Tensor], b: Sequence[Tensor], heads: int) -> Tensor:
    """mtn.py:248-267.  w/b are the four ``linears`` (q,k,v,out)."""
    if mask is not None:
        mask = mask.unsqueeze(1)                      # same mask for every head (mtn.py:252)
    B, d = query.shape[0], query.shape[-1]
    dk = d // heads

    def split(x: Tensor) -> Tensor:                   # mtn.py:257
        return x.reshape(B, -1, heads, dk).transpose(1, 2)

    q = split(linear(query, w[0], b[0]))
    k = split(linear(key, w[1], b[1]))
    v = split(linear(value, w[2], b[2]))
    o, _ = scaled_dot_attention(q, k, v, mask)
    o = o.transpose(1, 2).reshape(B, -1, d)           # mtn.py:265-266
    return linear(o, w[3], b[3])


def feed_forward(x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor) -> Tensor:
    """mtn.py:279-280 (dropout omitted)."""
    return linear(torch.relu(linear(x, w1, b1)), w2, b2)


def positional_encoding(length: int, d_model: int, dtype=torch.float32) -> Tensor:
    """mtn.py:298-303 — sinusoid table rows [0,length)."""
    pe = torch.zeros(length, d_model, dtype=dtype)
    position = torch.arange(0.0, length, dtype=dtype).unsqueeze(1)
    div_term = torch.exp(torch.arange(0.0, d_model, 2, dtype=dtype) * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def subsequent_mask(size: int) -> Tensor:
    """data_utils.py:10-14 — (1,size,size) bool, True on and below the diagonal."""
    return torch.tril(torch.ones(1, size, size, dtype=torch.bool))


def make_std_mask(tgt: Tensor, pad: int) -> Tensor:
    """data_utils.py:48-54."""
    return (tgt != pad).unsqueeze(-2) & subsequent_mask(tgt.size(-1))


def feature_mask_and_clean(ft: Tensor) -> Tuple[Tensor, Tensor]:
    """data_utils.py:28-30 — frames whose every element == 1.0 are padding; they are
    masked out and zeroed.  ft is (B,V,F) already permuted to batch-first."""
    mask = ((ft != 1).sum(dim=2) != 0).unsqueeze(-2)          # (B,1,V) bool
    clean = ft * mask.squeeze(-2).unsqueeze(-1).to(ft.dtype)
    return clean, mask


# --------------------------------------------------------------------------- batch
@dataclass
class OracleBatch:
    """Field names/shapes of data_utils.py:21-46 (no .cuda())."""
    query: Tensor
    his: Tensor
    cap: Tensor
    trg: Tensor
    trg_y: Tensor
    fts: List[Tensor]
    pad: int = 1
    query_mask: Tensor = field(init=False)
    his_mask: Tensor = field(init=False)
    cap_mask: Tensor = field(init=False)
    trg_mask: Tensor = field(init=False)
    fts_mask: List[Tensor] = field(init=False)
    ntokens: Tensor = field(init=False)

    def __post_init__(self):
        cleaned, masks = [], []
        for ft in self.fts:
            c, m = feature_mask_and_clean(ft)
            cleaned.append(c)
            masks.append(m)
        self.fts, self.fts_mask = cleaned, masks
        self.query_mask = (self.query != self.pad).unsqueeze(-2)
        self.his_mask = (self.his != self.pad).unsqueeze(-2)
        self.cap_mask = (self.cap != self.pad).unsqueeze(-2)
        self.trg_mask = make_std_mask(self.trg, self.pad)
        self.ntokens = (self.trg_y != self.pad).sum()


# --------------------------------------------------------------------------- model
class OracleMTN:
    """Functional MTN over a reference-schema ``state_dict``."""

    def __init__(self, cfg: OracleConfig, state: Dict[str, Tensor]):
        self.cfg = cfg
        self.p = {k: v for k, v in state.items()}
        # per-sublayer taps for fixture comparison: name -> tensor
        self.taps: Optional[Dict[str, Tensor]] = None

    # ---- helpers
    def _tap(self, name: str, t: Tensor):
        if self.taps is not None:
            self.taps[name] = t.detach().clone()

    def _ln(self, prefix: str, x: Tensor) -> Tensor:
        return layer_norm(x, self.p[prefix + ".a_2"], self.p[prefix + ".b_2"], self.cfg.ln_eps)

    def _mha(self, prefix: str, q: Tensor, kv: Tensor, mask: Tensor) -> Tensor:
        w = [self.p[f"{prefix}.linears.{i}.weight"] for i in range(4)]
        b = [self.p[f"{prefix}.linears.{i}.bias"] for i in range(4)]
        return multi_head_attention(q, kv, kv, mask, w, b, self.cfg.heads)

    def _ffn(self, prefix: str, x: Tensor) -> Tensor:
        return feed_forward(x, self.p[prefix + ".w_1.weight"], self.p[prefix + ".w_1.bias"],
                            self.p[prefix + ".w_2.weight"], self.p[prefix + ".w_2.bias"])

    def embed(self, which: str, tokens: Tensor) -> Tensor:
        """Embeddings + PositionalEncoding, mtn.py:289,308 (``which`` = 'query_embed' etc.)."""
        d = self.cfg.d_model
        e = self.p[which + ".0.lut.weight"][tokens] * math.sqrt(d)
        return e + positional_encoding(tokens.shape[1], d, e.dtype).unsqueeze(0)

    def vid_encode(self, fts: Sequence[Tensor]) -> List[Tensor]:
        """mtn.py:32-36 with the Sequential of mtn.py:378: Linear -> ReLU -> PE."""
        out = []
        for i, ft in enumerate(fts):
            y = torch.relu(linear(ft, self.p[f"vid_encoder.{i}.0.weight"], self.p[f"vid_encoder.{i}.0.bias"]))
            out.append(y + positional_encoding(ft.shape[1], self.cfg.d_model, y.dtype).unsqueeze(0))
        return out

    def encode(self, query, query_mask, his, his_mask, cap, cap_mask, vid, vid_mask):
        """mtn.py:38-56 + Encoder.forward (LayerNorm bank, mtn.py:83-101).

        Returns [q_mem, [vid_mem_i], cap_mem, his_mem, ae] with ae = list or None.
        Every text stream goes through ``query_embed`` (mtn.py:52)."""
        c = self.cfg
        n = 0

        def norm(x):
            nonlocal n
            y = self._ln(f"query_encoder.norm.{n}", x)
            n += 1
            return y

        q = norm(self.embed("query_embed", query))
        v = [norm(x) for x in self.vid_encode(vid)]
        cp = norm(self.embed("query_embed", cap))
        hs = norm(self.embed("query_embed", his))
        if not c.diff_encoder:
            return [q, v, cp, hs, None]
        ft = cap if c.auto_encoder_ft in ("caption", "summary") else query
        ae = []
        for i in range(len(vid)):
            src = f"auto_encoder_embed.{i}" if c.diff_embed else "query_embed"
            ae.append(norm(self.embed(src, ft)))
        return [q, v, cp, hs, ae]

    def _sublayer(self, prefix: str, k: int, x: Tensor, fn) -> Tensor:
        """SublayerConnection.forward mtn.py:125-127 (pre-norm residual, dropout omitted)."""
        y = x + fn(self._ln(f"{prefix}.sublayer.{k}.norm", x))
        self._tap(f"{prefix}.sublayer.{k}", y)
        return y

    def decoder_layer(self, n: int, x, cap_mem, cap_mask, his_mem, his_mask, q_mem, q_mask,
                      tgt_mask, vid_fts, vid_mask, ae_fts):
        """DecoderLayer.forward mtn.py:181-218; schedule of SURVEY.md §3.2."""
        L = f"decoder.layers.{n}"
        mode = self.cfg.auto_encoder_ft
        k = 0
        x = self._sublayer(L, k, x, lambda y: self._mha(L + ".self_attn", y, y, tgt_mask)); k += 1
        x = self._sublayer(L, k, x, lambda y: self._mha(L + ".his_attn", y, his_mem, his_mask)); k += 1
        if mode in ("caption", "summary"):
            x = self._sublayer(L, k, x, lambda y: self._mha(L + ".src_attn", y, q_mem, q_mask)); k += 1
            x = self._sublayer(L, k, x, lambda y: self._mha(L + ".cap_attn", y, cap_mem, cap_mask)); k += 1
            if ae_fts is None:
                ae_fts = cap_mem
            ae_mask = cap_mask
        elif mode == "query":
            x = self._sublayer(L, k, x, lambda y: self._mha(L + ".cap_attn", y, cap_mem, cap_mask)); k += 1
            x = self._sublayer(L, k, x, lambda y: self._mha(L + ".src_attn", y, q_mem, q_mask)); k += 1
            if ae_fts is None:
                ae_fts = q_mem
            ae_mask = q_mask
        else:
            raise ValueError("auto_encoder_ft must be query|caption|summary (mtn.py:187-202)")
        out_ae = []
        for i, vid in enumerate(vid_fts):
            ae = ae_fts[i] if isinstance(ae_fts, list) else ae_fts
            ae = self._sublayer(L, k, ae, lambda y: self._mha(f"{L}.auto_encoder_self_attn.{i}", y, y, ae_mask)); k += 1
            ae = self._sublayer(L, k, ae, lambda y: self._mha(f"{L}.auto_encoder_vid_attn.{i}", y, vid, vid_mask[i])); k += 1
            ae = self._sublayer(L, k, ae, lambda y: self._ffn(f"{L}.auto_encoder_feed_forward.{i}", y)); k += 1
            ae_now = ae
            x = self._sublayer(L, k, x, lambda y: self._mha(f"{L}.auto_encoder_attn.{i}", y, ae_now, ae_mask)); k += 1
            out_ae.append(ae)
        x = self._sublayer(L, k, x, lambda y: self._ffn(L + ".feed_forward", y))
        return x, out_ae

    def decode(self, vid_mem, his_mem, cap_mem, q_mem, vid_mask, his_mask, cap_mask, q_mask, tgt, tgt_mask, ae):
        """EncoderDecoder.decode mtn.py:58-60 + Decoder.forward mtn.py:158-164."""
        x = self.embed("tgt_embed", tgt)
        for n in range(self.cfg.n_layers):
            x, ae = self.decoder_layer(n, x, cap_mem, cap_mask, his_mem, his_mask, q_mem, q_mask,
                                       tgt_mask, vid_mem, vid_mask, ae)
        out_ae = [self._ln(f"decoder.ae_norm.{i}", a) for i, a in enumerate(ae)]
        return self._ln("decoder.norm", x), out_ae

    def forward(self, b) -> Tuple[Tensor, List[Tensor]]:
        """EncoderDecoder.forward mtn.py:28-30."""
        q, v, cp, hs, ae = self.encode(b.query, b.query_mask, b.his, b.his_mask, b.cap, b.cap_mask, b.fts, b.fts_mask)
        return self.decode(v, hs, cp, q, b.fts_mask, b.his_mask, b.cap_mask, b.query_mask, b.trg, b.trg_mask, ae)

    def generator(self, x: Tensor, which: str = "generator") -> Tensor:
        """Generator.forward mtn.py:68-69."""
        return torch.log_softmax(linear(x, self.p[which + ".proj.weight"], self.p[which + ".proj.bias"]), dim=-1)

    # ---- loss (harness, data_utils.py:123-156 + label_smoothing.py:9-32)
    def loss(self, b, out: Tensor, ae_out: List[Tensor], smoothing: float = 0.1, lam: float = 1.0,
             norm: Optional[Tensor] = None, ae_norm: Optional[Tensor] = None) -> Tensor:
        """Value that SimpleLossCompute calls .backward() on (data_utils.py:133-153)."""
        c = self.cfg
        pad = b.pad
        if c.auto_encoder_ft in ("caption", "summary"):
            ae_y = b.cap                                               # train.py:35-36
        else:
            ae_y = b.query                                             # train.py:38-39
        if norm is None:
            norm = b.ntokens
        if ae_norm is None:
            ae_norm = (ae_y != pad).sum()
        lp = self.generator(out)
        total = label_smoothing_kl(lp.reshape(-1, lp.shape[-1]), b.trg_y.reshape(-1), pad, smoothing) / norm.float()
        for i, a in enumerate(ae_out):
            which = f"auto_encoder_generator.{i}" if c.diff_gen else "generator"
            lpa = self.generator(a, which)
            total = total + lam * label_smoothing_kl(lpa.reshape(-1, lpa.shape[-1]), ae_y.reshape(-1), pad, smoothing) / ae_norm.float()
        return total


def label_smoothing_kl(logp: Tensor, target: Tensor, pad: int, smoothing: float) -> Tensor:
    """label_smoothing.py:20-32 — KLDivLoss(sum) against the smoothed one-hot.

    Quirk kept on purpose (label_smoothing.py:29): rows whose target is <pad> are zeroed
    only ``if mask.sum() > 0`` where ``mask`` holds the *indices* of padded rows, so a
    single padded row at flat index 0 is NOT zeroed."""
    size = logp.shape[1]
    true_dist = torch.full_like(logp, smoothing / (size - 2))
    true_dist.scatter_(1, target.unsqueeze(1), 1.0 - smoothing)
    true_dist[:, pad] = 0
    idx = torch.nonzero(target == pad)
    if idx.numel() > 0 and int(idx.sum()) > 0:
        true_dist.index_fill_(0, idx.reshape(-1), 0.0)
    true_dist = true_dist.detach()
    # KLDivLoss(reduction='sum'): sum t*(log t - x), with 0*log0 := 0
    pos = true_dist > 0
    safe = torch.where(pos, true_dist, torch.ones_like(true_dist))
    return (true_dist * (torch.log(safe) - logp)).sum()


def noam_rate(step: int, d_model: int, warmup: int, factor: float = 1.0) -> float:
    """data_utils.py:111-117."""
    return factor * (d_model ** -0.5) * min(step ** -0.5, step * warmup ** -1.5)


# --------------------------------------------------------------------------- decode
def beam_search(model: OracleMTN, b, max_len: int, sos: int, unk: int, eos: int,
                beam: int = 5, penalty: float = 1.0, nbest: int = 5, min_len: int = 1):
    """data_utils.py:188-242 restated (batch of ONE dialogue; full-prefix re-decode, no cache).

    Returns (list of (token list, score)) sorted by score desc, best completed score)."""
    q, v, cp, hs, ae = model.encode(b.query, b.query_mask, b.his, b.his_mask, b.cap, b.cap_mask, b.fts, b.fts_mask)
    hyps = [([], 0.0, torch.full((1, 1), sos, dtype=b.query.dtype))]
    best = None
    done: List[Tuple[List[int], float]] = []
    for l in range(max_len):
        new: List[Tuple[List[int], float, Tensor]] = []
        argmin = 0
        for out, lp, st in hyps:
            x, _ = model.decode(v, hs, cp, q, b.fts_mask, b.his_mask, b.cap_mask, b.query_mask,
                                st, subsequent_mask(st.size(1)), ae)
            lp_vec = (model.generator(x[:, -1]).reshape(-1) + lp).double().numpy().astype("float32")
            if l >= min_len:
                s = float(lp_vec[eos]) + penalty * (len(out) + 1)
                done.append((out, s))
                if best is None or best < s:
                    best = s
            for o in np.argsort(lp_vec)[::-1]:                                   # data_utils.py:219
                o = int(o)
                if o == unk or o == eos:
                    continue
                s = float(lp_vec[o])
                if len(new) == beam:
                    if new[argmin][1] < s:
                        new[argmin] = (out + [o], s, torch.cat([st, torch.full((1, 1), o, dtype=st.dtype)], dim=1))
                        argmin = min(range(len(new)), key=lambda i: new[i][1])
                    else:
                        break
                else:
                    new.append((out + [o], s, torch.cat([st, torch.full((1, 1), o, dtype=st.dtype)], dim=1)))
                    if len(new) == beam:
                        argmin = min(range(len(new)), key=lambda i: new[i][1])
        hyps = new
    if done:
        return sorted(done, key=lambda h: -h[1])[:nbest], best
    return [([], 0)], None


def greedy_search(model: OracleMTN, b, max_len: int, sos: int) -> List[int]:
    """data_utils.py:163-186 with decode() called at its real arity (the reference's own call has one argument too many and
    cannot run): argmax of generator(decode(prefix)[:, -1]) per step; returns the token list incl. <sos>."""
    q, v, cp, hs, ae = model.encode(b.query, b.query_mask, b.his, b.his_mask, b.cap, b.cap_mask, b.fts, b.fts_mask)
    ys = [sos]
    for _ in range(max_len - 1):
        st = torch.tensor([ys], dtype=b.query.dtype)
        x, _ = model.decode(v, hs, cp, q, b.fts_mask, b.his_mask, b.cap_mask, b.query_mask, st, subsequent_mask(st.size(1)), ae)
        ys.append(int(model.generator(x[:, -1]).argmax(dim=-1)[0]))
    return ys
