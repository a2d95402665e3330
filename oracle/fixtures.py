"""Deterministic weights and synthetic batches for parity fixtures.  TEST INFRASTRUCTURE ONLY.

The golden files under ``tests/golden`` hold only *outputs* of the reference.  Weights and
inputs are regenerated bit-identically on both sides from the formulas below
(``numpy.random.RandomState`` has a frozen stream), so fixtures stay small.
"""
from __future__ import annotations

import zlib
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

PAD, UNK, SOS, EOS = 1, 0, 2, 3          # data_handler.py:46 vocabulary specials


def _rs(key: str, seed: int) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(key.encode()) + 7919 * seed) % (2 ** 31))


def det_param(key: str, shape: Sequence[int], seed: int = 0) -> torch.Tensor:
    """Value for state_dict entry ``key`` (reference key schema)."""
    shape = tuple(int(s) for s in shape)
    r = _rs(key, seed)
    if key.endswith(".a_2"):
        v = 1.0 + 0.1 * r.standard_normal(shape)
    elif key.endswith(".b_2"):
        v = 0.05 * r.standard_normal(shape)
    elif key.endswith(".bias"):
        v = 0.02 * r.standard_normal(shape)
    elif len(shape) == 2:                               # Linear weight / embedding table: xavier-like scale
        v = r.standard_normal(shape) * np.sqrt(2.0 / (shape[0] + shape[1]))
        if ".lut." in key:
            v = r.standard_normal(shape) * (1.0 / np.sqrt(shape[1]))
    else:
        v = 0.1 * r.standard_normal(shape)
    return torch.from_numpy(v.astype(np.float32))


def det_state_dict(shapes: Dict[str, Sequence[int]], seed: int = 0) -> Dict[str, torch.Tensor]:
    """Deterministic replacement for every non-``pe`` entry of a state_dict."""
    return {k: det_param(k, s, seed) for k, s in shapes.items() if not k.endswith(".pe")}


def det_batch(vocab: int, B: int, Q: int, H: int, C: int, T: int, frames: Sequence[int],
              ft_sizes: Sequence[int], seed: int = 1, ragged: bool = True,
              blank_history_row: bool = True) -> Dict[str, object]:
    """Raw batch fields (int64 token matrices with <blank>=1 tails, frame features padded
    with 1.0 as data_handler.py:236 does).  Returns dict of numpy arrays."""
    r = np.random.RandomState(1000 + seed)

    def toks(L, min_len):
        x = r.randint(4, vocab, size=(B, L)).astype(np.int64)
        if ragged:
            lens = r.randint(min_len, L + 1, size=B)
            lens[0] = L                                  # at least one full-length row
            for i, n in enumerate(lens):
                x[i, n:] = PAD
        return x

    query = toks(Q, 2)
    his = toks(H, 1)
    if blank_history_row and ragged and B > 1:
        his[1, :] = PAD                                  # empty dialogue history -> uniform attention row
    cap = toks(C, 3)
    ans = toks(T + 1, 3)                                 # <sos> w1 .. wn <eos> layout
    ans[:, 0] = SOS
    trg, trg_y = ans[:, :-1].copy(), ans[:, 1:].copy()
    fts = []
    for V, F in zip(frames, ft_sizes):
        f = r.standard_normal((B, V, F)).astype(np.float32)
        if ragged:
            lens = r.randint(max(1, V // 2), V + 1, size=B)
            lens[0] = V
            for i, n in enumerate(lens):
                f[i, n:, :] = 1.0
        fts.append(f)
    return dict(query=query, his=his, cap=cap, trg=trg, trg_y=trg_y, fts=fts)


# Named parity configurations (cfg1 = BASELINE.json configs[0]; variants cover make_model flags).
GOLDEN_CONFIGS: Dict[str, dict] = {
    "cfg1_query": dict(vocab=100, N=2, d_model=128, d_ff=512, h=4, ft_sizes=[2048, 128], B=4, Q=20, H=20, C=20, T=20,
                       frames=[32, 32], diff_encoder=True, diff_embed=False, diff_gen=False, auto_encoder_ft="query"),
    "cfg1_caption": dict(vocab=100, N=2, d_model=128, d_ff=512, h=4, ft_sizes=[2048, 128], B=4, Q=20, H=20, C=20, T=20,
                         frames=[32, 32], diff_encoder=True, diff_embed=False, diff_gen=False, auto_encoder_ft="caption"),
    "small_shared": dict(vocab=60, N=2, d_model=64, d_ff=128, h=4, ft_sizes=[96, 32], B=3, Q=11, H=37, C=23, T=9,
                         frames=[17, 40], diff_encoder=False, diff_embed=False, diff_gen=False, auto_encoder_ft="query"),
    # d_model 512 / 8 heads (d_k = 64): the widths of BASELINE configs[1..4], so that the kernel instantiations the benchmark
    # runs (and the fused projection + attention kernel, which only exists at this width) are pinned to the REFERENCE
    "wide_n1": dict(vocab=96, N=1, d_model=512, d_ff=2048, h=8, ft_sizes=[64, 32], B=3, Q=12, H=24, C=16, T=12,
                    frames=[10, 6], diff_encoder=True, diff_embed=False, diff_gen=False, auto_encoder_ft="query"),
    "small_diffall": dict(vocab=60, N=1, d_model=64, d_ff=128, h=2, ft_sizes=[96], B=2, Q=7, H=5, C=13, T=6,
                          frames=[9], diff_encoder=True, diff_embed=True, diff_gen=True, auto_encoder_ft="summary"),
}


def state_shapes(vocab: int, N: int, d_model: int, d_ff: int, ft_sizes: Sequence[int], diff_encoder: bool,
                 diff_embed: bool, diff_gen: bool, **_unused) -> Dict[str, Tuple[int, ...]]:
    """Shapes of every learnable entry of the reference state_dict (SURVEY.md §3.3 key schema)."""
    d, F = d_model, len(ft_sizes)
    s: Dict[str, Tuple[int, ...]] = {}

    def lin(prefix, out_f, in_f):
        s[prefix + ".weight"] = (out_f, in_f)
        s[prefix + ".bias"] = (out_f,)

    def ln(prefix):
        s[prefix + ".a_2"] = (d,)
        s[prefix + ".b_2"] = (d,)

    def mha(prefix):
        for i in range(4):
            lin(f"{prefix}.linears.{i}", d, d)

    def ffn(prefix):
        lin(prefix + ".w_1", d_ff, d)
        lin(prefix + ".w_2", d, d_ff)

    s["query_embed.0.lut.weight"] = (vocab, d)
    s["tgt_embed.0.lut.weight"] = (vocab, d)
    if diff_embed:
        for i in range(F):
            s[f"auto_encoder_embed.{i}.0.lut.weight"] = (vocab, d)
    for j in range(3 + (2 * F if diff_encoder else F)):
        ln(f"query_encoder.norm.{j}")
    for i, ft in enumerate(ft_sizes):
        lin(f"vid_encoder.{i}.0", d, ft)
    for n in range(N):
        L = f"decoder.layers.{n}"
        for a in ("self_attn", "src_attn", "his_attn", "cap_attn"):
            mha(f"{L}.{a}")
        for i in range(F):
            mha(f"{L}.auto_encoder_self_attn.{i}")
            mha(f"{L}.auto_encoder_vid_attn.{i}")
            mha(f"{L}.auto_encoder_attn.{i}")
            ffn(f"{L}.auto_encoder_feed_forward.{i}")
        ffn(f"{L}.feed_forward")
        for k in range(5 + 4 * F):
            ln(f"{L}.sublayer.{k}.norm")
    ln("decoder.norm")
    for i in range(F):
        ln(f"decoder.ae_norm.{i}")
    lin("generator.proj", vocab, d)
    if diff_gen:
        for i in range(F):
            lin(f"auto_encoder_generator.{i}.proj", vocab, d)
    return s


def oracle_from_config(c: dict, seed: int = 0, requires_grad: bool = False):
    """(OracleMTN, OracleConfig) with deterministic weights for a GOLDEN_CONFIGS-style dict."""
    from oracle.mtn_oracle import OracleConfig, OracleMTN
    cfg = OracleConfig(vocab=c["vocab"], n_layers=c["N"], d_model=c["d_model"], d_ff=c["d_ff"], heads=c["h"],
                       ft_sizes=tuple(c["ft_sizes"]), diff_encoder=c["diff_encoder"], diff_embed=c["diff_embed"],
                       diff_gen=c["diff_gen"], auto_encoder_ft=c["auto_encoder_ft"])
    sd = det_state_dict(state_shapes(**c), seed)
    if requires_grad:
        sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    return OracleMTN(cfg, sd), cfg


def oracle_batch(raw: Dict[str, object]):
    from oracle.mtn_oracle import OracleBatch
    t = torch.from_numpy
    return OracleBatch(query=t(raw["query"]), his=t(raw["his"]), cap=t(raw["cap"]), trg=t(raw["trg"]),
                       trg_y=t(raw["trg_y"]), fts=[t(f) for f in raw["fts"]], pad=PAD)


def det_corpus(n_videos: int = 5, turns: int = 3, vocab: int = 60, ft_sizes: Sequence[int] = (16, 8), seed: int = 11,
               caption: bool = True) -> Dict[str, object]:
    """A deterministic miniature of what the reference's data_handler.load returns (data_handler.py:134-148): ``dialogs`` =
    [vid, qa_id, history, question, answer_in, answer_out(, caption)] with ragged int64 arrays, ``features`` = one dict per
    feature type mapping vid -> float32 [frames, ft] array (one frame of the first video is all ones: the reference treats
    such a frame as padding, data_utils.py:29)."""
    rs = np.random.RandomState(seed)
    tok = lambda lo, hi: rs.randint(4, vocab, size=rs.randint(lo, hi + 1)).astype(np.int64)
    dialogs, qa = [], 0
    vids = [f"vid{v:02d}" for v in range(n_videos)]
    for v in vids:
        cap = tok(3, 12)
        hist = np.zeros(0, np.int64)
        for t in range(turns):
            q, a = tok(2, 9), tok(2, 8)
            ans = np.concatenate([[SOS], a, [EOS]]).astype(np.int64)
            item = [v, qa, hist.copy() if len(hist) else np.array([PAD], np.int64), q, ans[:-1], ans[1:]]
            if caption:
                item.append(cap)
            dialogs.append(item)
            hist = np.concatenate([hist, q, a])
            qa += 1
    feats = []
    for fi, F in enumerate(ft_sizes):
        d = {}
        for k, v in enumerate(vids):
            arr = rs.randn(rs.randint(3, 10), F).astype(np.float32)
            if k == 0 and fi == 0:
                arr[1] = 1.0
            d[v] = arr
        feats.append(d)
    return {"dialogs": dialogs, "features": feats, "vocab": {"<blank>": PAD, "<unk>": UNK, "<sos>": SOS, "<eos>": EOS}}


def det_avsd_json(n_dialogs: int = 7, seed: int = 5) -> dict:
    """A deterministic miniature of the DSTC7-AVSD annotation file (the json schema data_handler.get_vocabulary / load read:
    dialogs[*] = {image_id, caption, summary, dialog: [{question, answer}]}).  Words come from a small pool so that some pass
    the reference's frequency cut (> 5 occurrences) and others become <unk>."""
    rs = np.random.RandomState(seed)
    common = "a the man woman is are in on room kitchen holding walks sits looks at book cup phone does he she yes no".split()
    rare = [f"rare{i}" for i in range(40)]

    def sent(lo, hi):
        n = rs.randint(lo, hi + 1)
        ws = [common[rs.randint(len(common))] if rs.rand() < 0.85 else rare[rs.randint(len(rare))] for _ in range(n)]
        return " ".join(ws)

    dialogs = []
    for d in range(n_dialogs):
        turns = [{"question": sent(2, 8) + " ?", "answer": sent(1, 9) + " ."} for _ in range(rs.randint(1, 6))]
        dialogs.append({"image_id": f"VID{d:02d}", "caption": sent(6, 16) + " .", "summary": sent(5, 12) + " .", "dialog": turns})
    return {"type": "mini", "version": "det", "dialogs": dialogs}
