"""Batch planning and device-side batch assembly (reference: data_handler.py:150-274, data_utils.py:23-54).

The reference keeps dialogs in Python lists and feature files on disk; every step it pads each field with numpy on the
host, `np.load`s the feature files, uploads, and derives masks with several torch passes (train.py:30).  On a GPU with
288 GB of HBM the whole corpus fits on the device, so here

* ``make_batch_indices`` plans the batches exactly as the reference does (length-sorted, batch size shrunk for long
  histories) — host logic, no device work;
* ``DeviceCorpus`` uploads the corpus ONCE: each token field as one flat int64 buffer + per-item start/length tables, each
  feature type as one flat [frames, F] float buffer + per-video tables;
* ``make_batch`` describes a batch by the ids of its items (one small H2D copy) and builds all padded tensors AND their
  masks with two grouped HIP launches (csrc/assemble.hip) — no host padding, no per-step file I/O, no mask passes.

``get_vocabulary`` / ``load`` read the DSTC7-AVSD json + per-video .npy features into that layout (host work, once per run;
reference data_handler.py:45-148): ``data`` is the dict the reference's ``load`` returns, except that the feature values are
the arrays themselves instead of (path, frames) pairs — the reference's planner accepts both (data_handler.py:158) and the
corpus goes to the device whole anyway.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Sequence

import numpy as np
import torch

from . import lib as L
from .data_utils import Batch

SPECIALS = {"<unk>": 0, "<blank>": 1, "<sos>": 2, "<eos>": 3}          # data_handler.py:46


def _caption_text(dialog: dict, include_caption: str):
    if include_caption in ("caption", "summary"):
        return dialog[include_caption]
    if include_caption == "caption,summary":
        return dialog["caption"] + dialog["summary"]      # (no separator: the reference concatenates the two strings as they are)
    return None


def get_vocabulary(dataset_file: str, cutoff: int = 1, include_caption: str = "none") -> Dict[str, int]:
    """data_handler.py:45-76.  Word ids in order of first appearance (caption, then all questions, then all answers of each
    dialog).  The reference ignores its ``cutoff`` argument: it loops over cutoffs 1..5 and returns the LAST vocabulary, i.e.
    words seen more than 5 times — reproduced here (``cutoff`` is accepted for signature compatibility)."""
    import json
    freq: Dict[str, int] = {}
    for dialog in json.load(open(dataset_file, "r"))["dialogs"]:
        cap = _caption_text(dialog, include_caption)
        if cap is not None:
            for w in cap.split():
                freq[w] = freq.get(w, 0) + 1
        for key in ("question", "answer"):
            for turn in dialog["dialog"]:
                for w in turn[key].split():
                    freq[w] = freq.get(w, 0) + 1
    vocab = dict(SPECIALS)
    for w, n in freq.items():
        if n > 5:
            vocab[w] = len(vocab)
    return vocab


def words2ids(text: str, vocab: Dict[str, int]) -> np.ndarray:
    """<sos> w1 .. wn <eos>, unknown words -> <unk> (data_handler.py:78-88)."""
    unk = vocab["<unk>"]
    return np.array([vocab["<sos>"]] + [vocab.get(w, unk) for w in text.split()] + [vocab["<eos>"]], dtype=np.int32)


def load(fea_types, fea_path: str, dataset_file: str, vocab: Dict[str, int], include_caption: str = "none",
         separate_caption: bool = False, max_history_length: int = -1, merge_source: bool = False, undisclosed_only: bool = False,
         read_features: bool = True) -> dict:
    """data_handler.py:90-148: one item [vid, qa_id, history, question, answer_in, answer_out(, caption)] per dialog turn; the
    history of turn n is the caption (or a single <blank> when the caption is kept apart) followed by the question+answer
    pairs of turns max(0, n - max_history_length) .. n-1.  ``features``: one dict per feature type, vid -> float32
    [frames, F] array read from fea_path with <FeaType> / <ImageID> substituted (.npy)."""
    import json
    raw = json.load(open(dataset_file, "r"))
    blank = np.array([vocab["<blank>"]], dtype=np.int32)
    has_cap = include_caption in ("caption", "summary", "caption,summary")
    items, vids, qa_id = [], [], 0
    for dialog in raw["dialogs"]:
        cap_text = _caption_text(dialog, include_caption)
        caption = words2ids(cap_text, vocab) if cap_text is not None else blank
        questions = [words2ids(t["question"], vocab) for t in dialog["dialog"]]
        answers = [words2ids(t["answer"], vocab) for t in dialog["dialog"]]
        pairs = [np.concatenate((q, a)).astype(np.int32) for q, a in zip(questions, answers)]
        vid = dialog["image_id"]
        if vid not in vids:
            vids.append(vid)
        turns = range(len(questions) - 1, len(questions)) if undisclosed_only else range(len(questions))
        for n in turns:
            if undisclosed_only:
                assert dialog["dialog"][n]["answer"] == "__UNDISCLOSED__"
            first = blank if (has_cap and separate_caption) else caption
            start = max(0, n - max_history_length) if max_history_length > 0 else 0
            history = np.concatenate([first] + pairs[start:n]) if n > start else first
            question = questions[n]
            if merge_source:
                question = np.concatenate((caption, history, question))
            item = [vid, qa_id, history, question, answers[n][:-1], answers[n][1:]]
            if has_cap and separate_caption:
                item.append(caption)
            items.append(item)
            qa_id += 1
    data = {"dialogs": items, "vocab": vocab, "features": [], "original": raw}
    if fea_types is not None and fea_types[0] != "none":
        for ftype in fea_types:
            base = fea_path.replace("<FeaType>", ftype)
            feats = {}
            for vid in vids:
                path = base.replace("<ImageID>", vid)
                feats[vid] = np.load(path).astype(np.float32, copy=False) if read_features else (path, int(np.load(path, mmap_mode="r").shape[0]))
            data["features"].append(feats)
    else:
        data["features"] = None
    return data


def feature_shape(data: dict) -> List[int]:
    """Feature width of every feature type (data_handler.py:277-285)."""
    out = []
    for feat in data["features"] or []:
        v = next(iter(feat.values()))
        out.append(int(np.load(v[0], mmap_mode="r").shape[-1]) if isinstance(v, tuple) else int(v.shape[-1]))
    return out


_FIELDS = (("his", 2), ("query", 3), ("trg", 4), ("trg_y", 5), ("cap", 6))     # dialog item columns (data_handler.py:130-133)


def make_batch_indices(data: dict, batchsize: int = 100, max_length: int = 20, separate_caption: bool = False):
    """data_handler.py:150-205: [(vids, qa_ids, x_len, h_len, q_len, a_len[, c_len], n)] and the sample count.
    Samples are sorted by (history, [caption,] frames, question, answer) length, longest first; a batch takes
    batchsize // (h_len // max_length + 1) samples (at least one) and is padded to the longest of each field."""
    idxlist = []
    for dialog in data["dialogs"]:
        vid = dialog[0]
        if data["features"] is not None:
            x_len = []
            for feat in data["features"]:
                value = feat[vid]
                x_len.append(value[1] if isinstance(value, tuple) else len(value))
        else:
            x_len = [0]
        item = (vid, dialog[1], x_len, len(dialog[2]), len(dialog[3]), len(dialog[4]))
        if separate_caption:
            item = item + (len(dialog[6]),)
        idxlist.append(item)
    if batchsize > 1:
        if separate_caption:
            idxlist = sorted(idxlist, key=lambda s: (-s[3], -s[6], -s[2][0], -s[4], -s[5]))
        else:
            idxlist = sorted(idxlist, key=lambda s: (-s[3], -s[2][0], -s[4], -s[5]))
    n_samples = len(idxlist)
    batch_indices = []
    bs = 0
    while bs < n_samples:
        in_len = idxlist[bs][3]
        bsize = int(batchsize / int(in_len / max_length + 1))
        be = min(bs + bsize, n_samples) if bsize > 0 else bs + 1
        chunk = idxlist[bs:be]
        x_len = [max(s[2][j] for s in chunk) for j in range(len(chunk[0][2]))]
        entry = ([s[0] for s in chunk], [s[1] for s in chunk], x_len, max(s[3] for s in chunk), max(s[4] for s in chunk),
                 max(s[5] for s in chunk))
        if separate_caption:
            entry = entry + (max(s[6] for s in chunk),)
        batch_indices.append(entry + (be - bs,))
        bs = be
    return batch_indices, n_samples


class DeviceCorpus:
    """The corpus resident in HBM.  Token fields: flat int64 + start/len per dialog (indexed by qa_id).  Features: flat
    [frames, F] float32 + start/len per video, in the order of ``data['features']``."""

    def __init__(self, data: dict, device="cuda"):
        self.device = torch.device(device)
        dialogs = data["dialogs"]
        self.n_dialogs = len(dialogs)
        self.has_caption = len(dialogs[0]) > 6
        by_id = sorted(dialogs, key=lambda d: d[1])
        assert [d[1] for d in by_id] == list(range(self.n_dialogs)), "qa ids must be 0..n-1 (data_handler.py:134)"
        self.tok: Dict[str, tuple] = {}
        for name, col in _FIELDS:
            if col == 6 and not self.has_caption:
                continue
            seqs = [np.asarray(d[col], dtype=np.int64).reshape(-1) for d in by_id]
            lens = np.array([len(s) for s in seqs], dtype=np.int32)
            start = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.int64)]).astype(np.int64)
            flat = np.concatenate(seqs) if len(seqs) else np.zeros(0, np.int64)
            self.tok[name] = tuple(torch.from_numpy(a).to(self.device) for a in (flat, start, lens))
        self.vid_index: Dict[object, int] = {}
        self.feat: List[tuple] = []
        if data.get("features"):
            vids = sorted(data["features"][0].keys(), key=str)
            self.vid_index = {v: i for i, v in enumerate(vids)}
            for feat in data["features"]:
                arrs = []
                for v in vids:
                    value = feat[v]
                    arrs.append(np.asarray(np.load(value[0]) if isinstance(value, tuple) else value, dtype=np.float32))
                lens = np.array([a.shape[0] for a in arrs], dtype=np.int32)
                start = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.int64)]).astype(np.int64)
                flat = np.concatenate(arrs, axis=0)
                self.feat.append(tuple(torch.from_numpy(a).to(self.device) for a in (flat, start, lens)) + (flat.shape[1],))

    def nbytes(self) -> int:
        n = sum(t.numel() * t.element_size() for f in self.tok.values() for t in f)
        return n + sum(t.numel() * t.element_size() for f in self.feat for t in f[:3])


def make_batch(corpus: DeviceCorpus, index, vocab, separate_caption: bool = False, skip: Sequence[int] = (1, 1, 1),
               out: Batch = None) -> Batch:
    """data_handler.py:219-274 on the device: ``index`` is one entry of make_batch_indices; ``vocab`` the vocabulary dict
    (its '<blank>' is the pad id) or the pad id itself.  Returns the same Batch the reference builds (fields, shapes,
    dtypes, mask semantics), with the masks' kernel images already attached.  ``out``: a Batch made by an earlier call with
    the same padded lengths and size — its tensors are refilled in place (what a captured hipGraph keeps reading)."""
    pad = int(vocab["<blank>"]) if isinstance(vocab, dict) else int(vocab)
    if separate_caption:
        x_len, h_len, q_len, a_len, c_len, n = index[2:]
    else:
        x_len, h_len, q_len, a_len, n = index[2:]
        c_len = None
    dev = corpus.device
    lib = L.load()
    ids = torch.tensor(list(index[1]), dtype=torch.int32).to(dev, non_blocking=True)
    plan = [("query", q_len), ("his", h_len), ("trg", a_len), ("trg_y", a_len)] + ([("cap", c_len)] if separate_caption else [])
    descs = (L.AssembleTokensDesc * len(plan))()
    reuse = out
    out: Dict[str, torch.Tensor] = {}
    masks: Dict[str, torch.Tensor] = {}
    if reuse is None:
        ntok = torch.zeros(1, dtype=torch.int64, device=dev)
        std = torch.empty(n, a_len, a_len, dtype=torch.uint8, device=dev)
    else:
        ntok, std = reuse._ntok, reuse.trg_mask._mtn_u8
        ntok.zero_()
        assert tuple(std.shape) == (n, a_len, a_len), "out= batch has other padded lengths"
    for k, (name, Lf) in enumerate(plan):
        flat, start, lens = corpus.tok[name]
        if reuse is None:
            out[name] = torch.empty(n, Lf, dtype=torch.int64, device=dev)
            masks[name] = torch.empty(n, Lf, dtype=torch.uint8, device=dev)
        else:
            out[name] = getattr(reuse, name)
            masks[name] = (reuse.trg_pad_u8 if name == "trg" else reuse.trg_y_pad_u8 if name == "trg_y"
                           else getattr(reuse, name + "_mask")._mtn_u8.view(n, Lf))
            assert tuple(out[name].shape) == (n, Lf), "out= batch has other padded lengths"
        D = descs[k]
        D.flat, D.start, D.len, D.ids, D.B, D.L, D.pad = flat.data_ptr(), start.data_ptr(), lens.data_ptr(), ids.data_ptr(), n, Lf, pad
        D.out, D.mask = out[name].data_ptr(), masks[name].data_ptr()
        if name == "trg":
            D.std_mask = std.data_ptr()
        if name == "trg_y":
            D.n_nonpad = ntok.data_ptr()
    L.check(lib.mtn_assemble_tokens(len(plan), descs, L.stream_ptr()))
    fts = fts_mask = None
    if corpus.feat:
        vids = torch.tensor([corpus.vid_index[v] for v in index[0]], dtype=torch.int32).to(dev, non_blocking=True)
        fd = (L.AssembleFeaturesDesc * len(corpus.feat))()
        fts, fmask = [], []
        for i, (flat, start, lens, F) in enumerate(corpus.feat):
            sk = int(skip[i]) if i < len(skip) else 1
            V = int(x_len[i])            # the reference pads to the UNskipped longest video (data_handler.py:236)
            if reuse is None:
                o = torch.empty(n, V, F, dtype=torch.float32, device=dev)
                mk = torch.empty(n, V, dtype=torch.uint8, device=dev)
            else:
                o, mk = reuse.fts[i], reuse.fts_mask[i]._mtn_u8.view(n, V)
                assert tuple(o.shape) == (n, V, F), "out= batch has other padded lengths"
            D = fd[i]
            D.flat, D.start, D.len, D.ids, D.B, D.V, D.F, D.skip = flat.data_ptr(), start.data_ptr(), lens.data_ptr(), vids.data_ptr(), n, V, F, sk
            D.out, D.mask = o.data_ptr(), mk.data_ptr()
            fts.append(o); fmask.append(mk)
        L.check(lib.mtn_assemble_features(len(corpus.feat), fd, L.stream_ptr()))
        fts_mask = [_as_bool(mk.unsqueeze(-2)) for mk in fmask]
    if reuse is not None:
        return reuse

    b = Batch.__new__(Batch)
    b._ntok, b.trg_pad_u8, b.trg_y_pad_u8 = ntok, masks["trg"], masks["trg_y"]
    b.query, b.his, b.his_st = out["query"], out["his"], None
    b.fts, b.fts_mask = fts, fts_mask
    b.query_mask, b.his_mask = _as_bool(masks["query"].unsqueeze(-2)), _as_bool(masks["his"].unsqueeze(-2))
    if separate_caption:
        b.cap, b.cap_mask = out["cap"], _as_bool(masks["cap"].unsqueeze(-2))
    else:
        b.cap = b.cap_mask = None
    b.trg, b.trg_y = out["trg"], out["trg_y"]
    b.trg_mask = _as_bool(std)
    b.ntokens = ntok[0]
    return b


def _as_bool(u8: torch.Tensor) -> torch.Tensor:
    """Reinterpret a 0/1 uint8 mask as bool (no copy) and keep the uint8 image the kernels read attached to it."""
    m = u8.view(torch.bool)
    m._mtn_u8 = u8
    return m
