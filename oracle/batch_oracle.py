"""CPU oracle (test infrastructure only) for batch assembly: numpy restatement of the reference's make_batch
(data_handler.py:219-274) + Batch mask logic (data_utils.py:23-54).  Pinned by tests/golden/batch_assembly.npz, which holds
the reference's own outputs (oracle/make_golden.py: run_batch_assembly).  Product code never imports this module."""
from typing import Dict, Sequence

import numpy as np


def pad_field(seqs: Sequence[np.ndarray], length: int, pad: int) -> np.ndarray:
    """data_handler.py:206-212 pad_seq: right-pad every sequence with ``pad`` to ``length``."""
    out = np.full((len(seqs), length), pad, dtype=np.int64)
    for i, s in enumerate(seqs):
        out[i, : len(s)] = s
    return out


def assemble(data: dict, index, pad: int, separate_caption: bool, skip: Sequence[int] = (1, 1, 1)) -> Dict[str, object]:
    """One batch as the reference builds it.  Returns query/his/trg/trg_y(/cap) int64 (B,L), *_mask bool (B,1,L), trg_mask
    bool (B,T,T), ntokens, fts list of float32 (B,V,F) with padded / all-ones frames zeroed, fts_mask list of bool (B,1,V)."""
    if separate_caption:
        x_len, h_len, q_len, a_len, c_len, n = index[2:]
    else:
        x_len, h_len, q_len, a_len, n = index[2:]
    dialogs = {d[1]: d for d in data["dialogs"]}
    rows = [dialogs[q] for q in index[1]]
    out: Dict[str, object] = {}
    out["his"] = pad_field([r[2] for r in rows], h_len, pad)
    out["query"] = pad_field([r[3] for r in rows], q_len, pad)
    out["trg"] = pad_field([r[4] for r in rows], a_len, pad)
    out["trg_y"] = pad_field([r[5] for r in rows], a_len, pad)
    if separate_caption:
        out["cap"] = pad_field([r[6] for r in rows], c_len, pad)
        out["cap_mask"] = (out["cap"] != pad)[:, None, :]                      # data_utils.py:36-37
    out["query_mask"] = (out["query"] != pad)[:, None, :]                      # data_utils.py:33
    out["his_mask"] = (out["his"] != pad)[:, None, :]                          # data_utils.py:34
    T = a_len
    causal = np.tril(np.ones((1, T, T), dtype=bool))                           # data_utils.py:10-14
    out["trg_mask"] = (out["trg"] != pad)[:, None, :] & causal                 # data_utils.py:48-54
    out["ntokens"] = int((out["trg_y"] != pad).sum())                          # data_utils.py:45
    fts, masks, raw = [], [], []
    for i, feat in enumerate(data["features"] or []):
        F = next(iter(feat.values())).shape[-1]
        x = np.ones((n, x_len[i], F), dtype=np.float32)                        # padded with ones (data_handler.py:236)
        for j, vid in enumerate(index[0]):
            fr = np.asarray(feat[vid], dtype=np.float32)[:: skip[i] if i < len(skip) else 1]   # data_handler.py:233
            x[j, : len(fr)] = fr
        raw.append(x.copy())
        m = (x != 1).sum(axis=2) != 0                                          # data_utils.py:29
        fts.append(x * m[:, :, None].astype(np.float32))                       # data_utils.py:30
        masks.append(m[:, None, :])
    out["fts"], out["fts_mask"], out["fts_padded_with_ones"] = fts, masks, raw
    return out
