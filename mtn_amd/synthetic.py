"""Synthetic batches of the reference's shapes: token ids uniform in [4,|V|) (specials 0-3 avoided,
data_handler.py:46), <blank>=1 padding, I3D / VGGish-like features ~ N(0,1) with padded frames = 1.0."""
from __future__ import annotations

from typing import Sequence

import torch

from .data_utils import Batch

PAD, UNK, SOS, EOS = 1, 0, 2, 3

# named workloads of BASELINE.json `configs` (B is per GPU)
CONFIGS = {
    "cfg1": dict(vocab=100, N=2, d_model=128, d_ff=512, h=4, ft_sizes=[2048, 128], B=4, Q=20, H=20, C=20, T=20, frames=[32, 32]),
    "cfg2": dict(vocab=3000, N=6, d_model=512, d_ff=2048, h=8, ft_sizes=[2048, 128], B=32, Q=20, H=128, C=40, T=20, frames=[32, 32]),
    "cfg3": dict(vocab=3000, N=6, d_model=512, d_ff=2048, h=8, ft_sizes=[2048, 128], B=64, Q=20, H=128, C=40, T=20, frames=[32, 32]),
    "cfg4": dict(vocab=3000, N=6, d_model=512, d_ff=2048, h=8, ft_sizes=[2048, 128], B=8, Q=20, H=512, C=40, T=20, frames=[256, 256]),
    # not a BASELINE config: the commonest padded shape of the ragged-corpus loop at AVSD lengths (tools/corpus_loop_probe.py: answers up to 56
    # tokens, questions 48, 40 frames) — what a real epoch's step looks like next to cfg2's 20-token targets; used for profiling only
    "avsd32": dict(vocab=3000, N=6, d_model=512, d_ff=2048, h=8, ft_sizes=[2048, 128], B=32, Q=48, H=192, C=40, T=56, frames=[40, 40]),
}


def synthetic_batch(vocab: int, B: int, Q: int, H: int, C: int, T: int, frames: Sequence[int], ft_sizes: Sequence[int],
                    device="cuda", seed: int = 1, ragged: bool = False) -> Batch:
    g = torch.Generator().manual_seed(seed)

    def toks(L, min_len=2):
        x = torch.randint(4, vocab, (B, L), generator=g, dtype=torch.int64)
        if ragged:
            lens = torch.randint(min_len, L + 1, (B,), generator=g)
            lens[0] = L
            x = torch.where(torch.arange(L).unsqueeze(0) < lens.unsqueeze(1), x, torch.full_like(x, PAD))
        return x

    query, his, cap = toks(Q), toks(H, 1), toks(C, 3)
    ans = toks(T + 1, 3)
    ans[:, 0] = SOS
    fts = []
    for V, F in zip(frames, ft_sizes):
        f = torch.randn(B, V, F, generator=g)
        if ragged:
            lens = torch.randint(max(1, V // 2), V + 1, (B,), generator=g)
            lens[0] = V
            f = torch.where((torch.arange(V).unsqueeze(0) < lens.unsqueeze(1)).unsqueeze(-1), f, torch.ones_like(f))
        fts.append(f.to(device))
    return Batch(query.to(device), his.to(device), None, fts, cap.to(device), ans[:, :-1].contiguous().to(device),
                 ans[:, 1:].contiguous().to(device), pad=PAD)


def flops_per_sample(N, d_model, d_ff, ft_sizes, Q, H, C, T, frames, vocab, **_):
    """Algorithmic train-step FLOPs per sample (2*MAC per GEMM/bmm; SURVEY.md §8d closed form, bwd = 2x fwd)."""
    d, ff = d_model, d_ff
    mha = lambda a, m: 4 * a * d * d + 4 * m * d * d + 4 * a * m * d
    ffn = lambda L: 4 * L * d * ff
    layer = mha(T, T) + mha(T, H) + mha(T, C) + mha(T, Q) + ffn(T)
    for V in frames:
        layer += mha(Q, Q) + mha(Q, V) + ffn(Q) + mha(T, Q)
    fwd = N * layer + sum(2 * V * f * d for V, f in zip(frames, ft_sizes))
    gen = 2 * d * vocab * (T + len(frames) * Q)
    return 3 * (fwd + gen)
