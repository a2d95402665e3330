"""Batch planning (host) and batch assembly (device) against the REFERENCE's own outputs: tests/golden/batch_assembly.npz
was produced by running data_handler.make_batch_indices / make_batch / Batch of the reference on fixtures.det_corpus
(oracle/make_golden.py: run_batch_assembly).  Integer / copy work: bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from oracle import batch_oracle

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "batch_assembly.npz"))
CASES = [(cap, bsz, mlen) for cap in (True, False) for bsz, mlen in ((4, 8), (1, 20), (6, 20))]


def _plan(cap, bsz, mlen):
    from mtn_amd.data_handler import make_batch_indices
    data = fx.det_corpus(caption=cap)
    idx, n = make_batch_indices(data, batchsize=bsz, max_length=mlen, separate_caption=cap)
    return data, idx, n


@pytest.mark.parametrize("cap,bsz,mlen", CASES)
def test_batch_plan_matches_reference(cap, bsz, mlen):
    data, idx, n = _plan(cap, bsz, mlen)
    tag = f"cap{int(cap)}.b{bsz}"
    assert n == int(GOLD[f"{tag}.n_samples"]) and len(idx) == int(GOLD[f"{tag}.n_batches"])
    for k, ix in enumerate(idx):
        assert list(ix[1]) == GOLD[f"{tag}.{k}.qa_ids"].tolist()
        assert [int(v[3:]) for v in ix[0]] == GOLD[f"{tag}.{k}.vids"].tolist()
        assert list(ix[2]) + list(ix[3:]) == GOLD[f"{tag}.{k}.lens"].tolist()


def _check(tag, k, got, cap, as_np):
    names = ["query", "his", "trg", "trg_y", "query_mask", "his_mask", "trg_mask"] + (["cap", "cap_mask"] if cap else [])
    for name in names:
        want = GOLD[f"{tag}.{k}.{name}"]
        have = as_np(got[name] if isinstance(got, dict) else getattr(got, name))
        assert have.shape == want.shape and have.dtype == want.dtype and np.array_equal(have, want), name
    nt = got["ntokens"] if isinstance(got, dict) else int(got.ntokens)
    assert nt == int(GOLD[f"{tag}.{k}.ntokens"])
    fts = got["fts"] if isinstance(got, dict) else got.fts
    msk = got["fts_mask"] if isinstance(got, dict) else got.fts_mask
    for i, (f, m) in enumerate(zip(fts, msk)):
        assert np.array_equal(as_np(f), GOLD[f"{tag}.{k}.fts.{i}"]) and np.array_equal(as_np(m), GOLD[f"{tag}.{k}.fts_mask.{i}"])


@pytest.mark.parametrize("cap,bsz,mlen", CASES)
def test_oracle_assembly_matches_reference(cap, bsz, mlen):
    data, idx, _ = _plan(cap, bsz, mlen)
    tag = f"cap{int(cap)}.b{bsz}"
    for k, ix in enumerate(idx):
        if f"{tag}.{k}.query" not in GOLD.files:
            continue
        got = batch_oracle.assemble(data, ix, fx.PAD, cap, skip=[1, 2] if bsz == 6 else [1, 1])
        _check(tag, k, got, cap, np.asarray)


@pytest.mark.gpu
@pytest.mark.parametrize("cap,bsz,mlen", CASES)
def test_device_assembly_matches_reference(cap, bsz, mlen):
    from mtn_amd.data_handler import DeviceCorpus, make_batch
    data, idx, _ = _plan(cap, bsz, mlen)
    corpus = DeviceCorpus(data, "cuda:0")
    tag = f"cap{int(cap)}.b{bsz}"
    for k, ix in enumerate(idx):
        if f"{tag}.{k}.query" not in GOLD.files:
            continue
        b = make_batch(corpus, ix, data["vocab"], separate_caption=cap, skip=[1, 2] if bsz == 6 else [1, 1])
        torch.cuda.synchronize()
        _check(tag, k, b, cap, lambda t: t.cpu().numpy())


@pytest.mark.gpu
def test_device_assembly_large_corpus_and_model_step():
    """A larger random corpus: every batch equals the oracle's; an assembled batch drives the model (masks' kernel images
    attached by the assembly are the ones the attention kernels read)."""
    from mtn_amd import make_model
    from mtn_amd.data_handler import DeviceCorpus, make_batch, make_batch_indices
    data = fx.det_corpus(n_videos=40, turns=6, vocab=200, ft_sizes=(64, 24), seed=5)
    idx, n = make_batch_indices(data, batchsize=16, max_length=20, separate_caption=True)
    assert n == 240 and sum(ix[-1] for ix in idx) == 240
    corpus = DeviceCorpus(data, "cuda:0")
    for ix in idx:
        b = make_batch(corpus, ix, fx.PAD, separate_caption=True)
        want = batch_oracle.assemble(data, ix, fx.PAD, True)
        torch.cuda.synchronize()
        for name in ("query", "his", "cap", "trg", "trg_y", "query_mask", "his_mask", "cap_mask", "trg_mask"):
            assert np.array_equal(getattr(b, name).cpu().numpy(), want[name]), name
        assert int(b.ntokens) == want["ntokens"]
        for f, m, wf, wm in zip(b.fts, b.fts_mask, want["fts"], want["fts_mask"]):
            assert np.array_equal(f.cpu().numpy(), wf) and np.array_equal(m.cpu().numpy(), wm)
    model = make_model(200, 200, N=1, d_model=64, d_ff=128, h=4, ft_sizes=[64, 24], diff_encoder=True, auto_encoder_ft="caption",
                       compute_dtype="bf16").to("cuda:0").eval()
    b = make_batch(corpus, idx[0], fx.PAD, separate_caption=True)
    from mtn_amd import Batch
    t = torch.from_numpy
    w = batch_oracle.assemble(data, idx[0], fx.PAD, True)
    hb = Batch(t(w["query"]), t(w["his"]), None, [t(f) for f in w["fts_padded_with_ones"]], t(w["cap"]), t(w["trg"]), t(w["trg_y"]), pad=fx.PAD, device="cuda:0")
    with torch.no_grad():
        o1 = model.forward(b)[0]
        o2 = model.forward(hb)[0]
    assert torch.equal(o1, o2)


@pytest.mark.gpu
def test_corpus_training_loop_learns():
    """The reference's epoch loop (train.py:23-50) end to end on a small ragged corpus: batch planning, device-side assembly,
    forward, fused loss head, backward, fused Noam/Adam — the mean loss per token falls over the epochs."""
    from mtn_amd import train
    means = train.main(["--corpus-videos", "6", "--nb-blocks", "1", "--d-model", "64", "--d-ff", "128", "--att-h", "4", "--vocab-size", "64",
                        "--ft-sizes", "32", "16", "--batch-size", "8", "--num-epochs", "4", "--warmup-steps", "30", "--report-interval", "1000",
                        "--dropout", "0.0"])
    assert len(means) == 4 and all(np.isfinite(means)) and means[-1] < 0.8 * means[0], means


@pytest.mark.gpu
def test_bucketed_graph_trainer_matches_eager_loop():
    """BucketedTrainer (lengths padded to the bucket, static batches refilled in place, one captured graph per padded shape,
    shared optimiser) follows the eager per-batch loop: same loss per step on the same batch sequence (dropout off), and the
    refilled static batch equals a freshly assembled one."""
    from mtn_amd import make_model, LabelSmoothing, NoamOpt, FusedAdam, SimpleLossCompute
    from mtn_amd.data_handler import DeviceCorpus, make_batch, make_batch_indices
    from mtn_amd.train_step import BucketedTrainer
    data = fx.det_corpus(n_videos=12, turns=4, vocab=64, ft_sizes=(32, 16), seed=3)
    idx, _ = make_batch_indices(data, batchsize=6, max_length=20, separate_caption=True)
    corpus = DeviceCorpus(data, "cuda:0")

    def model():
        torch.manual_seed(0)
        return make_model(64, 64, N=2, d_model=64, d_ff=128, h=4, dropout=0.0, ft_sizes=[32, 16], diff_encoder=True, auto_encoder_ft="query",
                          compute_dtype="bf16", attn_dropout=0.0).to("cuda:0").train()

    order = [0, 3, 1, 0, 2, 3, 1] if len(idx) > 3 else list(range(len(idx))) * 2
    m1 = model()
    tr = BucketedTrainer(m1, corpus, 64, pad=fx.PAD, warmup=50, bucket=4)
    graphed = []
    for k in order:
        loss, b = tr.step(idx[k])
        graphed.append(float(loss))
        fresh = make_batch(corpus, tr._padded(idx[k]), fx.PAD, separate_caption=True)
        for name in ("query", "his", "cap", "trg", "trg_y", "query_mask", "his_mask", "cap_mask", "trg_mask"):
            assert torch.equal(getattr(b, name), getattr(fresh, name)), name
        assert int(b.ntokens) == int(fresh.ntokens)
        for f1, f2, k1, k2 in zip(b.fts, fresh.fts, b.fts_mask, fresh.fts_mask):
            assert torch.equal(f1, f2) and torch.equal(k1, k2)
    assert len(tr.steps) <= len(set(order))
    m2 = model()
    lc = SimpleLossCompute(m2.generator, m2.auto_encoder_generator, LabelSmoothing(64, fx.PAD, 0.1),
                           opt=NoamOpt(64, 1, 50, FusedAdam(m2)), sync=False)
    eager = []
    for k in order:
        b = make_batch(corpus, idx[k], fx.PAD, separate_caption=True)        # un-padded lengths: padding must not matter
        out, ae_out = m2.forward(b)
        eager.append(float(lc(out, b.trg_y, b.ntokens, ae_out, b.query, (b.query != fx.PAD).sum())) / float(b.ntokens))
    g = [x for x in graphed]
    # SimpleLossCompute returns loss*norm with sync=False -> compare the per-token values
    assert max(abs(a - e * 1.0) / abs(e) for a, e in zip(g, eager)) < 2e-2, (g, eager)
