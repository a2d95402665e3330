#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE (/root/reference/mtn.py) on CPU.

Runs only in the build container (the reference never travels to the GPU box).  The
reference is imported, never copied: two shims are needed —
  * ``torchtext`` is absent and only used by dead code  -> stub modules;
  * ``Batch.__init__`` hard-codes ``.cuda()``             -> build the batch as a namespace
    with the same fields using the reference's own static ``Batch.make_std_mask`` and the
    feature-mask expression of data_utils.py:29.
Weights and inputs come from oracle/fixtures.py (deterministic formulas); only reference
OUTPUTS are stored.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py
"""
import os
import sys
import types
import warnings

sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle.fixtures import GOLDEN_CONFIGS, PAD, UNK, SOS, EOS, det_batch, det_state_dict  # noqa: E402

REF = "/root/reference"


def import_reference():
    tt = types.ModuleType("torchtext")
    tt_data = types.ModuleType("torchtext.data")
    tt_ds = types.ModuleType("torchtext.datasets")

    class Iterator:  # placeholder base class for the unused MyIterator
        pass

    tt_data.Iterator = Iterator
    tt_data.batch = lambda *a, **k: None
    tt.data, tt.datasets = tt_data, tt_ds
    sys.modules.update({"torchtext": tt, "torchtext.data": tt_data, "torchtext.datasets": tt_ds})
    sys.path.insert(0, REF)
    import mtn as ref_mtn                # noqa
    import data_utils as ref_du          # noqa
    import label_smoothing as ref_ls     # noqa
    return ref_mtn, ref_du, ref_ls


def ref_batch(ref_du, raw):
    b = types.SimpleNamespace()
    t = lambda a: torch.from_numpy(a)
    b.query, b.his, b.cap, b.trg, b.trg_y = t(raw["query"]), t(raw["his"]), t(raw["cap"]), t(raw["trg"]), t(raw["trg_y"])
    fts = [t(f) for f in raw["fts"]]                     # already (B,V,F): the permute of data_utils.py:28 done
    b.fts_mask = [(torch.sum(f != 1, dim=2) != 0).unsqueeze(-2) for f in fts]
    b.fts = [f * b.fts_mask[i].squeeze().unsqueeze(-1).expand_as(f).float() for i, f in enumerate(fts)]
    b.query_mask = (b.query != PAD).unsqueeze(-2)
    b.his_mask = (b.his != PAD).unsqueeze(-2)
    b.cap_mask = (b.cap != PAD).unsqueeze(-2)
    b.trg_mask = ref_du.Batch.make_std_mask(b.trg, PAD)
    b.ntokens = (b.trg_y != PAD).data.sum()
    b.his_st = None
    return b


def build_ref_model(ref_mtn, c, seed=0):
    m = ref_mtn.make_model(c["vocab"], c["vocab"], N=c["N"], d_model=c["d_model"], d_ff=c["d_ff"], h=c["h"],
                           dropout=0.0, ft_sizes=c["ft_sizes"], diff_encoder=c["diff_encoder"],
                           diff_embed=c["diff_embed"], diff_gen=c["diff_gen"], auto_encoder_ft=c["auto_encoder_ft"])
    sd = m.state_dict()
    new = det_state_dict({k: tuple(v.shape) for k, v in sd.items()}, seed)
    for k in sd:
        if k not in new:
            new[k] = sd[k]
    m.load_state_dict(new)
    return m


def run_config(name, c, ref_mtn, ref_du, ref_ls):
    out = {}
    torch.manual_seed(0)
    m = build_ref_model(ref_mtn, c)
    m.eval()
    raw = det_batch(c["vocab"], c["B"], c["Q"], c["H"], c["C"], c["T"], c["frames"], c["ft_sizes"], seed=1)
    b = ref_batch(ref_du, raw)

    # -- encode outputs
    q, v, cp, hs, ae = m.encode(b.query, b.query_mask, b.his, b.his_mask, b.cap, b.cap_mask, b.fts, b.fts_mask)
    out["enc.q"], out["enc.cap"], out["enc.his"] = q, cp, hs
    for i, x in enumerate(v):
        out[f"enc.vid.{i}"] = x
    if ae is not None:
        for i, x in enumerate(ae):
            out[f"enc.ae.{i}"] = x

    # -- per-sublayer outputs of decoder layer 0 (forward hooks on SublayerConnection)
    hooks = []
    for k, sl in enumerate(m.decoder.layers[0].sublayer):
        hooks.append(sl.register_forward_hook(lambda mod, inp, o, k=k: out.__setitem__(f"layer0.sublayer.{k}", o.detach().clone())))
    y, ae_out = m.forward(b)
    for h in hooks:
        h.remove()
    out["out"] = y
    for i, x in enumerate(ae_out):
        out[f"ae_out.{i}"] = x
    out["logp"] = m.generator(y)

    # -- loss + grads (SimpleLossCompute without optimiser, then .backward() by hand)
    crit = ref_ls.LabelSmoothing(size=c["vocab"], padding_idx=PAD, smoothing=0.1)
    ae_y = b.cap if c["auto_encoder_ft"] in ("caption", "summary") else b.query
    ae_norm = (ae_y != PAD).data.sum()
    m.zero_grad()
    y, ae_out = m.forward(b)
    gen = m.generator
    loss = crit(gen(y).contiguous().view(-1, c["vocab"]), b.trg_y.contiguous().view(-1)) / b.ntokens.float()
    for i, a in enumerate(ae_out):
        g = m.auto_encoder_generator[i] if m.auto_encoder_generator is not None else gen
        loss = loss + 1.0 * crit(g(a).contiguous().view(-1, c["vocab"]), ae_y.contiguous().view(-1)) / ae_norm.float()
    out["loss"] = loss.detach()
    loss.backward()
    gnames, gnorms = [], []
    for k, p in m.named_parameters():
        if p.grad is None:
            continue
        gnames.append(k)
        gnorms.append(float(p.grad.double().norm()))
        if p.grad.numel() <= 4096:
            out["grad." + k] = p.grad.detach().clone()
        else:
            out["gradhead." + k] = p.grad.detach().reshape(-1)[:256].clone()
    out["grad_names"] = np.array(gnames)
    out["grad_norms"] = np.array(gnorms, dtype=np.float64)

    # -- two optimiser steps exactly as train.py:190 + data_utils.py:123-156.  Kept in eval(): make_model
    #    builds MultiHeadedAttention(h, d_model) WITHOUT forwarding its dropout argument (mtn.py:339 vs :234),
    #    so attention-probability dropout is p=0.1 in train() whatever --dropout says, and is not reproducible.
    m.eval()
    m.zero_grad()
    opt = ref_du.NoamOpt(c["d_model"], 1, 10, torch.optim.Adam(m.parameters(), lr=0, betas=(0.9, 0.98), eps=1e-9))
    lc = ref_du.SimpleLossCompute(m.generator, m.auto_encoder_generator, crit, opt=opt, l=1.0)
    step_losses = []
    for _ in range(2):
        y, ae_out = m.forward(b)
        step_losses.append(float(lc(y, b.trg_y, b.ntokens, ae_out, ae_y, ae_norm)))
    out["step_losses"] = np.array(step_losses, dtype=np.float64)
    sd = m.state_dict()
    for k in ["decoder.layers.0.self_attn.linears.0.bias", "decoder.layers.0.sublayer.0.norm.a_2",
              "decoder.norm.b_2", "generator.proj.bias", "decoder.layers.0.feed_forward.w_2.bias"]:
        out["after2." + k] = sd[k].detach().clone()
    out["after2head.decoder.layers.0.his_attn.linears.1.weight"] = sd["decoder.layers.0.his_attn.linears.1.weight"].reshape(-1)[:256].clone()
    out["after2head.query_embed.0.lut.weight"] = sd["query_embed.0.lut.weight"].reshape(-1)[:256].clone()

    # -- beam search (data_utils.py:188) on the first dialogue, reference defaults beam=5,penalty=1,nbest=5
    m = build_ref_model(ref_mtn, c)
    m.eval()
    raw1 = det_batch(c["vocab"], 1, c["Q"], c["H"], c["C"], c["T"], c["frames"], c["ft_sizes"], seed=2, ragged=False)
    b1 = ref_batch(ref_du, raw1)
    with torch.no_grad():
        nbest, best = ref_du.beam_search_decode(m, b1, 8, SOS, UNK, EOS, PAD)
    out["beam.n"] = np.array(len(nbest))
    for i, (toks, score) in enumerate(nbest):
        out[f"beam.tokens.{i}"] = np.array([int(t) for t in toks], dtype=np.int64)
        out[f"beam.score.{i}"] = np.array(float(score), dtype=np.float64)
    out["beam.best"] = np.array(float(best), dtype=np.float64)

    npz = {}
    for k, v in out.items():
        npz[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **npz)
    print(f"{name}: {len(npz)} arrays, loss={float(out['loss']):.6f}, {os.path.getsize(path) / 1e6:.2f} MB")


def run_batch_assembly():
    """Batch planning + assembly of the reference (data_handler.py:150-274, data_utils.py:23-54) on the deterministic
    mini-corpus of fixtures.det_corpus -> tests/golden/batch_assembly.npz.  The reference hard-codes .cuda() in
    prepare_data / Batch (SURVEY §8c shim 2): for this run Tensor.cuda is an identity, the feature arrays are written to
    temporary .npy files because make_batch np.loads them."""
    import tempfile
    from oracle.fixtures import det_corpus
    import data_handler as ref_dh        # noqa
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    out = {}
    try:
        for cap in (True, False):
            data = det_corpus(caption=cap)
            tmp = tempfile.mkdtemp(prefix="mtn_golden_")
            feats = []
            for fi, d in enumerate(data["features"]):
                fd = {}
                for vid, arr in d.items():
                    path = os.path.join(tmp, f"f{fi}_{vid}.npy")
                    np.save(path, arr)
                    fd[vid] = (path, arr.shape[0])
                feats.append(fd)
            ref_data = {"dialogs": data["dialogs"], "features": feats, "vocab": data["vocab"]}
            for bsz, mlen in ((4, 8), (1, 20), (6, 20)):
                tag = f"cap{int(cap)}.b{bsz}"
                idx, n = ref_dh.make_batch_indices(ref_data, batchsize=bsz, max_length=mlen, separate_caption=cap)
                out[f"{tag}.n_samples"] = np.array(n)
                out[f"{tag}.n_batches"] = np.array(len(idx))
                for k, ix in enumerate(idx):
                    out[f"{tag}.{k}.qa_ids"] = np.array(ix[1], dtype=np.int64)
                    out[f"{tag}.{k}.vids"] = np.array([int(v[3:]) for v in ix[0]], dtype=np.int64)
                    out[f"{tag}.{k}.lens"] = np.array(list(ix[2]) + list(ix[3:]), dtype=np.int64)
                    if bsz == 1 and k >= 4:
                        continue
                    b = ref_dh.make_batch(ref_data, ix, data["vocab"], separate_caption=cap, skip=[1, 2] if bsz == 6 else [1, 1])
                    for name in ("query", "his", "trg", "trg_y", "query_mask", "his_mask", "trg_mask"):
                        out[f"{tag}.{k}.{name}"] = getattr(b, name).numpy()
                    if cap:
                        out[f"{tag}.{k}.cap"], out[f"{tag}.{k}.cap_mask"] = b.cap.numpy(), b.cap_mask.numpy()
                    out[f"{tag}.{k}.ntokens"] = np.array(int(b.ntokens))
                    for i, (f, m) in enumerate(zip(b.fts, b.fts_mask)):
                        out[f"{tag}.{k}.fts.{i}"], out[f"{tag}.{k}.fts_mask.{i}"] = f.numpy(), m.numpy()
    finally:
        torch.Tensor.cuda = orig_cuda
    path = os.path.join(ROOT, "tests", "golden", "batch_assembly.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays", os.path.getsize(path), "bytes")


DATASET_VARIANTS = [dict(include_caption="none", separate_caption=False, max_history_length=-1, merge_source=False),
                    dict(include_caption="caption", separate_caption=True, max_history_length=-1, merge_source=False),
                    dict(include_caption="caption,summary", separate_caption=True, max_history_length=2, merge_source=False),
                    dict(include_caption="summary", separate_caption=False, max_history_length=1, merge_source=True)]


def run_dataset_frontend():
    """The reference's dataset front end (data_handler.py:45-148: get_vocabulary, words2ids, load, feature_shape) on the
    deterministic mini annotation file of fixtures.det_avsd_json -> tests/golden/mini_avsd.json (the INPUT, our own synthetic
    text) and tests/golden/dataset_frontend.npz (the reference's OUTPUTS: vocabulary in id order, every item's arrays)."""
    import json
    import tempfile
    from oracle.fixtures import det_avsd_json
    import data_handler as ref_dh        # noqa
    raw = det_avsd_json()
    jpath = os.path.join(ROOT, "tests", "golden", "mini_avsd.json")
    json.dump(raw, open(jpath, "w"), indent=0)
    tmp = tempfile.mkdtemp(prefix="mtn_golden_ds_")
    rs = np.random.RandomState(3)
    dims = {"i3d": 12, "vgg": 5}
    for ft, F in dims.items():
        os.makedirs(os.path.join(tmp, ft))
        for d in raw["dialogs"]:
            np.save(os.path.join(tmp, ft, d["image_id"] + ".npy"), rs.randn(rs.randint(3, 9), F).astype(np.float32))
    fea_path = os.path.join(tmp, "<FeaType>", "<ImageID>.npy")
    out = {}
    for vi, kw in enumerate(DATASET_VARIANTS):
        vocab = ref_dh.get_vocabulary(jpath, include_caption=kw["include_caption"])
        words = sorted(vocab, key=vocab.get)
        assert [vocab[w] for w in words] == list(range(len(words)))
        out[f"v{vi}.vocab"] = np.array(words)
        data = ref_dh.load(list(dims), fea_path, jpath, vocab, **kw)
        out[f"v{vi}.n_items"] = np.array(len(data["dialogs"]))
        out[f"v{vi}.vids"] = np.array([it[0] for it in data["dialogs"]])
        out[f"v{vi}.qa_ids"] = np.array([it[1] for it in data["dialogs"]], dtype=np.int64)
        for col, name in ((2, "his"), (3, "query"), (4, "ans_in"), (5, "ans_out"), (6, "cap")):
            if col < len(data["dialogs"][0]):
                out[f"v{vi}.{name}.flat"] = np.concatenate([np.asarray(it[col]).ravel() for it in data["dialogs"]]).astype(np.int64)
                out[f"v{vi}.{name}.len"] = np.array([len(it[col]) for it in data["dialogs"]], dtype=np.int64)
        out[f"v{vi}.frames"] = np.array([[data["features"][f][v][1] for v in sorted(data["features"][f])] for f in range(len(dims))], dtype=np.int64)
        out[f"v{vi}.feature_dims"] = np.array(ref_dh.feature_shape(data), dtype=np.int64)
    path = os.path.join(ROOT, "tests", "golden", "dataset_frontend.npz")
    np.savez_compressed(path, **out)
    print("wrote", jpath, "and", path, len(out), "arrays", os.path.getsize(path), "bytes")


def main():
    ref_mtn, ref_du, ref_ls = import_reference()
    torch.set_num_threads(8)
    if "--batch-assembly-only" in sys.argv:
        run_batch_assembly()
        return
    if "--dataset-only" in sys.argv:
        run_dataset_frontend()
        return
    if "--only" in sys.argv:
        name = sys.argv[sys.argv.index("--only") + 1]
        run_config(name, GOLDEN_CONFIGS[name], ref_mtn, ref_du, ref_ls)
        return
    run_batch_assembly()
    run_dataset_frontend()
    for name, c in GOLDEN_CONFIGS.items():
        run_config(name, c, ref_mtn, ref_du, ref_ls)


if __name__ == "__main__":
    main()
