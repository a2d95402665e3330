"""The dataset front end (mtn_amd.data_handler.get_vocabulary / words2ids / load / feature_shape; reference
data_handler.py:45-148, 277-285) against the REFERENCE's own outputs on a deterministic mini annotation file
(tests/golden/mini_avsd.json -> tests/golden/dataset_frontend.npz, both written by oracle/make_golden.py --dataset-only), and
the run.sh command line against mtn_amd.train's argument parser."""
import os

import numpy as np

from oracle.fixtures import det_avsd_json

GOLD = os.path.join(os.path.dirname(__file__), "golden")
VARIANTS = [dict(include_caption="none", separate_caption=False, max_history_length=-1, merge_source=False),
            dict(include_caption="caption", separate_caption=True, max_history_length=-1, merge_source=False),
            dict(include_caption="caption,summary", separate_caption=True, max_history_length=2, merge_source=False),
            dict(include_caption="summary", separate_caption=False, max_history_length=1, merge_source=True)]


def _features(tmp_path, raw):
    rs = np.random.RandomState(3)                     # same stream as oracle/make_golden.py:run_dataset_frontend
    dims = {"i3d": 12, "vgg": 5}
    for ft, F in dims.items():
        os.makedirs(tmp_path / ft)
        for d in raw["dialogs"]:
            np.save(tmp_path / ft / (d["image_id"] + ".npy"), rs.randn(rs.randint(3, 9), F).astype(np.float32))
    return dims, str(tmp_path / "<FeaType>" / "<ImageID>.npy")


def test_vocabulary_and_load_match_the_reference(tmp_path):
    import json
    from mtn_amd import data_handler as dh
    g = np.load(os.path.join(GOLD, "dataset_frontend.npz"))
    jpath = os.path.join(GOLD, "mini_avsd.json")
    raw = json.load(open(jpath))
    assert raw == det_avsd_json()                     # the committed input is the deterministic fixture
    dims, fea_path = _features(tmp_path, raw)
    for vi, kw in enumerate(VARIANTS):
        vocab = dh.get_vocabulary(jpath, include_caption=kw["include_caption"])
        assert sorted(vocab, key=vocab.get) == [str(w) for w in g[f"v{vi}.vocab"]]
        data = dh.load(list(dims), fea_path, jpath, vocab, **kw)
        items = data["dialogs"]
        assert len(items) == int(g[f"v{vi}.n_items"])
        assert [it[0] for it in items] == [str(v) for v in g[f"v{vi}.vids"]]
        assert [it[1] for it in items] == list(g[f"v{vi}.qa_ids"])
        for col, name in ((2, "his"), (3, "query"), (4, "ans_in"), (5, "ans_out"), (6, "cap")):
            if f"v{vi}.{name}.flat" not in g:
                assert col >= len(items[0])
                continue
            assert [len(it[col]) for it in items] == list(g[f"v{vi}.{name}.len"]), (vi, name)
            assert np.array_equal(np.concatenate([np.asarray(it[col]).ravel() for it in items]), g[f"v{vi}.{name}.flat"]), (vi, name)
        frames = [[len(data["features"][f][v]) for v in sorted(data["features"][f])] for f in range(len(dims))]
        assert np.array_equal(np.array(frames), g[f"v{vi}.frames"])
        assert dh.feature_shape(data) == list(g[f"v{vi}.feature_dims"])
        # the arrays are what make_batch_indices plans on (same lengths as the reference's (path, frames) pairs)
        idx, n = dh.make_batch_indices(data, batchsize=4, max_length=8, separate_caption=kw["separate_caption"] and kw["include_caption"] != "none")
        assert n == len(items) and sum(len(ix[1]) for ix in idx) == n


def test_run_sh_command_line_parses():
    """run.sh:109-140 with its default variable values (run.sh:13-52): every flag must be accepted."""
    from mtn_amd.train import parse
    argv = ["--gpu", "0", "--fea-type", "vggish", "i3d_flow", "--train-path", "data/<FeaType>/<ImageID>.npy",
            "--train-set", "data/train_set4DSTC7-AVSD.json", "--valid-path", "data/<FeaType>/<ImageID>.npy",
            "--valid-set", "data/valid_set4DSTC7-AVSD.json", "--num-epochs", "20", "--batch-size", "32", "--max-length", "256",
            "--model", "exp/mtn/avsd_model", "--rand-seed", "1", "--report-interval", "100", "--nb-blocks", "6",
            "--include-caption", "caption,summary", "--max-history-length", "3", "--separate-his-embed", "0",
            "--separate-caption", "1", "--merge-source", "0", "--separate-cap-embed", "0", "--warmup-steps", "9660",
            "--nb-blocks", "6", "--d-model", "512", "--d-ff", "2048", "--att-h", "8", "--dropout", "0.2", "--cut-a", "1",
            "--loss-l", "1", "--diff-encoder", "1", "--diff-embed", "0", "--auto-encoder-ft", "query", "--diff-gen", "0"]
    a = parse(argv)
    assert a.fea_type == ["vggish", "i3d_flow"] and a.train_set.endswith("train_set4DSTC7-AVSD.json")
    assert a.max_length == 256 and a.include_caption == "caption,summary" and a.max_history_length == 3 and a.dropout == 0.2


import pytest  # noqa: E402


@pytest.mark.gpu
def test_run_sh_style_training_on_the_mini_dataset(tmp_path):
    """`python train.py` with run.sh's data flags on the mini annotation file + .npy features: vocabulary, load, device corpus,
    two epochs with validation; writes <model>.conf (vocab, args), <model>_params.txt and state_dict checkpoints."""
    import json
    import pickle
    import torch
    from mtn_amd import train
    raw = json.load(open(os.path.join(GOLD, "mini_avsd.json")))
    dims, fea_path = _features(tmp_path, raw)
    model_prefix = str(tmp_path / "exp" / "mini")
    argv = ["--fea-type", "i3d", "vgg", "--train-path", fea_path, "--train-set", os.path.join(GOLD, "mini_avsd.json"),
            "--valid-path", fea_path, "--valid-set", os.path.join(GOLD, "mini_avsd.json"), "--num-epochs", "2", "--batch-size", "4",
            "--max-length", "256", "--model", model_prefix, "--include-caption", "caption,summary", "--separate-caption", "1",
            "--max-history-length", "3", "--nb-blocks", "1", "--d-model", "64", "--d-ff", "128", "--att-h", "4", "--dropout", "0.1",
            "--warmup-steps", "20", "--report-interval", "1000"]
    means = train.main(argv)
    assert len(means) == 2 and all(m == m and m > 0 for m in means) and means[1] < means[0] * 1.05
    vocab, args = pickle.load(open(model_prefix + ".conf", "rb"))
    assert vocab["<blank>"] == 1 and len(vocab) == args.vocab_size
    sd = torch.load(model_prefix + "_2.pth.tar")
    assert sd["generator.proj.weight"].shape[0] == len(vocab) and os.path.exists(model_prefix + "_best.pth.tar")
    assert os.path.exists(model_prefix + "_params.txt")
