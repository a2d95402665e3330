"""LayerNorm backward by linearity (round 4; include/mtn_hip.h mtn_ln_epilogue, csrc/gemm.hip ln_consume_epilogue):
the kernels that produce dq (fused head backward) / dh (the FFN's dh GEMM) emit per-row partial dot products against the fold
vectors u = W a2, c = b + W b2, and the dLN-out GEMM applies LayerNorm backward in its epilogue — instead of a LayerNorm-backward
launch per group.  Reference math: the autograd of mtn.py:111-114 under mtn.py:127.

Checked here: the fold vectors themselves; the epilogue path against the separate-launch path on IDENTICAL forward state
(same saved buffers, same dropout masks: only the two row sums are computed differently); which path ran (library counter).
The oracle-level parity of the path is the job of tests/test_fused_gpu.py and tests/test_full_size_gpu.py, which run with the
epilogue on (the default)."""
import os

import pytest
import torch

from tests.test_fused_gpu import CFGS
from tests.test_model_gpu import build_model, dev, dev_batch, raw_batch  # noqa: F401  (dev is a fixture)
from tests.util import relmax

pytestmark = pytest.mark.gpu


def _set_epi(on):
    from mtn_amd import lib
    if on:
        os.environ.pop("MTN_LN_EPI", None)
    else:
        os.environ["MTN_LN_EPI"] = "0"
    lib.reload_env()


def _grads(model, b):
    model.prepare()
    model.zero_glue_grads()
    out, ae = model.forward(b)
    loss = (out.float() ** 2).mean() + sum((a.float() ** 2).mean() for a in ae)
    loss.backward()
    torch.cuda.synchronize()
    return {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}


def _randomise_layer_norms(model, seed=3):
    """Gains and biases away from (1, 0): u = W a2 and c = b + W b2 must really be used."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith(".a_2"):
                p.copy_((1.0 + 0.3 * torch.randn(p.shape, generator=g)).to(p.device))
            elif n.endswith(".b_2"):
                p.copy_((0.2 * torch.randn(p.shape, generator=g)).to(p.device))


def test_fold_vectors_match_their_definition(dev):
    c = CFGS["query_b5"]
    model = build_model(c, torch.bfloat16, dev).train()
    _randomise_layer_norms(model)
    model.prepare()
    assert model._ln_fold_table is not None
    model.fold_layer_norms()
    torch.cuda.synchronize()
    seen = 0
    for layer in model.decoder.layers:
        nF = len(layer.auto_encoder_vid_attn)
        pairs = [(layer.sublayer[0], layer.self_attn, True), (layer.sublayer[1], layer.his_attn, False),
                 (layer.sublayer[4], layer.auto_encoder_self_attn[0], True), (layer.sublayer[6], layer.auto_encoder_feed_forward[0], None),
                 (layer.sublayer[4 + 4 * nF], layer.feed_forward, None)]
        for sc, mod, self_attn in pairs:
            assert sc._ln_fold is not None and sc._ln_fold[0] == id(mod)
            got = sc._ln_fold[1].double().cpu()
            f = mod._fused
            if self_attn is None:
                w, bias = f["w1_lp"], f["b1"]
            else:
                K = 3 * c["d_model"] if self_attn else c["d_model"]
                w, bias = f["w_qkv_lp"][:K], f["b_qkv"][:K]
            w = w.double().cpu()
            u = w @ sc.norm.a_2.double().cpu()
            cc = bias.double().cpu() + w @ sc.norm.b_2.double().cpu()
            K = w.size(0)
            assert got.numel() == 2 * K
            assert float((got[:K] - u).abs().max()) < 1e-4 * max(1.0, float(u.abs().max()))
            assert float((got[K:] - cc).abs().max()) < 1e-4 * max(1.0, float(cc.abs().max()))
            seen += 1
    assert seen == 5 * c["N"]


@pytest.mark.parametrize("name", ["query_b5", "query_b32", "caption_b3", "shared_b7"])
@pytest.mark.parametrize("dropout", [0.0, 0.1], ids=["nodrop", "drop"])
def test_epilogue_path_equals_layernorm_launch(dev, name, dropout):
    """Every parameter gradient with LayerNorm backward in the GEMM epilogue vs in its own launch.  The forward pass is the same
    code in both runs (bit-identical saved buffers and masks); the backward differs only in how the two row sums of LayerNorm
    backward are obtained — from the fp32 accumulators of g = dq W, or as dq . u and dq . (q - c) with q the SAVED bf16 projection
    output — so the two gradients agree far inside the bf16 bars."""
    from mtn_amd import lib
    c = CFGS[name]
    torch.manual_seed(0)
    model = build_model(c, torch.bfloat16, dev, dropout=dropout, attn_dropout=dropout).train()
    _randomise_layer_norms(model)
    b = dev_batch(raw_batch(c), dev)
    model.prepare()
    model._seed.fill_(4321)
    seed0 = model._seed.clone()
    try:
        _set_epi(False)
        n0 = lib.load().mtn_ln_epilogue_groups()
        gref = _grads(model, b)
        assert lib.load().mtn_ln_epilogue_groups() == n0, "MTN_LN_EPI=0 must keep the LayerNorm-backward launches"
        model._seed.copy_(seed0)
        _set_epi(True)
        ggot = _grads(model, b)
        ran = lib.load().mtn_ln_epilogue_groups() - n0
    finally:
        _set_epi(True)
    if c["auto_encoder_ft"] in ("query", "caption", "summary") and c["diff_encoder"]:
        assert ran > 0, "no backward group took the LayerNorm epilogue"
    worst = 0.0
    for k in gref:
        r, g = gref[k].float().flatten(), ggot[k].float().flatten()
        assert torch.isfinite(g).all(), k
        if float(r.abs().max()) == 0.0 or k.endswith("linears.1.bias"):
            continue
        cos = float(torch.dot(r, g) / (r.norm() * g.norm() + 1e-30))
        rel = float((g - r).norm() / r.norm())
        worst = max(worst, rel)
        assert cos > 0.99995, (k, cos)
        assert rel < 1e-2, (k, rel)
        assert relmax(g, r) < 2e-2, (k, relmax(g, r))
    print(f"{name} dropout={dropout}: groups on the epilogue {ran}, worst ||diff||/||ref|| {worst:.2e}")


def test_epilogue_kernels_on_eight_and_sixteen_waves_agree_bitwise(dev):
    """The LayerNorm-epilogue GEMMs run the 64 x 64 tile on sixteen waves by default (MTN_GEMM_NW16 bit 1); the row-sum gather keeps
    its eight threads per row and its order, so every gradient must have the same bits as with the eight-wave kernels."""
    from mtn_amd import lib
    c = CFGS["query_b32"]
    torch.manual_seed(0)
    model = build_model(c, torch.bfloat16, dev, dropout=0.1, attn_dropout=0.1).train()
    _randomise_layer_norms(model)
    b = dev_batch(raw_batch(c), dev)
    model.prepare()
    model._seed.fill_(777)
    seed0 = model._seed.clone()
    res = []
    try:
        for nw in ("3", "0"):
            os.environ["MTN_GEMM_NW16"] = nw
            lib.reload_env()
            model._seed.copy_(seed0)
            res.append(_grads(model, b))
    finally:
        os.environ.pop("MTN_GEMM_NW16", None)
        lib.reload_env()
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), k
