"""The fused first launch of a sublayer group (csrc/fused.hip: LayerNorm -> head slice of the projections -> attention, or
LayerNorm -> w_1 column slice + ReLU + dropout, one kernel) against (a) the four-launch path it replaces, on identical inputs,
weights and dropout streams, and (b) the CPU oracle, at the shapes the fused kernel is built for (d_model 512, 8 heads).
Reference ops: mtn.py:125-127, 248-267, 221-231, 279-280."""
import os

import pytest
import torch

from oracle import fixtures as fx
from tests.test_model_gpu import build_model, dev, dev_batch, raw_batch  # noqa: F401  (dev is a fixture)
from tests.util import relmax

pytestmark = pytest.mark.gpu

# d = 512 / h = 8 variants of the golden configurations' shapes (ragged lengths, an empty history row, padded frames)
CFGS = {
    "query_b5": dict(vocab=120, N=2, d_model=512, d_ff=2048, h=8, ft_sizes=[64, 32], B=5, Q=13, H=37, C=29, T=20, frames=[17, 9],
                     diff_encoder=True, diff_embed=False, diff_gen=False, auto_encoder_ft="query"),
    "query_b32": dict(vocab=120, N=1, d_model=512, d_ff=2048, h=8, ft_sizes=[64, 32], B=32, Q=20, H=128, C=40, T=20, frames=[32, 32],
                      diff_encoder=True, diff_embed=False, diff_gen=False, auto_encoder_ft="query"),
    # batch 64 (BASELINE configs[2] per-GPU batch): the attention groups need two rounds of the chip -> two unit sizes, parts of a
    # sublayer as members of their own (csrc/fused.hip fh_plan): dropout indices and every per-row pointer of the second part
    "query_b64": dict(vocab=120, N=1, d_model=512, d_ff=2048, h=8, ft_sizes=[64, 32], B=64, Q=20, H=128, C=40, T=20, frames=[32, 32],
                      diff_encoder=True, diff_embed=False, diff_gen=False, auto_encoder_ft="query"),
    "caption_b3": dict(vocab=90, N=1, d_model=512, d_ff=1024, h=8, ft_sizes=[48], B=3, Q=9, H=70, C=44, T=33, frames=[21],
                       diff_encoder=True, diff_embed=False, diff_gen=False, auto_encoder_ft="caption"),
    "shared_b7": dict(vocab=60, N=1, d_model=512, d_ff=2048, h=8, ft_sizes=[40, 24], B=7, Q=8, H=5, C=16, T=7, frames=[6, 11],
                      diff_encoder=False, diff_embed=False, diff_gen=False, auto_encoder_ft="query"),
}


def _run(model, b, fused, train):
    from mtn_amd import lib
    prev = lib.load().mtn_fused_enable(1 if fused else 0)
    try:
        model.prepare()
        model.zero_glue_grads()       # gradients live in the flat buffer: path weights are written, glue gradients accumulate
        out, ae = model.forward(b)
        res = [out.detach().clone()] + [a.detach().clone() for a in ae]
        grads = None
        if train:
            loss = (out.float() ** 2).mean() + sum((a.float() ** 2).mean() for a in ae)
            loss.backward()
            torch.cuda.synchronize()
            grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        torch.cuda.synchronize()
        return res, grads
    finally:
        lib.load().mtn_fused_enable(1 if prev != 0 else 0)


@pytest.mark.parametrize("name", list(CFGS))
@pytest.mark.parametrize("dropout", [0.0, 0.1], ids=["nodrop", "drop"])
def test_fused_stage_equals_four_launch_path(dev, name, dropout):
    """Same inputs, same weights, same dropout seed (hence the same masks): the fused kernel reproduces the four-launch path's
    outputs.  Same arithmetic, different fp32 summation orders (LayerNorm sums inside 16-lane rows, the contraction in two
    halves), so single bf16 roundings of xn / q / k / v / hidden flip here and there and the difference grows through the
    layers: measured 2-3e-3 relative to max on the decoder outputs; the bar is half the bf16 tolerance."""
    c = CFGS[name]
    model = build_model(c, torch.bfloat16, dev, dropout=dropout, attn_dropout=dropout)
    model.train() if dropout > 0 else model.eval()
    b = dev_batch(raw_batch(c), dev)
    model.prepare()
    seed0 = model._seed.clone()
    ref, _ = _run(model, b, fused=False, train=False)
    model._seed.copy_(seed0)
    got, _ = _run(model, b, fused=True, train=False)
    for r, g in zip(ref, got):
        assert torch.isfinite(g).all()
        assert relmax(g, r) < 5e-3, (name, relmax(g, r))


@pytest.mark.parametrize("name", ["query_b5", "caption_b3", "shared_b7"])
def test_fused_model_matches_oracle(dev, name):
    """d = 512 / 8 heads model with the fused launches on, eval mode, against the CPU oracle: bf16 tolerance 1e-2 (relative to
    max) on the decoder output and the auto-encoder outputs."""
    c = CFGS[name]
    model = build_model(c, torch.bfloat16, dev).eval()
    raw = raw_batch(c)
    b = dev_batch(raw, dev)
    oracle, _ = fx.oracle_from_config(c)
    with torch.no_grad():
        ref_out, ref_ae = oracle.forward(fx.oracle_batch(raw))
        got, _ = _run(model, b, fused=True, train=False)
    assert relmax(got[0], ref_out) < 1e-2, relmax(got[0], ref_out)
    for g, r in zip(got[1:], ref_ae):
        assert relmax(g, r) < 1e-2, relmax(g, r)


@pytest.mark.parametrize("name", ["query_b5", "shared_b7", "query_b64"])
@pytest.mark.parametrize("dropout", [0.0, 0.1], ids=["nodrop", "drop"])
def test_fused_backward_equals_four_launch_path(dev, name, dropout):
    """Gradients of every parameter with the fused launches on vs off — without dropout, and with the same dropout streams (the
    backward kernel regenerates the attention-dropout mask from the seed).  The backward consumes the saved buffers of whichever
    forward ran, so the two gradients differ by the forward's bf16 rounding differences; a hidden unit whose pre-activation sits
    at a rounding boundary of zero switches its ReLU gate, which moves single entries of the w_1 gradient by a few percent of the
    largest entry.  Measured over 6 seeds x 2 shapes (tools/dbg_fused_bwd_spread.py): per-tensor cosine 0.99986-0.99996 and
    ||diff|| / ||ref|| 0.9-1.7e-2 under dropout (cosine > 0.9999 without), single entries <= 1.5e-2 of the largest (w_1: <= 5.8e-2).
    Bars: cosine 0.9999 (0.9997 under dropout), ||diff||/||ref|| 2.5e-2, single entries 5e-2 (w_1: 1e-1)."""
    c = CFGS[name]
    torch.manual_seed(0)
    model = build_model(c, torch.bfloat16, dev, dropout=dropout, attn_dropout=dropout).train()
    b = dev_batch(raw_batch(c), dev)
    model.prepare()
    model._seed.fill_(1234)                    # (the dropout seed defaults to torch.initial_seed(): pin it)
    seed0 = model._seed.clone()
    _, gref = _run(model, b, fused=False, train=True)
    model._seed.copy_(seed0)
    _, ggot = _run(model, b, fused=True, train=True)
    assert gref.keys() == ggot.keys()
    for k in gref:
        r, g = gref[k].float().flatten(), ggot[k].float().flatten()
        if float(r.abs().max()) == 0.0 or k.endswith("linears.1.bias"):
            continue        # key-projection biases: mathematically zero gradient (a key bias shifts every score of a row equally) = rounding noise
        cos = float(torch.dot(r, g) / (r.norm() * g.norm() + 1e-30))
        assert cos > (0.9997 if dropout > 0 else 0.9999), (k, cos)
        assert float((g - r).norm() / r.norm()) < 2.5e-2, (k, float((g - r).norm() / r.norm()))
        assert relmax(g, r) < (1e-1 if ".w_1." in k else 5e-2), (k, relmax(g, r))


def _rand_cfg(seed):
    import random
    r = random.Random(seed)
    nF = r.choice([1, 2])
    return dict(vocab=r.randint(40, 150), N=1, d_model=512, d_ff=r.choice([1024, 2048]), h=8, ft_sizes=[r.choice([24, 40, 64]) for _ in range(nF)],
                B=r.randint(1, 9), Q=r.randint(2, 34), H=r.randint(2, 150), C=r.randint(3, 70), T=r.randint(2, 34),     # (captions: the fixture generator draws >= 3 tokens)
                frames=[r.randint(2, 40) for _ in range(nF)], diff_encoder=r.random() < 0.7, diff_embed=False, diff_gen=False,
                auto_encoder_ft=r.choice(["query", "caption"]))


# Round 3 ran 80 seeds once: 78 passed, one sat at cosine 0.99896 (3 samples) and one drew a shape the fixture generator cannot
# build.  Round 4 (LayerNorm backward in the GEMM epilogue, MTN_FUZZ_N=80): the 78 drawable seeds all pass at the bars below — the
# near-miss did not reproduce — and the two undrawable ones (a 2-token caption) are gone from the generator above.
@pytest.mark.parametrize("seed", list(range(int(os.environ.get("MTN_FUZZ_N", "10")))))
def test_fused_random_shapes_equal_four_launch_path(dev, seed):
    """Random batch sizes and sequence lengths (1 .. just past the kernels' tilings: 32 query rows, 128-row memories, odd sample
    counts per workgroup, two-row sequences), dropout on: outputs and gradients with the fused launches on vs off.  Shapes outside
    a fused kernel's tiling take the per-stage path in both runs.  The gradient bars are wider than in the fixed-shape tests: with
    2-3 samples a tensor's gradient is a sum over few rows and the forward's bf16 rounding differences average out less (measured
    cosine down to 0.9995); an indexing or masking error moves them by orders of magnitude more."""
    c = _rand_cfg(seed)
    torch.manual_seed(seed)
    model = build_model(c, torch.bfloat16, dev, dropout=0.1, attn_dropout=0.1).train()
    b = dev_batch(raw_batch(c), dev)
    model.prepare()
    model._seed.fill_(777 + seed)
    seed0 = model._seed.clone()
    oref, gref = _run(model, b, fused=False, train=True)
    model._seed.copy_(seed0)
    ogot, ggot = _run(model, b, fused=True, train=True)
    for r, g in zip(oref, ogot):
        assert torch.isfinite(g).all()
        assert relmax(g, r) < 5e-3, (c, relmax(g, r))
    for k in gref:
        r, g = gref[k].float().flatten(), ggot[k].float().flatten()
        if float(r.abs().max()) == 0.0 or k.endswith("linears.1.bias"):
            continue
        assert torch.isfinite(g).all(), k
        cos = float(torch.dot(r, g) / (r.norm() * g.norm() + 1e-30))
        assert cos > 0.999, (c, k, cos)
        assert float((g - r).norm() / r.norm()) < 5e-2, (c, k, float((g - r).norm() / r.norm()))


def _free_inputs(c, T, dev, seed=5):
    d = c["d_model"]
    g = torch.Generator().manual_seed(seed)
    shapes = dict(x=(c["B"], T, d), cap=(c["B"], c["C"], d), his=(c["B"], c["H"], d), q=(c["B"], c["Q"], d),
                  v0=(c["B"], c["frames"][0], d), v1=(c["B"], c["frames"][1], d), ae0=(c["B"], c["Q"], d), ae1=(c["B"], c["Q"], d))
    host = {k: torch.randn(*s_, generator=g) for k, s_ in shapes.items()}
    gy = {k: torch.randn(*shapes[k], generator=g) for k in ("x", "ae0", "ae1")}
    return host, gy


def _dev_leaves(host, dev):
    from mtn_amd import ops
    dl = {k: t.to(dev).requires_grad_() for k, t in host.items()}
    for k in ("cap", "his", "q", "v0", "v1"):
        dl[k]._mtn_lp = ops.cast_to_lp(dl[k].detach(), torch.bfloat16)       # what the Encoder leaves on a memory
    return dl


@pytest.mark.parametrize("T,H", [(20, 37), (31, 37), (40, 37), (54, 37), (64, 37), (20, 300), (20, 520), (40, 520), (20, 576),
                                 (56, 192), (56, 472), (64, 300), (50, 160)])
def test_fused_group_backward_matches_oracle_directly(dev, T, H):
    """ONE lockstep group through mtn_sublayer_group_fwd / _bwd with the fused kernels on (csrc/fused.hip, fused_bwd.hip), d_model
    512 / 8 heads, three attention members of the three kinds the kernels serve, each with its own free input:
      * self-attention of the target stream (causal + padding mask), mtn.py:183;
      * cross-attention over a memory whose K|V were projected ahead of the layer loop — the history, with an EMPTY row (fully
        masked: uniform attention, -1e9 semantics), mtn.py:185;
      * cross-attention over an un-projected memory (the target stream attends an auto-encoder output), mtn.py:215;
    against the oracle's autograd of x + multi_head_attention(layer_norm(x), mem, mem, mask) (mtn.py:125-127, 248-267, 221-231),
    dropout off.  Outputs and the dq/dk/dv-side gradients — x.grad, mem.grad, w_qkv.grad, b_q/b_v.grad, w_o.grad — at the bf16 bar
    of 2e-2 relative to the tensor's largest entry (the fused backward kernel had only been compared with the
    four-launch path).  T = 40 / 54 / 64 are AVSD's longer targets: 2 query blocks of 32 in the backward kernel.
    H = 300 / 520 / 576 are long histories (BASELINE configs[3] has 512 tokens; data_handler.py:182 lets them grow): the forward
    kernel puts the V image over the dead xn image and splits the keys over its 8 waves, the backward kernel streams K / V
    through its two-slot key ring.  T = 50 / 56 / 64 WITH H >= 160 (round 6) are the ragged corpus' own shapes — an AVSD-length answer
    attending a long history: four row tiles per workgroup (a 64 KB dy image) beside the key ring; until round 5 that member, and with it
    its whole group, left the fused backward kernel."""
    from mtn_amd import lib, ops
    c = dict(vocab=80, N=1, d_model=512, d_ff=2048, h=8, ft_sizes=[64, 32], B=5, Q=13, H=H, C=29, T=T, frames=[17, 9],
             diff_encoder=True, diff_embed=False, diff_gen=False, auto_encoder_ft="query")
    model = build_model(c, torch.bfloat16, dev).train()             # dropout 0: train mode only to keep the autograd path
    raw = raw_batch(c)
    b = dev_batch(raw, dev)
    ob = fx.oracle_batch(raw)
    model.prepare()
    host, _ = _free_inputs(c, T, dev)
    g = torch.Generator().manual_seed(17)
    xs = [torch.randn(c["B"], T, c["d_model"], generator=g) for _ in range(3)]
    gys = [torch.randn(c["B"], T, c["d_model"], generator=g) for _ in range(3)]
    L = "decoder.layers.0"
    # ---- oracle
    oracle, _ = fx.oracle_from_config(c, requires_grad=True)
    ox = [t.clone().requires_grad_() for t in xs]
    ohis, oae = host["his"].clone().requires_grad_(), host["ae0"].clone().requires_grad_()
    oy = [oracle._sublayer(L, 0, ox[0], lambda y: oracle._mha(L + ".self_attn", y, y, ob.trg_mask)),
          oracle._sublayer(L, 1, ox[1], lambda y: oracle._mha(L + ".his_attn", y, ohis, ob.his_mask)),
          oracle._sublayer(L, 7, ox[2], lambda y: oracle._mha(L + ".auto_encoder_attn.0", y, oae, ob.query_mask))]
    sum((y * gy).sum() for y, gy in zip(oy, gys)).backward()
    # ---- HIP: the same three sublayers as ONE group
    prev = lib.load().mtn_fused_enable(1)
    try:
        layer = model.decoder.layers[0]
        sl = layer.sublayer
        dl = _dev_leaves(host, dev)
        dx = [t.to(dev).requires_grad_() for t in xs]
        ops.prepare_masks(b.trg_mask, b.his_mask, b.query_mask)
        model.zero_glue_grads()
        c0 = lib.fused_counters()
        model.hoist_memory_kv(dl["cap"], dl["his"], dl["q"], [dl["v0"], dl["v1"]])
        try:
            ys = layer._run_group([(sl[0], layer.self_attn, None, b.trg_mask, dx[0]), (sl[1], layer.his_attn, dl["his"], b.his_mask, dx[1]),
                                   (sl[7], layer.auto_encoder_attn[0], dl["ae0"], b.query_mask, dx[2])])
        finally:
            model.clear_memory_kv()
        sum((y * gy.to(dev)).sum() for y, gy in zip(ys, gys)).backward()
        torch.cuda.synchronize()
        c1 = lib.fused_counters()
    finally:
        lib.load().mtn_fused_enable(1 if prev != 0 else 0)
    assert (c1[0] - c0[0], c1[1] - c0[1]) == (1, 0), "the forward group left the fused kernel"
    assert (c1[2] - c0[2], c1[3] - c0[3]) == (1, 0), "the backward group left the fused kernel"
    for y, r in zip(ys, oy):
        assert relmax(y, r) < 1e-2, relmax(y, r)
    for got, want, name in [(dx[0].grad, ox[0].grad, "x(self)"), (dx[1].grad, ox[1].grad, "x(history)"), (dx[2].grad, ox[2].grad, "x(ae)"),
                            (dl["his"].grad, ohis.grad, "history memory"), (dl["ae0"].grad, oae.grad, "auto-encoder memory")]:
        assert relmax(got, want) < 2e-2, (name, relmax(got, want))
    sd = dict(model.named_parameters())
    for n in ("self_attn", "his_attn", "auto_encoder_attn.0"):
        for j in range(4):
            for part in ("weight", "bias"):
                if j == 1 and part == "bias":
                    continue            # key bias: mathematically zero gradient (shifts every score of a row equally)
                key = f"{L}.{n}.linears.{j}.{part}"
                assert relmax(sd[key].grad, oracle.p[key].grad) < 2e-2, (key, relmax(sd[key].grad, oracle.p[key].grad))


@pytest.mark.parametrize("T", [20, 40, 54])
def test_fused_layer_gradients_match_oracle(dev, T):
    """A whole DecoderLayer (7 forward groups, 6 backward groups with attention members, all on the fused kernels) driven with
    free inputs against the oracle's autograd of mtn.py:183-218.  The gradient of the target stream passes through seven chained
    bf16 sublayers here (the group-level test above holds single sublayers to 2e-2): bars 4e-2 relative to max and cosine 0.9995."""
    from mtn_amd import lib, ops
    c = dict(vocab=80, N=1, d_model=512, d_ff=2048, h=8, ft_sizes=[64, 32], B=5, Q=13, H=37, C=29, T=T, frames=[17, 9],
             diff_encoder=True, diff_embed=False, diff_gen=False, auto_encoder_ft="query")
    model = build_model(c, torch.bfloat16, dev).train()
    raw = raw_batch(c)
    b = dev_batch(raw, dev)
    ob = fx.oracle_batch(raw)
    model.prepare()
    host, gy = _free_inputs(c, T, dev)
    oracle, _ = fx.oracle_from_config(c, requires_grad=True)
    ol = {k: t.clone().requires_grad_() for k, t in host.items()}
    ox, oae = oracle.decoder_layer(0, ol["x"], ol["cap"], ob.cap_mask, ol["his"], ob.his_mask, ol["q"], ob.query_mask, ob.trg_mask,
                                   [ol["v0"], ol["v1"]], ob.fts_mask, [ol["ae0"], ol["ae1"]])
    ((ox * gy["x"]).sum() + (oae[0] * gy["ae0"]).sum() + (oae[1] * gy["ae1"]).sum()).backward()
    prev = lib.load().mtn_fused_enable(1)
    try:
        dl = _dev_leaves(host, dev)
        ops.prepare_masks(b.trg_mask, b.his_mask, b.cap_mask, b.query_mask, b.fts_mask)
        model.zero_glue_grads()
        c0 = lib.fused_counters()
        model.hoist_memory_kv(dl["cap"], dl["his"], dl["q"], [dl["v0"], dl["v1"]])
        try:
            y, aes = model.decoder.layers[0](dl["x"], dl["cap"], b.cap_mask, dl["his"], b.his_mask, dl["q"], b.query_mask, b.trg_mask,
                                             [dl["v0"], dl["v1"]], b.fts_mask, [dl["ae0"], dl["ae1"]], "query")
        finally:
            model.clear_memory_kv()
        loss = (y * gy["x"].to(dev)).sum() + (aes[0] * gy["ae0"].to(dev)).sum() + (aes[1] * gy["ae1"].to(dev)).sum()
        loss.backward()
        torch.cuda.synchronize()
        c1 = lib.fused_counters()
    finally:
        lib.load().mtn_fused_enable(1 if prev != 0 else 0)
    assert c1[0] - c0[0] == 7 and c1[1] == c0[1], "a forward group left the fused kernel"
    assert c1[2] - c0[2] == 6 and c1[3] == c0[3], "a backward group with attention members left the fused kernel"
    assert relmax(y, ox) < 1e-2 and relmax(aes[0], oae[0]) < 1e-2 and relmax(aes[1], oae[1]) < 1e-2
    for k in host:
        got, want = dl[k].grad.float().flatten().cpu(), ol[k].grad.flatten()
        cos = float(torch.dot(got, want) / (got.norm() * want.norm()))
        assert relmax(got, want) < 4e-2 and cos > 0.9995, (k, relmax(got, want), cos)


def _round_like_the_device(sd, c):
    """The fixture weights as the bf16 path sees them: every matrix that is a GEMM operand on the device (the compute-dtype copy of
    the flat parameter buffer) rounded to bf16; vectors (biases, LayerNorm gains) and the embedding tables stay fp32 there."""
    out = {}
    for k, v in sd.items():
        is_operand = v.dim() == 2 and (".linears." in k or ".w_1." in k or ".w_2." in k or k.startswith("vid_encoder.") or "generator" in k)
        out[k] = (v.to(torch.bfloat16) if is_operand else v).double()
    return out


@pytest.mark.parametrize("name", ["query_b5", "caption_b3"])
def test_fused_gradients_match_fp64_oracle_on_rounded_operands(dev, name):
    """The tight version of the bf16 gradient check: the oracle runs in fp64 on exactly the operands the
    device path sees (weights and input features rounded to bf16), so what is left between the two is the path's own rounding of
    ACTIVATIONS (xn, q|k|v, P, o, hidden, the bf16 gradient operands) — not the 2^-9 perturbation of every weight that the plain
    bf16-vs-fp32 comparison of tests/test_model_gpu.py has to allow for with its 0.25 per-tensor bar.  Fused kernels on (forward,
    backward, LayerNorm backward in the GEMM epilogue), dropout off, loss = sum of the mean squares of the decoder outputs."""
    from mtn_amd import lib
    from oracle.mtn_oracle import OracleMTN
    c = CFGS[name]
    model = build_model(c, torch.bfloat16, dev).eval()
    raw = raw_batch(c)
    b = dev_batch(raw, dev)
    n0 = lib.fused_counters()
    _, got = _run(model, b, fused=True, train=True)
    n1 = lib.fused_counters()
    assert n1[0] > n0[0] and n1[2] > n0[2], "the fused kernels did not run"
    _, ocfg = fx.oracle_from_config(c)
    sd = {k: v.clone().requires_grad_(True) for k, v in _round_like_the_device(fx.det_state_dict(fx.state_shapes(**c), 0), c).items()}
    ob = fx.oracle_batch(raw)
    ob.fts = [f.to(torch.bfloat16).double() for f in ob.fts]
    out, ae = OracleMTN(ocfg, sd).forward(ob)
    loss = (out ** 2).mean() + sum((a ** 2).mean() for a in ae)
    loss.backward()
    worst, worst_k, dot, n1_, n2_ = 0.0, None, 0.0, 0.0, 0.0
    for k, g in got.items():
        ref = sd[k].grad
        if ref is None or float(ref.abs().max()) == 0.0 or k.endswith("linears.1.bias"):
            continue
        g64 = g.double().cpu()
        e = relmax(g64, ref)
        if e > worst and ".w_1." not in k:
            worst, worst_k = e, k
        dot += float((g64 * ref).sum()); n1_ += float(g64.norm()) ** 2; n2_ += float(ref.norm()) ** 2
        cos = float((g64 * ref).sum() / (g64.norm() * ref.norm() + 1e-300))
        # (w_1: a hidden unit whose pre-activation rounds across zero flips its ReLU gate and moves single entries of this gradient
        #  by a few percent of the largest one — the same effect, and bar, as in test_fused_backward_equals_four_launch_path)
        assert e < (1e-1 if ".w_1." in k else 3e-2), (k, e)
        assert cos > 0.9995, (k, cos)
    cos_all = dot / (n1_ ** 0.5 * n2_ ** 0.5)
    print(f"{name}: worst per-tensor gradient error (relative to max, w_1 aside) {worst:.2e} at {worst_k}; cosine over all tensors {cos_all:.7f}")
    assert cos_all > 0.99995, cos_all
