"""GPU parity of the individual HIP kernels (through the C ABI) against the CPU oracle's leaf functions
(oracle/mtn_oracle.py, pinned to the reference by tests/test_oracle_golden.py) on the same seeded inputs."""
import math

import pytest
import torch

from tests.util import DTYPES, TOL, absmax, lp_round, relmax

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


def _gemm_problem(L, A, B, M, N, K, at, bt, lda, ldb):
    p = L.GemmProblem()
    p.A, p.B, p.lda, p.ldb, p.M, p.N, p.K, p.a_trans, p.b_trans, p.gate_scale = A.data_ptr(), B.data_ptr(), lda, ldb, M, N, K, at, bt, 1.0
    return p


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("at,bt", [(0, 0), (0, 1), (1, 1), (1, 0)])
@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (104, 72, 136), (640, 512, 512), (20, 1536, 128), (384, 128, 1000)])
def test_gemm_layouts(dev, dtype, at, bt, M, N, K):
    """Asymmetric random operands (catches row/col swaps of the MFMA fragment maps), ragged M/N/K."""
    from mtn_amd import lib as L, ops
    if at and M % 4: pytest.skip("transposed A needs M%4==0")
    if bt and N % 4: pytest.skip("transposed B needs N%4==0")
    if (at and M % 8) or (bt and N % 8): pytest.skip("leading dimension must be a multiple of 16 bytes")
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + at * 2 + bt)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(N, K, generator=g) * torch.linspace(0.5, 1.5, N).unsqueeze(1)
    ref = lp_round(a, dtype).double() @ lp_round(b, dtype).double().t()
    A = (a.t().contiguous() if at else a).to(dev, dtype)
    B = (b.t().contiguous() if bt else b).to(dev, dtype)
    out = torch.full((M, N), float("nan"), device=dev)
    p = _gemm_problem(L, A, B, M, N, K, at, bt, A.size(1), B.size(1))
    p.out_f32, p.ldc = out.data_ptr(), N
    ops.gemm(L.dtype_code(dtype), [p])
    torch.cuda.synchronize()
    tol = 1e-5 if dtype == torch.float32 else 1e-5      # operands pre-rounded: only accumulation order differs
    assert relmax(out, ref) < tol * math.sqrt(K)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_epilogue_and_group(dev, dtype):
    """bias + relu + gate + residual + both outputs + row sums, three problems in one launch."""
    from mtn_amd import lib as L, ops
    g = torch.Generator().manual_seed(5)
    probs, checks = [], []
    for (M, N, K) in [(96, 128, 64), (33, 64, 256), (200, 192, 128)]:
        a, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
        bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
        gate = torch.randn(M, N, generator=g)
        A, B = a.to(dev, dtype), b.to(dev, dtype)
        Bias, Res, Gate = bias.to(dev), res.to(dev), gate.to(dev, dtype)
        of = torch.empty(M, N, device=dev)
        ol = torch.empty(M, N, device=dev, dtype=dtype)
        rs = torch.empty(M, device=dev)
        p = _gemm_problem(L, A, B, M, N, K, 0, 0, K, K)
        p.bias, p.relu, p.gate, p.gate_scale = Bias.data_ptr(), 1, Gate.data_ptr(), 1.25
        p.residual, p.ldr, p.out_f32, p.out_lp, p.ldc, p.rowsum_out = Res.data_ptr(), N, of.data_ptr(), ol.data_ptr(), N, rs.data_ptr()
        probs.append(p)
        ar, br = lp_round(a, dtype).double(), lp_round(b, dtype).double()
        v = torch.relu(ar @ br.t() + bias.double())
        v = torch.where(lp_round(gate, dtype).double() > 0, v * 1.25, torch.zeros_like(v)) + res.double()
        checks.append((of, ol, rs, v, ar.sum(1), (A, B, Bias, Res, Gate)))
    ops.gemm(L.dtype_code(dtype), probs)
    torch.cuda.synchronize()
    for of, ol, rs, v, rsum, _keep in checks:
        assert relmax(of, v) < 2e-4
        assert relmax(ol.float(), v) < (1e-5 if dtype == torch.float32 else 1e-2)
        assert relmax(rs, rsum) < 1e-4


def test_gemm_rejects_misaligned(dev):
    from mtn_amd import lib as L, ops
    A = torch.zeros(64, 70, device=dev, dtype=torch.bfloat16)
    p = _gemm_problem(L, A, A, 64, 64, 70, 0, 0, 70, 70)
    out = torch.empty(64, 64, device=dev)
    p.out_f32, p.ldc = out.data_ptr(), 64
    with pytest.raises(L.MtnHipError):
        ops.gemm(L.MTN_BF16, [p])


# ------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("rows,d", [(1, 64), (37, 128), (640, 512), (130, 2048)])
def test_layernorm_fwd_bwd(dev, rows, d):
    from mtn_amd import ops
    from oracle.mtn_oracle import layer_norm as ref_ln
    g = torch.Generator().manual_seed(rows + d)
    x = (torch.randn(rows, d, generator=g) * 2 + 0.3)
    a2, b2 = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    gy = torch.randn(rows, d, generator=g)
    xr, ar, br = x.clone().requires_grad_(), a2.clone().requires_grad_(), b2.clone().requires_grad_()
    yr = ref_ln(xr, ar, br, 1e-6)
    yr.backward(gy)
    xd, ad, bd = x.to(dev).requires_grad_(), a2.to(dev).requires_grad_(), b2.to(dev).requires_grad_()
    y, y_lp = ops.layer_norm(xd, ad, bd, 1e-6, torch.bfloat16)
    y.backward(gy.to(dev))
    torch.cuda.synchronize()
    assert absmax(y, yr) < 1e-4
    assert relmax(y_lp.float(), yr) < 1e-2
    assert relmax(xd.grad, xr.grad) < 1e-4
    assert relmax(ad.grad, ar.grad) < 1e-4
    assert relmax(bd.grad, br.grad) < 1e-4


@pytest.mark.parametrize("d", [512, 640, 96])
def test_embedding_backward_is_deterministic(dev, d, monkeypatch):
    """Table gradient without atomics (csrc/layernorm.hip embed_bwd_det_kernel): same bits on every run, exact sums against
    fp64 index_add — with a frequent id (40 % pads), more tokens than one staged chunk, d above one 512-column pass."""
    import ctypes as C
    import math
    from mtn_amd import lib as L
    lib = L.load()
    monkeypatch.setenv("MTN_EMBED_DETERMINISTIC", "1")
    __import__("mtn_amd.lib", fromlist=["lib"]).reload_env()
    V = 300
    g = torch.Generator().manual_seed(11)
    shapes = [(6, 1400), (5, 1800), (3, 333)]               # 8400 + 9000 + 999 rows > one 16384-token chunk
    toks, dxs = [], []
    for B, Lq in shapes:
        t = torch.randint(0, V, (B, Lq), generator=g)
        t[torch.rand(B, Lq, generator=g) < 0.4] = 1
        toks.append(t.to(dev).reshape(-1).contiguous())
        dxs.append(torch.randn(B * Lq, d, generator=g).to(dev))
    want = torch.zeros(V, d, dtype=torch.float64, device=dev)
    for t, x in zip(toks, dxs):
        want.index_add_(0, t, x.double() * math.sqrt(d))
    outs = []
    for _ in range(3):
        dlut = torch.zeros(V, d, device=dev)
        descs = (L.EmbedBwdDesc * len(toks))()
        for E, t, x in zip(descs, toks, dxs):
            E.rows, E.d, E.tokens, E.dx, E.emb_scale = t.numel(), d, t.data_ptr(), x.data_ptr(), math.sqrt(d)
            E.dlut, E.lut_rows = dlut.data_ptr(), V
        L.check(lib.mtn_embed_bwd_group(len(toks), descs, L.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(dlut)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert relmax(outs[0].double(), want) < 1e-5


def test_fused_embedding_streams(dev):
    """Embeddings * sqrt(d) + positional encoding (+ dropout) (+ Encoder LayerNorm) for several token streams in one grouped
    launch (mtn.py:288-289, 307-309, 83-101): outputs, table / LayerNorm gradients, and dropout-mask consistency."""
    import math
    from mtn_amd import ops
    from oracle.mtn_oracle import layer_norm as ref_ln, positional_encoding
    V, d, B = 50, 64, 3
    g = torch.Generator().manual_seed(5)
    lut = torch.randn(V, d, generator=g)
    lut2 = torch.randn(V, d, generator=g)
    toks = [torch.randint(0, V, (B, Lq), generator=g) for Lq in (7, 12, 5)]
    lns = [(1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)) for _ in range(2)]
    gys = [torch.randn(B, t.size(1), d, generator=g) for t in toks]
    pe = positional_encoding(64, d)
    # oracle: streams 0,1 share `lut` and are normalised; stream 2 uses lut2 and is not
    lr, l2r = lut.clone().requires_grad_(), lut2.clone().requires_grad_()
    lnr = [(a.clone().requires_grad_(), b.clone().requires_grad_()) for a, b in lns]
    ref = []
    for i, t in enumerate(toks):
        e = (lr if i < 2 else l2r)[t] * math.sqrt(d) + pe[: t.size(1)]
        ref.append(ref_ln(e, lnr[i][0], lnr[i][1], 1e-6) if i < 2 else e)
    torch.autograd.backward(ref, gys)
    # HIP
    ld, l2d = lut.to(dev).requires_grad_(), lut2.to(dev).requires_grad_()
    lnd = [(a.to(dev), b.to(dev), torch.zeros(d, device=dev), torch.zeros(d, device=dev)) for a, b in lns]
    ped = pe.to(dev).contiguous()
    def spec(p, seed):
        streams = [dict(tokens=t.to(dev), lut=0 if i < 2 else 1, pe=ped, scale=math.sqrt(d), p=p, salt=70 + i,
                        ln=(lnd[i][0], lnd[i][1], 1e-6, lnd[i][2], lnd[i][3]) if i < 2 else None) for i, t in enumerate(toks)]
        return dict(streams=streams, lp_dtype=torch.bfloat16, seed=seed, queue=None)
    sp = spec(0.0, None)
    ys = ops.EmbedNormFn.apply(sp, ld, l2d)
    torch.autograd.backward(ys, [t.to(dev) for t in gys])
    torch.cuda.synchronize()
    for i in range(3):
        assert relmax(ys[i], ref[i]) < 1e-5
    assert relmax(sp["_lp_out"][0].float(), ref[0]) < 1e-2 and sp["_lp_out"][2] is None
    assert relmax(ld.grad, lr.grad) < 1e-4 and relmax(l2d.grad, l2r.grad) < 1e-5
    for i in range(2):
        assert relmax(lnd[i][2], lnr[i][0].grad) < 1e-4 and relmax(lnd[i][3], lnr[i][1].grad) < 1e-4
    # dropout on the un-normalised stream: kept elements scaled by 1/(1-p); backward uses the same mask
    seed = torch.full((1,), 1234, device=dev, dtype=torch.int64)
    l2d.grad = None
    ld.grad = None
    ys = ops.EmbedNormFn.apply(spec(0.25, seed), ld, l2d)
    y2 = ys[2]
    keep = (y2 != 0)
    frac = keep.float().mean().item()
    assert 0.65 < frac < 0.85
    assert absmax(y2, ref[2].detach().to(dev) * keep / 0.75) < 1e-4
    torch.autograd.backward(ys, [t.to(dev) for t in gys])
    torch.cuda.synchronize()
    want = torch.zeros(V, d, device=dev).index_add_(0, toks[2].to(dev).reshape(-1),
                                                    (gys[2].to(dev) * keep / 0.75).reshape(-1, d) * math.sqrt(d))
    assert relmax(l2d.grad, want) < 1e-5


def _feature_spec(dev, x, w, b, lnp, grads, pe, p, seed, lp):
    streams = [dict(x=x, w_lp=w if lp == torch.float32 else w.to(lp), bias=b, grad_w=grads[0], grad_b=grads[1], pe=pe, p=p, salt=33,
                    ln=(lnp[0], lnp[1], 1e-6, grads[2], grads[3]))]
    return dict(streams=streams, lp_dtype=lp, seed=seed, queue=None)


@pytest.mark.parametrize("dtype", DTYPES)
def test_feature_stream_encode(dev, dtype):
    """vid_encoder (Linear -> ReLU -> +PE -> dropout, mtn.py:378) + Encoder LayerNorm on the HIP path: outputs and parameter
    gradients against the oracle without dropout; with dropout on, the backward uses the forward's mask (directional
    derivative of the loss along a bias perturbation, same seed, fp32 mode)."""
    from mtn_amd import ops
    from oracle.mtn_oracle import layer_norm as ref_ln, positional_encoding
    B, V, F, d = 3, 9, 40, 64
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, V, F, generator=g)
    w, b = torch.randn(d, F, generator=g) * F ** -0.5, 0.1 * torch.randn(d, generator=g)
    a2, b2 = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    gy = torch.randn(B, V, d, generator=g)
    pe = positional_encoding(32, d)
    leaves = [t.clone().requires_grad_() for t in (w, b, a2, b2)]
    yr = ref_ln(torch.relu(x @ leaves[0].t() + leaves[1]) + pe[:V], leaves[2], leaves[3], 1e-6)
    yr.backward(gy)
    D = lambda t: t.to(dev)
    wd, bd, ad, b2d = D(w).requires_grad_(), D(b), D(a2), D(b2)
    grads = [torch.zeros_like(t) for t in (wd, bd, ad, b2d)]
    spec = _feature_spec(dev, D(x), wd.detach(), bd, (ad, b2d), grads, D(pe).contiguous(), 0.0, None, dtype)
    y = ops.FeatureEncodeFn.apply(spec, wd)[0]
    y.backward(D(gy))
    torch.cuda.synchronize()
    tol = TOL[dtype]
    assert relmax(y, yr) < tol
    for got, ref in zip(grads, leaves):
        if dtype == torch.float32:
            assert relmax(got, ref.grad) < tol * 3
        else:      # bf16 operands: rounding of the pre-LayerNorm activation is amplified by the LayerNorm backward projection
            cos = torch.nn.functional.cosine_similarity(got.flatten().cpu().double(), ref.grad.flatten().double(), dim=0)
            assert relmax(got, ref.grad) < 0.25 and cos > 0.999
    if dtype != torch.float32:
        assert relmax(spec["_lp_out"][0].float(), yr) < 2e-2
        return
    seed = torch.full((1,), 77, device=dev, dtype=torch.int64)
    v = D(torch.randn(d, generator=g))

    def loss_at(bias):
        sp = _feature_spec(dev, D(x), wd.detach(), bias, (ad, b2d), [torch.zeros_like(t) for t in (wd, bd, ad, b2d)], D(pe).contiguous(), 0.3, seed, dtype)
        out = ops.FeatureEncodeFn.apply(sp, wd)[0]
        return (out * D(gy)).sum(), sp

    l0, sp0 = loss_at(bd)
    l0.backward()
    torch.cuda.synchronize()
    gb = sp0["streams"][0]["grad_b"]
    eps = 1e-3
    num = (float(loss_at(bd + eps * v)[0]) - float(loss_at(bd - eps * v)[0])) / (2 * eps)
    ana = float((gb * v).sum())
    assert abs(num - ana) < 2e-2 * max(1.0, abs(ana)), (num, ana)


# ------------------------------------------------------------------------------------------ attention core
def _attn_ref(q, k, v, mask, h):
    from oracle.mtn_oracle import scaled_dot_attention
    B, a, d = q.shape
    sp = lambda t: t.reshape(B, -1, h, d // h).transpose(1, 2)
    o, p = scaled_dot_attention(sp(q), sp(k), sp(v), None if mask is None else mask.unsqueeze(1))
    return o.transpose(1, 2).reshape(B, a, d), p


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,h,a,m,dk,kind", [
    (2, 4, 20, 20, 32, "causal"), (3, 8, 20, 128, 64, "pad"), (2, 2, 7, 37, 16, "pad"), (1, 4, 20, 300, 64, "pad"),
    (2, 4, 33, 70, 64, "none"), (2, 8, 54, 54, 64, "causal"), (2, 8, 20, 520, 64, "pad"), (1, 4, 20, 1000, 64, "pad"),
    (2, 8, 100, 100, 64, "causal"), (1, 4, 238, 45, 64, "pad"), (2, 2, 70, 33, 32, "none"), (2, 4, 256, 256, 64, "causal")])
def test_attention_fwd_bwd(dev, dtype, B, h, a, m, dk, kind):
    """Causal / key-padding / absent masks, a fully masked row (uniform attention, zero score-gradient), key counts that
    are not multiples of the 64/32-key tiles, multi-tile online softmax, and query counts beyond 64 rows (the backward walks
    the queries in passes of 32 and accumulates dK / dV — in an fp32 workspace, rounded to bf16 once by the last pass: captions as
    the auto-encoder stream reach 238 tokens; a = 256 is eight passes)."""
    from mtn_amd import ops
    d = h * dk
    g = torch.Generator().manual_seed(B * 100 + a + m)
    q, k, v = (lp_round(torch.randn(B, L, d, generator=g), dtype) for L in (a, m, m))
    go = lp_round(torch.randn(B, a, d, generator=g), dtype)
    if kind == "causal":
        mask = torch.tril(torch.ones(1, a, m, dtype=torch.bool)).expand(B, a, m).clone()
        mask[0, :, -3:] = False
    elif kind == "pad":
        lens = torch.randint(1, m + 1, (B,), generator=g)
        mask = (torch.arange(m).unsqueeze(0) < lens.unsqueeze(1)).unsqueeze(1)
        mask[B - 1] = False                      # an empty memory: every score masked -> uniform attention
    else:
        mask = None
    qr, kr, vr = q.clone().requires_grad_(), k.clone().requires_grad_(), v.clone().requires_grad_()
    orf, _ = _attn_ref(qr, kr, vr, mask, h)
    orf.backward(go)
    qd, kd, vd = (t.to(dev, dtype) for t in (q, k, v))
    md = None if mask is None else mask.to(dev)
    o, lse = ops.attention(qd, kd, vd, md, h)
    dq, dk_, dv = ops.attention_bwd(qd, kd, vd, o, lse, go.to(dev, dtype), md, h)
    torch.cuda.synchronize()
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    assert relmax(o.float(), orf) < tol
    assert relmax(dq.float(), qr.grad) < tol * 2
    assert relmax(dk_.float(), kr.grad) < tol * 2
    assert relmax(dv.float(), vr.grad) < tol * 2


def test_attention_dropout_statistics_and_consistency(dev):
    """Keep-rate of the counter-based mask, and backward uses the same mask as forward: with V = identity-like
    probes the dropped entries of P are visible in O, and dV must vanish exactly where P was dropped."""
    from mtn_amd import ops
    B, h, a, m, dk = 2, 2, 16, 64, 64
    d = h * dk
    seed = torch.tensor([12345], device=dev, dtype=torch.int64)
    q = torch.zeros(B, a, d, device=dev)                      # uniform attention: P = 1/m everywhere
    k = torch.randn(B, m, d, device=dev)
    v = torch.zeros(B, m, d, device=dev)
    for hh in range(h):
        v[:, :, hh * dk:(hh + 1) * dk] = torch.eye(m, dk, device=dev)        # O[i, j] = Pdrop[i, j] for j < dk
    o, lse = ops.attention(q, k, v, None, h, p_drop=0.25, seed=seed, salt=3)
    kept = (o != 0).float().mean().item()
    assert abs(kept - 0.75) < 0.03
    nz = o[o != 0]
    assert torch.allclose(nz, torch.full_like(nz, (1.0 / m) / 0.75), rtol=1e-4)
    o2, _ = ops.attention(q, k, v, None, h, p_drop=0.25, seed=seed, salt=3)
    assert torch.equal(o, o2)                                  # pure function of (seed, salt, index)
    o3, _ = ops.attention(q, k, v, None, h, p_drop=0.25, seed=seed + 1, salt=3)
    assert not torch.equal(o, o3)
    go = torch.zeros_like(o)
    go[:, 0, :] = 1.0                                          # only query row 0 of every head sends gradient
    _, _, dv = ops.attention_bwd(q, k, v, o, lse, go, None, h, p_drop=0.25, seed=seed, salt=3)
    torch.cuda.synchronize()
    for b in range(B):
        for hh in range(h):
            p_row0 = o[b, 0, hh * dk:(hh + 1) * dk]            # = Pdrop[0, j]
            dv_col = dv[b, :dk, hh * dk]                       # dV[j, c] = Pdrop[0, j] * go[0, c]
            assert torch.allclose(dv_col, p_row0, rtol=1e-4, atol=1e-7)

def test_gemm_tt_dma_128_tile_group(dev):
    """bf16 parameter-gradient GEMMs on the 128x128-tile LDS-DMA kernel (transposing LDS reads): grouped, ragged output sizes,
    ragged contraction (not a multiple of the 64-row stage), row sums = bias gradients."""
    import os
    from mtn_amd import lib as L, ops
    dtype = torch.bfloat16
    os.environ["MTN_GEMM_TTB_MIN_TILES"] = "1"
    L.reload_env()
    g = torch.Generator().manual_seed(10)
    probs, checks = [], []
    for (M, N, K) in [(512, 512, 640), (1536, 512, 100), (200, 136, 333), (3000, 512, 64), (128, 2048, 4096)]:
        a, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
        A, B = a.t().contiguous().to(dev, dtype), b.t().contiguous().to(dev, dtype)
        out = torch.full((M, N), float("nan"), device=dev)
        rs = torch.full((M,), float("nan"), device=dev)
        p = _gemm_problem(L, A, B, M, N, K, 1, 1, M, N)
        p.out_f32, p.ldc, p.rowsum_out = out.data_ptr(), N, rs.data_ptr()
        probs.append(p)
        ar, br = lp_round(a, dtype).double(), lp_round(b, dtype).double()
        checks.append((out, rs, ar @ br.t(), ar.sum(1), (A, B)))
    try:
        L.load().mtn_census_begin()
        ops.gemm(L.dtype_code(dtype), probs)
        torch.cuda.synchronize()
        assert L.load().mtn_census_end() == 1
        info = L.CensusLaunch()
        import ctypes as C
        L.check(L.load().mtn_census_info(0, C.byref(info)))
        assert L.load().mtn_census_variant_name(info.variant).decode() == "gemm_tt_dma128_kernel"
    finally:
        del os.environ["MTN_GEMM_TTB_MIN_TILES"]
        L.reload_env()
    for out, rs, ref, rsum, _keep in checks:
        assert relmax(out, ref) < 1e-4
        assert relmax(rs, rsum) < 1e-4


@pytest.mark.parametrize("M,N,K", [(300, 200, 72), (128, 1024, 512), (1000, 136, 1000)])
def test_gemm_dma_128_tile_row_major(dev, M, N, K):
    """The opt-in 128x128-tile LDS-DMA kernel for row-major x row-major problems (ragged M/N, K not a multiple of the
    64-element stage, bias + ReLU + residual epilogue, fp32 and bf16 outputs)."""
    import os
    from mtn_amd import lib as L, ops
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(M + N + K)
    a, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    A, B = a.to(dev, dtype), b.to(dev, dtype)
    out = torch.full((M, N), float("nan"), device=dev)
    out_lp = torch.empty(M, N, device=dev, dtype=dtype)
    bd, rd = bias.to(dev), res.to(dev)
    p = _gemm_problem(L, A, B, M, N, K, 0, 0, K, K)
    p.bias, p.relu, p.residual, p.ldr, p.out_f32, p.out_lp, p.ldc = bd.data_ptr(), 1, rd.data_ptr(), N, out.data_ptr(), out_lp.data_ptr(), N
    os.environ["MTN_GEMM_NTB_MIN_TILES"] = "1"
    L.reload_env()
    try:
        ops.gemm(L.dtype_code(dtype), [p])
        torch.cuda.synchronize()
    finally:
        del os.environ["MTN_GEMM_NTB_MIN_TILES"]
        L.reload_env()
    ref = torch.relu(lp_round(a, dtype).double() @ lp_round(b, dtype).double().t() + bias.double()) + res.double()
    assert relmax(out, ref) < 1e-4 and relmax(out_lp.float(), ref) < 1e-2


@pytest.mark.parametrize("with_adam", [False, True], ids=["plain", "optimiser-epilogue"])
def test_gemm_tt_table_form(dev, with_adam):
    """All parameter-gradient problems in one launch (mtn_gemm_tt_table): ragged output sizes (partial 128-tiles), ragged
    contraction, row sums, many problems; with the optimiser epilogue (mtn_adam_fuse) the result is Adam applied to the weight
    block — p, m, v, the compute-dtype copy and the TRANSPOSED copy, including a row block of a taller weight (ldT > M) —
    checked against the same update done in fp64 on the host; launched twice (the second launch reuses a staging slot)."""
    import ctypes as C
    from mtn_amd import lib as L
    lib = L.load()
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(21)
    shapes = [(512, 512, 640), (1536, 512, 100), (200, 136, 333), (128, 2048, 1000), (8, 8, 8), (384, 128, 64)] + [(128, 128, 96)] * 20
    state = torch.tensor([3.0, 2e-3, 1 - 0.9 ** 3, 1 - 0.98 ** 3, 0, 0, 0, 0], device=dev)
    probs, keep, checks = [], [], []
    for idx, (M, N, K) in enumerate(shapes):
        a, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
        A, B = a.t().contiguous().to(dev, dtype), b.t().contiguous().to(dev, dtype)
        out = torch.full((M, N), float("nan"), device=dev)
        rs = torch.full((M,), float("nan"), device=dev)
        p = _gemm_problem(L, A, B, M, N, K, 1, 1, M, N)
        p.out_f32, p.ldc, p.rowsum_out = out.data_ptr(), N, rs.data_ptr()
        ar, br = lp_round(a, dtype).double(), lp_round(b, dtype).double()
        ref = ar @ br.t()
        entry = dict(out=out, rs=rs, ref=ref, rsum=ar.sum(1))
        if with_adam:
            rows_total, r0 = (M + 64, 24) if idx % 2 else (M, 0)         # odd problems: a row block of a taller weight
            w = torch.randn(M, N, generator=g).to(dev)
            m0, v0 = (0.01 * torch.randn(M, N, generator=g)).to(dev), (1e-4 * torch.rand(M, N, generator=g)).to(dev)
            p_, m_, v_ = w.clone(), m0.clone(), v0.clone()
            lp = torch.zeros(M, N, device=dev, dtype=dtype)
            lpT = torch.full((N, rows_total), 7.0, device=dev, dtype=dtype)
            f = L.AdamFuse()
            f.p, f.m, f.v, f.p_lp = p_.data_ptr(), m_.data_ptr(), v_.data_ptr(), lp.data_ptr()
            f.p_lpT, f.ldT = lpT.data_ptr() + 2 * r0, rows_total
            f.write_grad, f.state, f.grad_scale, f.beta1, f.beta2, f.eps = (idx % 3 == 0), state.data_ptr(), None, 0.9, 0.98, 1e-9
            p.adam = C.pointer(f)
            entry.update(w=w, m0=m0, v0=v0, p=p_, m=m_, v=v_, lp=lp, lpT=lpT, r0=r0, wg=bool(f.write_grad))
            keep.append(f)
        probs.append(p)
        keep += [A, B]
        checks.append(entry)
    arr = (L.GemmProblem * len(probs))(*probs)
    for launch in range(2):
        L.check(lib.mtn_gemm_tt_table(L.dtype_code(dtype), len(probs), arr, L.stream_ptr()))
        torch.cuda.synchronize()
        for e in checks:
            assert relmax(e["rs"], e["rsum"]) < 1e-4
            if not with_adam:
                assert relmax(e["out"], e["ref"]) < 1e-4
                continue
            gref = e["ref"]
            if launch == 0:
                e["wm"], e["mm"], e["vm"] = e["w"].double().cpu(), e["m0"].double().cpu(), e["v0"].double().cpu()
            mm = 0.9 * e["mm"] + 0.1 * gref
            vv = 0.98 * e["vm"] + 0.02 * gref * gref
            lr, bc1, bc2 = 2e-3, 1 - 0.9 ** 3, 1 - 0.98 ** 3
            ww = e["wm"] - (lr / bc1) * mm / (vv.sqrt() / math.sqrt(bc2) + 1e-9)
            e["wm"], e["mm"], e["vm"] = ww, mm, vv
            assert relmax(e["m"], mm) < 1e-5 and relmax(e["v"], vv) < 1e-5
            assert absmax(e["p"], ww) < 2e-6 * max(1.0, float(ww.abs().max()))
            M = e["p"].size(0)
            assert torch.equal(e["lp"], e["p"].to(dtype))
            assert torch.equal(e["lpT"][:, e["r0"]:e["r0"] + M], e["p"].to(dtype).t())
            rest = torch.cat([e["lpT"][:, :e["r0"]], e["lpT"][:, e["r0"] + M:]], dim=1)
            assert bool((rest == 7.0).all())                                   # nothing outside the block was touched
            if e["wg"]:
                assert relmax(e["out"], gref) < 1e-4
            else:
                assert bool(torch.isnan(e["out"]).all())                       # the gradient never went to memory


@pytest.mark.parametrize("rows,V,k,ld_pad", [(1, 7, 3, 0), (5, 3000, 7, 0), (8, 3000, 1, 5), (3, 5000, 6, 0), (2, 16, 16, 0)])
def test_topk_rows_matches_sort(dev, rows, V, k, ld_pad):
    """csrc/select.hip against a stable descending sort (equal values in ascending column order): values, columns and the
    extra column; rows with repeated values; a row stride larger than V; rows longer than the register path (V > 4096)."""
    from mtn_amd import ops
    g = torch.Generator(device="cpu").manual_seed(rows * 1000 + V + k)
    x = torch.randn(rows, V + ld_pad, generator=g)
    x[0, : min(V, 5)] = 1.5                                   # ties at the top of a row
    if rows > 1:
        x[1].round_(decimals=1)                               # many repeated values
    xd = x.to(dev)[:, :V]
    out = ops.topk_rows(xd, k, extra_col=min(3, V - 1)).cpu()
    ref = x[:, :V].double()
    order = torch.argsort(-ref, dim=1, stable=True)[:, :k]
    assert torch.equal(out[:, k:2 * k].long(), order)
    assert torch.equal(out[:, :k].double(), torch.gather(ref, 1, order))
    assert torch.equal(out[:, 2 * k].double(), ref[:, min(3, V - 1)])


@pytest.mark.parametrize("persistent", [True, False])
def test_gemm_k512_many_large_problems(dev, persistent):
    """csrc/gemm_k512.hip (taken for launches of >= 512 output tiles of 128 x 128 with K = 512, plain bias epilogue into a bf16
    output — the memories' K|V projections): several problems in one launch, row counts that are not multiples of the tile, a
    narrow N, row strides larger than the row, against fp32 matmul of the same bf16 operands (one bf16 rounding of the result).
    Both forms: the persistent one (256 resident workgroups walking 64-row units, x tiles two units ahead; N = 520 and 264 make a
    workgroup change its column tile between units, M = 10 leaves most XCDs without a unit of that problem) and the tile-per-
    workgroup one (MTN_K512_PERSIST=0)."""
    import ctypes as C
    import os
    from mtn_amd import lib as L
    lib = L.load()
    os.environ["MTN_K512_PERSIST"] = "1" if persistent else "0"
    L.reload_env()
    g = torch.Generator(device="cpu").manual_seed(5)
    shapes = [(4096, 1024, 512, 1024), (1000, 1024, 512, 1024), (640, 520, 640, 528), (4096, 1024, 512, 1024), (4096, 1024, 512, 1024), (1280, 1024, 512, 1024),
              (10, 1024, 512, 1024), (777, 264, 520, 272), (4090, 512, 512, 512), (65, 8, 512, 16)]
    probs = (L.GemmProblem * len(shapes))()
    keep, refs = [], []
    for i, (M, N, lda, ldc) in enumerate(shapes):
        a = (torch.randn(M, lda, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        w = (torch.randn(N, 512, generator=g) * 0.05).to(torch.bfloat16).to(dev)
        b = torch.randn(N, generator=g).to(dev)
        out = torch.full((M, ldc), 7.0, device=dev, dtype=torch.bfloat16)
        p = probs[i]
        p.A, p.B, p.lda, p.ldb, p.M, p.N, p.K = a.data_ptr(), w.data_ptr(), lda, 512, M, N, 512
        p.bias, p.gate_scale, p.out_lp, p.ldc = b.data_ptr(), 1.0, out.data_ptr(), ldc
        keep += [a, w, b]
        refs.append((out, a[:, :512].float() @ w.float().t() + b, N))
    try:
        lib.mtn_census_begin()
        L.check(lib.mtn_gemm(L.MTN_BF16, len(shapes), probs, L.stream_ptr()))
        torch.cuda.synchronize()
        assert lib.mtn_census_end() == 1
    finally:
        os.environ.pop("MTN_K512_PERSIST", None)
        L.reload_env()
    info = L.CensusLaunch(); lib.mtn_census_info(0, C.byref(info))
    assert lib.mtn_census_variant_name(info.variant).decode() == "gemm_k512_kernel"
    assert info.workgroups == (256 if persistent else sum(((M + 127) // 128) * ((N + 127) // 128) for M, N, _, _ in shapes))
    for out, ref, N in refs:
        assert relmax(out[:, :N].float(), ref) < 1e-2
        assert bool((out[:, N:].float() == 7.0).all())             # nothing written past the problem's columns


@pytest.mark.parametrize("force", ["", "MTN_GEMM_TILE=64", "MTN_GEMM_TILE=32", "MTN_GEMM_TILE=64,MTN_GEMM_FORCE_HALF=1", "MTN_GEMM_TILE=32,MTN_GEMM_FORCE_HALF=1"])
def test_gemm_contraction_major_b_on_lds_dma(dev, force):
    """dX = dY W with W as the forward pass keeps it (b_trans = 1: B stored [K][N]) on gemm_dma_kernel's [k][n]-tile variant
    (transposing LDS reads): every tile / stage size, ragged M, N (multiples of 8) and K (tails inside a stage and across
    stages), grouped, with the gate + residual + both-outputs epilogue the FFN backward uses.  The census must show the
    LDS-DMA kernel, not the register-staged fallback."""
    import ctypes as C
    import os
    from mtn_amd import lib as L, ops
    dtype = torch.bfloat16
    env = dict(kv.split("=") for kv in force.split(",") if kv)
    os.environ.update(env)
    L.reload_env()
    lib = L.load()
    g = torch.Generator().manual_seed(21)
    probs, checks = [], []
    for (M, N, K) in [(640, 512, 2048), (640, 2048, 512), (104, 72, 136), (1000, 1536, 512), (33, 8, 40), (96, 520, 304)]:
        a, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * torch.linspace(0.5, 1.5, N).unsqueeze(1)
        res, gate = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
        A, B = a.to(dev, dtype), b.t().contiguous().to(dev, dtype)            # B: [K][N]
        Res, Gate = res.to(dev), gate.to(dev, dtype)
        of = torch.full((M, N), float("nan"), device=dev)
        ol = torch.empty(M, N, device=dev, dtype=dtype)
        p = _gemm_problem(L, A, B, M, N, K, 0, 1, K, N)
        p.gate, p.gate_scale, p.residual, p.ldr, p.out_f32, p.out_lp, p.ldc = Gate.data_ptr(), 1.25, Res.data_ptr(), N, of.data_ptr(), ol.data_ptr(), N
        probs.append(p)
        v = lp_round(a, dtype).double() @ lp_round(b, dtype).double().t()
        v = torch.where(lp_round(gate, dtype).double() > 0, v * 1.25, torch.zeros_like(v)) + res.double()
        checks.append((of, ol, v, K, (A, B, Res, Gate)))
    try:
        lib.mtn_census_begin()
        for p in probs[:2]:
            ops.gemm(L.MTN_BF16, [p])                # the two FFN-backward shapes of the train step, one launch each
        ops.gemm(L.MTN_BF16, probs[2:])              # the ragged ones grouped
        torch.cuda.synchronize()
        n = lib.mtn_census_end()
        names = []
        for i in range(n):
            info = L.CensusLaunch()
            L.check(lib.mtn_census_info(i, C.byref(info)))
            names.append(lib.mtn_census_variant_name(info.variant).decode())
    finally:
        for k in env:
            del os.environ[k]
        L.reload_env()
    assert n == 3 and all("dma" in nm.lower() for nm in names), names
    for of, ol, v, K, _keep in checks:
        assert relmax(of, v) < 1e-5 * math.sqrt(K)
        assert relmax(ol.float(), v) < 1e-2


@pytest.mark.parametrize("bt", [0, 1])
def test_gemm_128x128_four_stage_kernel(dev, bt):
    """gemm_dma128x_kernel (launches of >= 192 tiles of 128 x 128: the memory-gradient GEMM of the step): both B layouts (row-major,
    and the weight as it lies = b_trans 1), ragged M / N / K (K tails inside a stage, fewer stages than the pipeline is deep),
    grouped, with the residual-accumulate + both-outputs epilogue of the memory gradient."""
    import ctypes as C
    import os
    from mtn_amd import lib as L, ops
    dtype = torch.bfloat16
    os.environ["MTN_GEMM_128X_MIN_TILES"] = "1"
    L.reload_env()
    lib = L.load()
    g = torch.Generator().manual_seed(31 + bt)
    probs, checks = [], []
    for (M, N, K) in [(8064, 512, 1024), (1000, 1536, 520), (200, 136, 264), (130, 8, 256), (257, 384, 3000)]:
        a, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * torch.linspace(0.5, 1.5, N).unsqueeze(1)
        res = torch.randn(M, N, generator=g)
        A = a.to(dev, dtype)
        B = (b.t().contiguous() if bt else b).to(dev, dtype)
        Res = res.to(dev)
        of = torch.full((M, N), float("nan"), device=dev)
        ol = torch.empty(M, N, device=dev, dtype=dtype)
        p = _gemm_problem(L, A, B, M, N, K, 0, bt, K, N if bt else K)
        p.residual, p.ldr, p.out_f32, p.out_lp, p.ldc = Res.data_ptr(), N, of.data_ptr(), ol.data_ptr(), N
        probs.append(p)
        v = lp_round(a, dtype).double() @ lp_round(b, dtype).double().t() + res.double()
        checks.append((of, ol, v, K, (A, B, Res)))
    try:
        lib.mtn_census_begin()
        ops.gemm(L.MTN_BF16, probs)
        torch.cuda.synchronize()
        n = lib.mtn_census_end()
        info = L.CensusLaunch()
        L.check(lib.mtn_census_info(0, C.byref(info)))
        name = lib.mtn_census_variant_name(info.variant).decode()
    finally:
        del os.environ["MTN_GEMM_128X_MIN_TILES"]
        L.reload_env()
    assert n == 1 and name.startswith("gemm_dma128x_kernel"), name
    for of, ol, v, K, _keep in checks:
        assert relmax(of, v) < 1e-5 * math.sqrt(K)
        assert relmax(ol.float(), v) < 1e-2


@pytest.mark.parametrize("bt", [0, 1])
def test_gemm_wave_count_and_epilogue_prefetch_do_not_change_a_bit(dev, bt):
    """Round 4: the 64 x 64 LDS-DMA tile runs on sixteen waves (MTN_GEMM_NW16, default) and the ordinary epilogue's operands are
    loaded ahead of the contraction (MTN_GEMM_EPI_PRE, default).  Neither changes the arithmetic — same contraction order per
    output, same epilogue — so eight waves / epilogue-time loads must give the SAME BITS: bias + dropout + residual (the forward
    second launch), gate + residual (the FFN backward), ragged rows, K = 512 and a K with a tail, both B layouts."""
    import os
    from mtn_amd import lib as L, ops
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(33)
    seed = torch.full((1,), 1234567, device=dev, dtype=torch.int64)
    shapes = [(640, 512, 512), (1000, 512, 2048), (200, 192, 520)]
    keep, probs, outs = [], [], []
    for (M, N, K) in shapes:
        a, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
        A = a.to(dev, dtype)
        B = (b.t().contiguous() if bt else b).to(dev, dtype)
        bias, res, gate = torch.randn(N, generator=g).to(dev), torch.randn(M, N, generator=g).to(dev), torch.randn(M, N, generator=g).to(dev, dtype)
        for mode in ("bias_drop_res", "gate_res"):
            of = torch.full((M, N), float("nan"), device=dev)
            ol = torch.empty(M, N, device=dev, dtype=dtype)
            p = _gemm_problem(L, A, B, M, N, K, 0, bt, K, N if bt else K)
            p.residual, p.ldr, p.out_f32, p.out_lp, p.ldc = res.data_ptr(), N, of.data_ptr(), ol.data_ptr(), N
            if mode == "bias_drop_res":
                p.bias = bias.data_ptr()
                p.drop.p, p.drop.salt, p.drop.seed = 0.1, 17, seed.data_ptr()
            else:
                p.gate, p.gate_scale = gate.data_ptr(), 1.0 / 0.9
            probs.append(p); outs.append((of, ol))
        keep.append((A, B, bias, res, gate))

    def run(env):
        os.environ.update(env)
        L.reload_env()
        try:
            for i in range(0, len(probs), 2):
                ops.gemm(L.MTN_BF16, [probs[i]])
                ops.gemm(L.MTN_BF16, [probs[i + 1]])
            torch.cuda.synchronize()
            return [(of.clone(), ol.clone()) for of, ol in outs]
        finally:
            for k in env:
                del os.environ[k]
            L.reload_env()

    base = run({"MTN_GEMM_TILE": "64"})
    assert all(torch.isfinite(of).all() for of, _ in base)
    for env in ({"MTN_GEMM_TILE": "64", "MTN_GEMM_NW16": "0"}, {"MTN_GEMM_TILE": "64", "MTN_GEMM_EPI_PRE": "0"},
                {"MTN_GEMM_TILE": "64", "MTN_GEMM_NW16": "0", "MTN_GEMM_EPI_PRE": "0"}, {"MTN_GEMM_TILE": "32"}, {"MTN_GEMM_TILE": "32", "MTN_GEMM_EPI_PRE": "0"}):
        got = run(env)
        for (rf, rl), (gf, gl) in zip(base, got):
            assert torch.equal(rf, gf) and torch.equal(rl, gl), env
