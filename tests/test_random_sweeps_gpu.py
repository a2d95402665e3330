"""Seeded random sweeps over shapes the fixed parametrisations do not list: grouped GEMMs in every operand layout (tile edges,
group offsets that shift the XCD-aware tile mapping, contraction tails), attention with ragged memories / masks, LayerNorm row
counts — each against the oracle's leaf functions."""
import random

import pytest
import torch

from tests.util import lp_round, relmax

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("seed", range(12))
def test_random_gemm_groups(dev, seed):
    from mtn_amd import lib as L, ops
    rnd = random.Random(seed)
    dtype = torch.bfloat16 if seed % 3 else torch.float32
    at, bt = rnd.choice([(0, 0), (0, 0), (1, 1), (0, 1), (1, 0)])
    unit = 8 if dtype == torch.bfloat16 else 4
    g = torch.Generator().manual_seed(seed)
    probs, checks, keep = [], [], []
    for _ in range(rnd.randint(1, 7)):
        M = rnd.choice([unit * rnd.randint(1, 40), 64 * rnd.randint(1, 12), 640])
        N = rnd.choice([unit * rnd.randint(1, 40), 64 * rnd.randint(1, 10), 512])
        K = rnd.choice([unit * rnd.randint(1, 90), 512, 640, 1024])
        a, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
        A = (a.t().contiguous() if at else a).to(dev, dtype)
        B = (b.t().contiguous() if bt else b).to(dev, dtype)
        bias = torch.randn(N, generator=g)
        out = torch.full((M, N), float("nan"), device=dev)
        bd = bias.to(dev)
        p = L.GemmProblem()
        p.A, p.B, p.lda, p.ldb, p.M, p.N, p.K, p.a_trans, p.b_trans, p.gate_scale = A.data_ptr(), B.data_ptr(), (M if at else K), (N if bt else K), M, N, K, at, bt, 1.0
        p.bias, p.relu, p.out_f32, p.ldc = bd.data_ptr(), seed & 1, out.data_ptr(), N
        probs.append(p)
        ref = lp_round(a, dtype).double() @ lp_round(b, dtype).double().t() + bias.double()
        checks.append((out, torch.relu(ref) if seed & 1 else ref))
        keep += [A, B, bd]
    ops.gemm(L.dtype_code(dtype), probs)
    torch.cuda.synchronize()
    for out, ref in checks:
        assert relmax(out, ref) < 2e-4


@pytest.mark.parametrize("seed", range(10))
def test_random_attention(dev, seed):
    from mtn_amd import ops
    from tests.test_kernels_gpu import _attn_ref
    rnd = random.Random(100 + seed)
    dtype = torch.bfloat16 if seed % 2 else torch.float32
    B, h = rnd.randint(1, 4), rnd.choice([1, 2, 4, 8])
    dk = rnd.choice([16, 32, 64])
    a, m = rnd.randint(1, 32), rnd.randint(1, 300)
    d = h * dk
    g = torch.Generator().manual_seed(seed)
    q, k, v = (torch.randn(B, n, d, generator=g) for n in (a, m, m))
    lens = torch.randint(1, m + 1, (B,), generator=g)
    mask = (torch.arange(m).unsqueeze(0) < lens.unsqueeze(1)).unsqueeze(1)
    if seed % 3 == 0:
        mask[0] = False                                      # a fully masked row: uniform attention (masked_fill -1e9)
    gy = torch.randn(B, a, d, generator=g)
    qr, kr, vr = (lp_round(t, dtype).requires_grad_() for t in (q, k, v))
    o_ref, _ = _attn_ref(qr, kr, vr, mask, h)
    o_ref.backward(gy)
    qd, kd, vd = (t.to(dev, dtype) for t in (q, k, v))
    o, lse = ops.attention(qd, kd, vd, mask.to(dev), h)
    dq, dk_, dv = ops.attention_bwd(qd, kd, vd, o, lse, gy.to(dev, dtype), mask.to(dev), h)
    torch.cuda.synchronize()
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-4
    assert relmax(o.float(), o_ref) < tol
    assert relmax(dq.float(), qr.grad) < 3 * tol and relmax(dk_.float(), kr.grad) < 3 * tol and relmax(dv.float(), vr.grad) < 3 * tol


@pytest.mark.parametrize("seed", range(6))
def test_random_layernorm(dev, seed):
    from mtn_amd import ops
    from oracle.mtn_oracle import layer_norm as ref_ln
    rnd = random.Random(200 + seed)
    rows, d = rnd.randint(1, 700), rnd.choice([64, 128, 512, 768, 1024])
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, d, generator=g) * rnd.uniform(0.1, 5) + rnd.uniform(-3, 3)
    a2, b2 = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    gy = torch.randn(rows, d, generator=g)
    xr, ar, br = x.clone().requires_grad_(), a2.clone().requires_grad_(), b2.clone().requires_grad_()
    ref_ln(xr, ar, br, 1e-6).backward(gy)
    xd, ad, bd = x.to(dev).requires_grad_(), a2.to(dev).requires_grad_(), b2.to(dev).requires_grad_()
    y, _ = ops.layer_norm(xd, ad, bd, 1e-6, torch.bfloat16)
    y.backward(gy.to(dev))
    torch.cuda.synchronize()
    assert relmax(y, ref_ln(x, a2, b2, 1e-6)) < 1e-4
    assert relmax(xd.grad, xr.grad) < 2e-4 and relmax(ad.grad, ar.grad) < 2e-4 and relmax(bd.grad, br.grad) < 2e-4
