"""Independence of the in-kernel dropout streams (csrc/common.h: keep(idx) = top 24 bits of a one-mixer counter hash of
(device seed, site salt, element index) against p * 2^24).  The reference draws every nn.Dropout site and every step
independently (mtn.py:123, 230, 246, 277); here masks of different sites / steps come from ONE mixer keyed differently, so this
checks what that construction has to deliver: masks of two sites at EQUAL indices, of one site at consecutive seeds (the step's
seed advance of EncoderDecoder.advance_dropout_seed), and of one site at neighbouring indices are uncorrelated, and the keep
bits of 8-element neighbourhoods follow the binomial law."""
import ctypes as C

import pytest
import torch

from tests.test_model_gpu import dev  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

N = 1 << 24
SEED_STEP = 0x9E3779B97F4A7C15 & 0x7FFFFFFFFFFFFFFF      # EncoderDecoder.advance_dropout_seed


def _mask(dev, p, salt, seed_value, n=N):
    """keep bits (uint8) of elements 0..n-1 of site `salt` under device seed `seed_value`, from the library's dropout kernel."""
    from mtn_amd import lib as L
    from mtn_amd import ops
    src = torch.ones(n, device=dev, dtype=torch.float32)
    dst = torch.empty(n, device=dev, dtype=torch.float32)
    seed = torch.full((1,), seed_value, device=dev, dtype=torch.int64)
    L.check(L.load().mtn_dropout_bwd_to_lp(L.MTN_F32, n, src.data_ptr(), ops._drop(p, salt, seed), dst.data_ptr(), L.stream_ptr()))
    torch.cuda.synchronize()
    keep = dst > 0
    assert torch.all(dst[keep] == 1.0 / (1.0 - p))
    return keep


def _corr(a, b):
    a = a.double() - a.double().mean()
    b = b.double() - b.double().mean()
    return float((a * b).mean() / (a.std(unbiased=False) * b.std(unbiased=False)))


@pytest.mark.parametrize("p", [0.1, 0.5])
def test_masks_of_sites_and_steps_are_uncorrelated(dev, p):
    seed0 = 0x1234567812345678
    base = _mask(dev, p, 17, seed0)
    rate = float(base.double().mean())
    assert abs(rate - (1.0 - p)) < 5e-4, rate                       # sigma = sqrt(p (1 - p) / 2^24) <= 1.3e-4
    bound = 1.2e-3                                                   # |rho| of independent masks: sigma = 2^-12 = 2.4e-4
    # other sites, equal indices: the salts of one sublayer (4 s + {0, 1, 2}), of neighbouring sublayers, of the embedding sites
    for salt in (16, 18, 19, 21, 4 * 65 + 1, 900000, 910001):
        rho = _corr(base, _mask(dev, p, salt, seed0))
        assert abs(rho) < bound, (salt, rho)
    # the same site at the next steps' seeds, and at seeds that differ in one low / one high bit
    for k, seed in (("step+1", (seed0 + SEED_STEP) & 0x7FFFFFFFFFFFFFFF), ("step+2", (seed0 + 2 * SEED_STEP) & 0x7FFFFFFFFFFFFFFF),
                    ("bit0", seed0 ^ 1), ("bit40", seed0 ^ (1 << 40))):
        rho = _corr(base, _mask(dev, p, 17, seed))
        assert abs(rho) < bound, (k, rho)
    # neighbouring elements of one mask (lags 1, 2, 64, 512: along a row, across rows of d_model = 512)
    for lag in (1, 2, 64, 512):
        rho = _corr(base[:-lag], base[lag:])
        assert abs(rho) < bound, (lag, rho)


@pytest.mark.parametrize("p", [0.1, 0.25])
def test_eight_element_neighbourhoods_follow_the_binomial_law(dev, p):
    """chi-square of the 256 keep patterns of aligned 8-element groups against p^(dropped) (1-p)^(kept): 255 degrees of freedom
    (mean 255, sigma 22.6): the bar is mean + 5 sigma."""
    keep = _mask(dev, p, 33, 0x0BADC0DE12345)
    w = (2 ** torch.arange(8, device=keep.device)).to(torch.int64)
    pat = (keep.view(-1, 8).to(torch.int64) * w).sum(1)
    obs = torch.bincount(pat, minlength=256).double().cpu()
    kept = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.float64)
    exp = (N // 8) * (1.0 - p) ** kept * p ** (8 - kept)
    chi2 = float(((obs - exp) ** 2 / exp).sum())
    assert chi2 < 255 + 5 * 22.6, chi2


def test_seed_advanced_inside_a_graph_is_seen_by_the_next_kernel(dev):
    """The kernels read the dropout seed with a SCALAR load (common.h drop_init: a vector load made its consumer wait for every load in
    flight).  The seed is advanced on the device once per step by an ordinary kernel (EncoderDecoder.advance_dropout_seed), inside
    the captured step: every replay's dropout kernels must see THAT replay's seed — never a value a CU's scalar cache kept from the
    replay before.  300 replays of {seed += step; mask = dropout(ones)} against masks made eagerly from the same seed values."""
    from mtn_amd import lib as L
    from mtn_amd import ops
    n, p, salt = 1 << 16, 0.3, 77
    seed0 = 0x0123456789ABCDEF
    src = torch.ones(n, device=dev, dtype=torch.float32)
    dst = torch.empty(n, device=dev, dtype=torch.float32)
    seed = torch.full((1,), seed0, device=dev, dtype=torch.int64)
    drop = ops._drop(p, salt, seed)
    lib = L.load()

    def body():
        seed.add_(SEED_STEP)
        seed.bitwise_and_(0x7FFFFFFFFFFFFFFF)
        # two launches per replay: the second must see the same seed as the first (it runs on whatever CUs are free)
        L.check(lib.mtn_dropout_bwd_to_lp(L.MTN_F32, n, src.data_ptr(), drop, dst.data_ptr(), L.stream_ptr()))
        L.check(lib.mtn_dropout_bwd_to_lp(L.MTN_F32, n, src.data_ptr(), drop, dst2.data_ptr(), L.stream_ptr()))

    dst2 = torch.empty_like(dst)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body()                                              # warm-up outside capture
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    seed.fill_(seed0)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    seed.fill_(seed0)
    torch.cuda.synchronize()
    value = seed0
    for it in range(300):
        g.replay()
        torch.cuda.synchronize()
        value = (value + SEED_STEP) & 0x7FFFFFFFFFFFFFFF
        assert int(seed.item()) == value
        if it % 25 == 0 or it > 290:
            want = _mask(dev, p, salt, value, n)
            assert torch.equal(dst > 0, want), it
        assert torch.equal(dst, dst2), it
