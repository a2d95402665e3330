"""Pin the CPU oracle (oracle/mtn_oracle.py) to outputs of the reference itself.

tests/golden/*.npz were produced by oracle/make_golden.py, which imports /root/reference/mtn.py
on CPU.  Weights/inputs are regenerated from oracle/fixtures.py formulas; only outputs are stored.
The reference has no tests or golden vectors of its own - these are the only pins.
"""
import os

import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from oracle.mtn_oracle import beam_search, label_smoothing_kl, noam_rate

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 2e-5      # fp32 CPU vs fp32 CPU, different op order only


def load(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False))


def close(a, b, tol=TOL):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    err = np.abs(a.astype(np.float64) - b.astype(np.float64)).max()
    scale = max(1.0, np.abs(b).max())
    assert err <= tol * scale, f"max err {err} (scale {scale})"


@pytest.mark.parametrize("name", list(fx.GOLDEN_CONFIGS))
def test_forward_matches_reference(name):
    c = fx.GOLDEN_CONFIGS[name]
    g = load(name)
    m, cfg = fx.oracle_from_config(c)
    b = fx.oracle_batch(fx.det_batch(c["vocab"], c["B"], c["Q"], c["H"], c["C"], c["T"], c["frames"], c["ft_sizes"], seed=1))
    with torch.no_grad():
        q, v, cp, hs, ae = m.encode(b.query, b.query_mask, b.his, b.his_mask, b.cap, b.cap_mask, b.fts, b.fts_mask)
        close(q, g["enc.q"]); close(cp, g["enc.cap"]); close(hs, g["enc.his"])
        for i, x in enumerate(v):
            close(x, g[f"enc.vid.{i}"])
        if ae is not None:
            for i, x in enumerate(ae):
                close(x, g[f"enc.ae.{i}"])
        else:
            assert "enc.ae.0" not in g
        m.taps = {}
        out, ae_out = m.forward(b)
        n_sub = 5 + 4 * len(c["ft_sizes"])
        for k in range(n_sub):
            close(m.taps[f"decoder.layers.0.sublayer.{k}"], g[f"layer0.sublayer.{k}"])
        close(out, g["out"])
        for i, a in enumerate(ae_out):
            close(a, g[f"ae_out.{i}"])
        close(m.generator(out), g["logp"], 5e-5)


@pytest.mark.parametrize("name", list(fx.GOLDEN_CONFIGS))
def test_loss_and_grads_match_reference(name):
    c = fx.GOLDEN_CONFIGS[name]
    g = load(name)
    m, cfg = fx.oracle_from_config(c, requires_grad=True)
    b = fx.oracle_batch(fx.det_batch(c["vocab"], c["B"], c["Q"], c["H"], c["C"], c["T"], c["frames"], c["ft_sizes"], seed=1))
    out, ae_out = m.forward(b)
    loss = m.loss(b, out, ae_out)
    close(loss, g["loss"], 1e-5)
    loss.backward()
    norms = dict(zip([str(s) for s in g["grad_names"]], g["grad_norms"]))
    checked = 0
    for k, p in m.p.items():
        if k not in norms:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        gn = float(p.grad.double().norm())
        assert abs(gn - norms[k]) <= 1e-4 * max(1.0, norms[k]) + 1e-6, (k, gn, norms[k])
        if "grad." + k in g:
            close(p.grad, g["grad." + k], 1e-4)
        else:
            close(p.grad.reshape(-1)[:256], g["gradhead." + k], 1e-4)
        checked += 1
    assert checked == len(norms)


def test_two_adam_noam_steps_match_reference():
    """train.py:190 optimiser (Adam b=(0.9,0.98) eps=1e-9, Noam lr) driven by the oracle's grads."""
    name = "cfg1_query"
    c = fx.GOLDEN_CONFIGS[name]
    g = load(name)
    m, cfg = fx.oracle_from_config(c, requires_grad=True)
    b = fx.oracle_batch(fx.det_batch(c["vocab"], c["B"], c["Q"], c["H"], c["C"], c["T"], c["frames"], c["ft_sizes"], seed=1))
    params = list(m.p.values())
    opt = torch.optim.Adam(params, lr=0.0, betas=(0.9, 0.98), eps=1e-9)
    losses = []
    for step in (1, 2):
        out, ae_out = m.forward(b)
        loss = m.loss(b, out, ae_out)
        opt.zero_grad()
        loss.backward()
        for grp in opt.param_groups:
            grp["lr"] = noam_rate(step, c["d_model"], 10)
        opt.step()
        losses.append(float(loss.detach()) * float(b.ntokens))
    np.testing.assert_allclose(losses, g["step_losses"], rtol=2e-5)
    for k in g:
        if k.startswith("after2."):
            close(m.p[k[len("after2."):]], g[k], 1e-4)
        elif k.startswith("after2head."):
            close(m.p[k[len("after2head."):]].reshape(-1)[:256], g[k], 1e-4)


@pytest.mark.parametrize("name", ["cfg1_query", "small_shared"])
def test_beam_search_matches_reference(name):
    c = fx.GOLDEN_CONFIGS[name]
    g = load(name)
    m, cfg = fx.oracle_from_config(c)
    b = fx.oracle_batch(fx.det_batch(c["vocab"], 1, c["Q"], c["H"], c["C"], c["T"], c["frames"], c["ft_sizes"], seed=2, ragged=False))
    with torch.no_grad():
        nbest, best = beam_search(m, b, 8, fx.SOS, fx.UNK, fx.EOS)
    assert len(nbest) == int(g["beam.n"])
    for i, (toks, score) in enumerate(nbest):
        assert list(toks) == list(g[f"beam.tokens.{i}"]), i
        assert abs(score - float(g[f"beam.score.{i}"])) < 1e-3
    assert abs(best - float(g["beam.best"])) < 1e-3


def test_label_smoothing_index0_quirk():
    """label_smoothing.py:29: a single padded row at flat index 0 is not zeroed."""
    logp = torch.log_softmax(torch.randn(3, 7, generator=torch.Generator().manual_seed(0)), -1)
    t_pad0 = torch.tensor([1, 4, 5])
    t_pad1 = torch.tensor([4, 1, 5])
    a = label_smoothing_kl(logp, t_pad0, 1, 0.1)
    bb = label_smoothing_kl(logp, t_pad1, 1, 0.1)
    # row 0 with target==pad keeps its smoothing mass (quirk); row 1 with target==pad is zeroed
    assert abs(float(a) - float(label_smoothing_kl(logp[1:], t_pad0[1:], 1, 0.1))) > 1e-3
    rows = torch.tensor([0, 2])
    assert abs(float(bb) - float(label_smoothing_kl(logp[rows], t_pad1[rows], 1, 0.1))) < 1e-5
