"""Decode path (BASELINE configs[4]) on the GPU: beam search against the n-best lists the REFERENCE produced
(tests/golden/*.npz, beam.* keys, made by oracle/make_golden.py) and against the CPU oracle; greedy against the oracle.
fp32 mode must reproduce the reference's token sequences exactly and its scores to 1e-3; bf16 mode is held to the scores
(1e-2 relative on the best hypothesis) — near-ties between candidates may legitimately reorder under bf16 rounding."""
import os

import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from oracle import mtn_oracle as orc
from tests.test_model_gpu import build_model, dev_batch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


def one_dialogue(c, seed=2):
    return fx.det_batch(c["vocab"], 1, c["Q"], c["H"], c["C"], c["T"], c["frames"], c["ft_sizes"], seed=seed, ragged=False)


@pytest.mark.parametrize("kv_cache", [False, True], ids=["full-prefix", "kv-cache"])
@pytest.mark.parametrize("use_graph", [True, False])
@pytest.mark.parametrize("name", list(fx.GOLDEN_CONFIGS))
def test_beam_search_matches_reference_nbest(dev, name, use_graph, kv_cache):
    """The reference's own n-best lists (beam 5): with the full-prefix pass and with the prefix K/V cache (one new position
    per token, the cache rows following their parent hypotheses)."""
    from mtn_amd.decode import beam_search_decode
    c = fx.GOLDEN_CONFIGS[name]
    g = dict(np.load(os.path.join(GOLD, name + ".npz")))
    model = build_model(c, torch.float32, dev).eval()
    b = dev_batch(one_dialogue(c), dev)
    nbest, best = beam_search_decode(model, b, 8, fx.SOS, fx.UNK, fx.EOS, fx.PAD, use_graph=use_graph, kv_cache=kv_cache)
    assert len(nbest) == int(g["beam.n"])
    for i, (toks, score) in enumerate(nbest):
        assert list(toks) == list(g[f"beam.tokens.{i}"]), i
        assert abs(score - float(g[f"beam.score.{i}"])) < 1e-3
    assert abs(best - float(g["beam.best"])) < 1e-3


@pytest.mark.parametrize("name", ["cfg1_query", "small_diffall"])
def test_beam_search_bf16_scores_and_other_widths(dev, name):
    """bf16 compute: best score within 1e-2 of the oracle's; beam 4 / nbest 3 / min_len 2 / penalty 0.5 against the oracle."""
    from mtn_amd.decode import beam_search_decode
    c = fx.GOLDEN_CONFIGS[name]
    raw = one_dialogue(c, seed=7)
    m_or, _ = fx.oracle_from_config(c)
    with torch.no_grad():
        ref_n, ref_best = orc.beam_search(m_or, fx.oracle_batch(raw), 10, fx.SOS, fx.UNK, fx.EOS, beam=4, penalty=0.5, nbest=3, min_len=2)
    b = dev_batch(raw, dev)
    model = build_model(c, torch.float32, dev).eval()
    got_n, got_best = beam_search_decode(model, b, 10, fx.SOS, fx.UNK, fx.EOS, fx.PAD, beam=4, penalty=0.5, nbest=3, min_len=2)
    assert [list(t) for t, _ in got_n] == [list(t) for t, _ in ref_n]
    assert max(abs(a[1] - r[1]) for a, r in zip(got_n, ref_n)) < 1e-3 and abs(got_best - ref_best) < 1e-3
    model16 = build_model(c, torch.bfloat16, dev).eval()
    n16, best16 = beam_search_decode(model16, b, 10, fx.SOS, fx.UNK, fx.EOS, fx.PAD, beam=4, penalty=0.5, nbest=3, min_len=2)
    assert len(n16) == len(ref_n)
    assert abs(best16 - ref_best) < 1e-2 * max(1.0, abs(ref_best))


@pytest.mark.parametrize("name", ["cfg1_caption", "small_shared"])
def test_greedy_decode_matches_oracle(dev, name):
    from mtn_amd.decode import greedy_decode
    c = fx.GOLDEN_CONFIGS[name]
    raw = one_dialogue(c, seed=3)
    m_or, _ = fx.oracle_from_config(c)
    with torch.no_grad():
        want = orc.greedy_search(m_or, fx.oracle_batch(raw), 12, fx.SOS)
    model = build_model(c, torch.float32, dev).eval()
    ys = greedy_decode(model, dev_batch(raw, dev), 12, fx.SOS, fx.PAD)
    assert ys.shape == (1, 12) and ys[0].tolist() == want


def test_decode_session_logprobs_match_full_decode(dev):
    """The fixed-shape, auto-encoder-cached target pass gives the log-probabilities of model.decode + generator on the
    same prefixes (several hypotheses at once, prefix shorter than the session's max_len)."""
    from mtn_amd.decode import DecodeSession
    from mtn_amd import subsequent_mask
    c = fx.GOLDEN_CONFIGS["cfg1_query"]
    model = build_model(c, torch.bfloat16, dev).eval()
    b = dev_batch(one_dialogue(c, seed=9), dev)
    sess = DecodeSession(model, b, max_len=9, width=3, pad=fx.PAD)
    prefixes = [[fx.SOS, 5, 9, 11], [fx.SOS, 7, 7, 30], [fx.SOS, 40, 2, 6]]
    got = sess.step(prefixes).clone()
    got2 = sess.step(prefixes).clone()            # graph replay
    assert torch.equal(got, got2)
    with torch.no_grad():
        q, v, cp, hs, ae = model.encode(b.query, b.query_mask, b.his, b.his_mask, b.cap, b.cap_mask, b.fts, b.fts_mask)
        for i, p in enumerate(prefixes):
            st = torch.tensor([p], device=dev)
            x, _ = model.decode(v, hs, cp, q, b.fts_mask, b.his_mask, b.cap_mask, b.query_mask, st, subsequent_mask(len(p), device=dev), ae)
            want = model.generator(x[:, -1]).float()[0]
            assert (got[i] - want).abs().max() < 2e-2 * want.abs().max()


def test_beam_search_many_dialogues_equals_one_by_one(dev):
    """D ragged dialogues decoded side by side (D x beam hypotheses per pass) give, for every dialogue, the n-best list of the
    single-dialogue search (and, for the first one, the reference's golden list)."""
    from mtn_amd.decode import beam_search_decode, beam_search_decode_many
    name = "cfg1_query"
    c = fx.GOLDEN_CONFIGS[name]
    model = build_model(c, torch.float32, dev).eval()
    raws = [one_dialogue(c, seed=2), one_dialogue(c, seed=5), one_dialogue(c, seed=8)]
    singles = [beam_search_decode(model, dev_batch(r, dev), 8, fx.SOS, fx.UNK, fx.EOS, fx.PAD) for r in raws]
    g = dict(np.load(os.path.join(GOLD, name + ".npz")))
    assert [list(t) for t, _ in singles[0][0]] == [list(g[f"beam.tokens.{i}"]) for i in range(int(g["beam.n"]))]
    # make the batch ragged: cut the 2nd dialogue's history/caption and pad
    raws[1]["his"][:, 11:] = fx.PAD
    raws[1]["cap"][:, 7:] = fx.PAD
    singles[1] = beam_search_decode(model, dev_batch(raws[1], dev), 8, fx.SOS, fx.UNK, fx.EOS, fx.PAD)
    cat = {k: (np.concatenate([r[k] for r in raws], 0) if k != "fts" else [np.concatenate([r["fts"][i] for r in raws], 0) for i in range(len(raws[0]["fts"]))])
           for k in raws[0]}
    many = beam_search_decode_many(model, dev_batch(cat, dev), 8, fx.SOS, fx.UNK, fx.EOS, fx.PAD)
    assert len(many) == 3
    for (nb1, best1), (nbm, bestm) in zip(singles, many):
        assert [list(t) for t, _ in nbm] == [list(t) for t, _ in nb1]
        assert max(abs(a[1] - b[1]) for a, b in zip(nbm, nb1)) < 1e-3 and abs(best1 - bestm) < 1e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_kv_cache_decode_equals_full_prefix_decode_on_a_long_search(dev, dtype):
    """max_len 48 (beyond KV_CACHE_FROM: the cache is the default there), d_model 512 so that the fused forward kernel serves
    both passes: beam 4 over 3 dialogues side by side.  fp32 mode: identical n-best token lists and scores within 1e-4; bf16:
    best score within 1e-2 and greedy token chains equal."""
    from mtn_amd.decode import beam_search_decode_many, greedy_decode
    c = dict(vocab=96, N=2, d_model=512, d_ff=1024, h=8, ft_sizes=[64, 32], B=3, Q=9, H=30, C=14, T=8, frames=[11, 7],
             diff_encoder=True, diff_embed=False, diff_gen=False, auto_encoder_ft="query")
    model = build_model(c, dtype, dev).eval()
    raw = fx.det_batch(c["vocab"], c["B"], c["Q"], c["H"], c["C"], c["T"], c["frames"], c["ft_sizes"], seed=2, ragged=True)
    b = dev_batch(raw, dev)
    L = 48
    full = beam_search_decode_many(model, b, L, fx.SOS, fx.UNK, fx.EOS, fx.PAD, beam=4, nbest=4, kv_cache=False)
    cached = beam_search_decode_many(model, b, L, fx.SOS, fx.UNK, fx.EOS, fx.PAD, beam=4, nbest=4)          # default: cache on
    for (nf, bf), (nc, bc) in zip(full, cached):
        if dtype == torch.float32:
            assert [t for t, _ in nf] == [t for t, _ in nc]
            assert all(abs(sf - sc) < 1e-4 * max(1.0, abs(sf)) for (_, sf), (_, sc) in zip(nf, nc))
        else:
            assert abs(bf - bc) < 1e-2 * max(1.0, abs(bf))
    one = dev_batch({k: (v[:1] if k != "fts" else [f[:1] for f in v]) for k, v in raw.items()}, dev)
    g_full = greedy_decode(model, one, L, fx.SOS, fx.PAD, kv_cache=False)
    g_cache = greedy_decode(model, one, L, fx.SOS, fx.PAD)
    if dtype == torch.float32:
        assert torch.equal(g_full, g_cache)
    else:
        assert float((g_full == g_cache).float().mean()) > 0.5      # bf16 near-ties may fork the chain late


@pytest.mark.parametrize("beam", [14, 16])
def test_wide_beams_keep_working(dev, beam):
    """Beams wider than the device-side selection holds (csrc/select.hip keeps 16 entries per row: beam + 3 <= 16) take the
    torch.topk branch: same n-best lists as the oracle's search (data_utils.py:188-242 with beam > 13)."""
    from mtn_amd.decode import beam_search_decode
    c = fx.GOLDEN_CONFIGS["cfg1_query"]
    raw = one_dialogue(c, seed=4)
    m_or, _ = fx.oracle_from_config(c)
    with torch.no_grad():
        ref_n, ref_best = orc.beam_search(m_or, fx.oracle_batch(raw), 6, fx.SOS, fx.UNK, fx.EOS, beam=beam, nbest=5)
    model = build_model(c, torch.float32, dev).eval()
    got_n, got_best = beam_search_decode(model, dev_batch(raw, dev), 6, fx.SOS, fx.UNK, fx.EOS, fx.PAD, beam=beam, nbest=5)
    assert [list(t) for t, _ in got_n] == [list(t) for t, _ in ref_n]
    assert max(abs(a[1] - r[1]) for a, r in zip(got_n, ref_n)) < 1e-3 and abs(got_best - ref_best) < 1e-3


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("kind,dtype", [("launch", torch.float32), ("launch", torch.bfloat16), ("persistent", torch.bfloat16)],
                         ids=["launch-fp32", "launch-bf16", "persistent-bf16"])
def test_cfg5_full_depth_first_decode_steps_match_oracle(dev, kind, dtype):
    """BASELINE configs[4] at its real depth: the 6-layer d_model=512 / 8-head model (cfg2's widths, |V| = 3000, Q/H/C = 20/128/40,
    32 + 32 frames), beam 4.  The log-probabilities of decode steps 1, 2, 3 and 10 (generate.py -> data_utils.py:202-208: full
    decode of every live prefix, generator on the last position) against the CPU ORACLE, for a beam-like walk over the model's own
    best extensions — on the launch-per-sublayer pass (DecodeSession) AND on the persistent step kernel the decode bench line is
    measured on (MegaDecodeSession, csrc/decode.hip: step 10 reads prefix-cache rows written by the nine launches before it).
    fp32 mode 1e-3, bf16 mode 1e-2 (north_star's bars), relative to the row's largest |log p|."""
    from mtn_amd.decode import DecodeSession, MegaDecodeSession
    from mtn_amd.synthetic import CONFIGS
    k = CONFIGS["cfg2"]
    c = dict(vocab=k["vocab"], N=k["N"], d_model=k["d_model"], d_ff=k["d_ff"], h=k["h"], ft_sizes=list(k["ft_sizes"]), B=1, Q=k["Q"], H=k["H"],
             C=k["C"], T=k["T"], frames=list(k["frames"]), diff_encoder=True, diff_embed=False, diff_gen=False, auto_encoder_ft="query")
    raw = one_dialogue(c, seed=5)
    m_or, _ = fx.oracle_from_config(c)
    ob = fx.oracle_batch(raw)
    beam = 4
    model = build_model(c, dtype, dev).eval()
    b = dev_batch(raw, dev)
    if kind == "persistent":
        assert MegaDecodeSession.supported(model, b, 20, beam)
        sess = MegaDecodeSession(model, b, max_len=20, width=beam, pad=fx.PAD)
    else:
        sess = DecodeSession(model, b, max_len=20, width=beam, pad=fx.PAD)
    tol = 1e-3 if dtype == torch.float32 else 1e-2
    checked = (0, 1, 2, 9)
    prefixes = [[fx.SOS]]
    worst = 0.0
    with torch.no_grad():
        q, v, cp, hs, ae = m_or.encode(ob.query, ob.query_mask, ob.his, ob.his_mask, ob.cap, ob.cap_mask, ob.fts, ob.fts_mask)
        for step in range(10):
            got = sess.step(prefixes).float().cpu()
            if kind == "persistent":
                sess.check()
            if step in checked:
                for i, p in enumerate(prefixes):
                    st = torch.tensor([p], dtype=ob.query.dtype)
                    x, _ = m_or.decode(v, hs, cp, q, ob.fts_mask, ob.his_mask, ob.cap_mask, ob.query_mask, st, orc.subsequent_mask(len(p)), ae)
                    want = m_or.generator(x[:, -1])[0]
                    err = float((got[i] - want).abs().max() / want.abs().max())
                    worst = max(worst, err)
                    assert err < tol, (kind, step, i, err)
            # the next step's prefixes: every live hypothesis extended by its best tokens (<unk> / <eos> never extend one) until the beam is full
            nxt = []
            for i, p in enumerate(prefixes):
                top = [int(t) for t in torch.argsort(got[i], descending=True)[:beam + 2] if int(t) not in (fx.UNK, fx.EOS)]
                nxt += [p + [t] for t in top[:max(1, beam // len(prefixes))]]
            prefixes = nxt[:beam]
    print(f"cfg5 {kind} {dtype}: worst relative log-probability error against the oracle over steps 1-3 and 10: {worst:.2e}")


def _mega_vs_launch_pass(dev, model, b, width, max_len, steps):
    """log-probabilities of `steps` decode steps of a greedy-ish walk on the persistent launch (csrc/decode.hip) and on the
    launch-per-sublayer pass, same bf16 weights, same prefixes"""
    from mtn_amd.decode import DecodeSession, MegaDecodeSession
    assert MegaDecodeSession.supported(model, b, max_len, width)
    mega = MegaDecodeSession(model, b, max_len, width, pad=fx.PAD)
    ref = DecodeSession(model, b, max_len, width, pad=fx.PAD)
    prefixes = [[fx.SOS]]
    worst = 0.0
    for l in range(steps):
        got = mega.step(prefixes).clone()
        want = ref.step(prefixes).clone()
        mega.check()
        err = float((got - want).abs().max() / want.abs().max())
        worst = max(worst, err)
        assert err < 2e-2, (l, err)
        assert torch.equal(got.argmax(-1), want.argmax(-1)) or float((want.max(-1).values - want.gather(1, got.argmax(-1, keepdim=True)).squeeze(1)).abs().max()) < 2e-2
        # next prefixes: every live hypothesis extended by its best and (while the beam fills) its second-best token, as a beam step would
        nxt = []
        top2 = want.topk(2, dim=-1).indices.tolist()
        for p, t2 in zip(prefixes, top2):
            for t in t2:
                if len(nxt) < width and t not in (fx.EOS, fx.UNK):
                    nxt.append(p + [int(t)])
        prefixes = nxt[:width] if nxt else [p + [5] for p in prefixes]
    return worst


@pytest.mark.parametrize("name,width", [("cfg1_query", 5), ("cfg1_caption", 3), ("wide_n1", 1), ("wide_n1", 8), ("wide_n1", 16), ("cfg1_query", 13)])
def test_persistent_decode_step_matches_launch_pass_small(dev, name, width):
    """csrc/decode.hip against the launch-per-sublayer pass on the golden configurations (d_model 128, 4 heads of 32: the two-tile /
    k-split plans of the small-M Linear, caption-mode order of the cross-attentions; d_model 512 with one hypothesis = greedy, with
    eight, and with sixteen = the launch's maximum: rows 9-16 ride in the second half of the 16-row MFMA tiles; 13 = a ragged second half):
    log-probabilities of 7 steps of a beam-like walk within 2e-2 of the row's largest magnitude (both paths bf16)."""
    c = fx.GOLDEN_CONFIGS[name]
    model = build_model(c, torch.bfloat16, dev).eval()
    b = dev_batch(one_dialogue(c, seed=9), dev)
    worst = _mega_vs_launch_pass(dev, model, b, width, 9, 7)
    print(f"{name} width {width}: worst relative log-probability difference {worst:.2e}")


def test_persistent_decode_step_matches_launch_pass_cfg5(dev):
    """The benchmark's decode shape (cfg5: d_model 512, 6 layers, 8 heads, beam 4, H = 128, frames 32): 10 steps."""
    from mtn_amd import make_model
    from mtn_amd.synthetic import CONFIGS, synthetic_batch
    cfg = dict(CONFIGS["cfg2"])
    torch.manual_seed(0)
    model = make_model(cfg["vocab"], cfg["vocab"], N=cfg["N"], d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"], dropout=0.1,
                       ft_sizes=cfg["ft_sizes"], diff_encoder=True, auto_encoder_ft="query", compute_dtype=torch.bfloat16).to(dev).eval()
    b = synthetic_batch(cfg["vocab"], 1, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=100, ragged=True)
    worst = _mega_vs_launch_pass(dev, model, b, 4, 20, 10)
    print(f"cfg5: worst relative log-probability difference {worst:.2e}")


def test_persistent_decode_beam_search_equals_launch_pass(dev):
    """Whole searches (beam 5, reference defaults) through beam_search_decode with the persistent launch on (default for <= 8
    hypotheses on a bf16 model) and off (MTN_DECODE_MEGA=0): same number of hypotheses, best score within 1e-2 — and against the
    reference's golden best score within the bf16 bar."""
    from mtn_amd import decode as D
    name = "cfg1_query"
    c = fx.GOLDEN_CONFIGS[name]
    g = dict(np.load(os.path.join(GOLD, name + ".npz")))
    model = build_model(c, torch.bfloat16, dev).eval()
    b = dev_batch(one_dialogue(c), dev)
    D._SESSIONS.clear()
    n1, best1 = D.beam_search_decode(model, b, 8, fx.SOS, fx.UNK, fx.EOS, fx.PAD)
    assert any(isinstance(s[0], D.MegaDecodeSession) for s in D._SESSIONS.values()), "the persistent launch was not taken"
    os.environ["MTN_DECODE_MEGA"] = "0"
    try:
        D._SESSIONS.clear()
        n0, best0 = D.beam_search_decode(model, b, 8, fx.SOS, fx.UNK, fx.EOS, fx.PAD)
    finally:
        del os.environ["MTN_DECODE_MEGA"]
        D._SESSIONS.clear()
    assert len(n1) == len(n0) == int(g["beam.n"])
    assert abs(best1 - best0) < 1e-2 * max(1.0, abs(best0))
    assert abs(best1 - float(g["beam.best"])) < 1e-2 * max(1.0, abs(float(g["beam.best"])))


def test_captured_encoder_side_pass_equals_eager_over_several_dialogues(dev):
    """A session reused for further dialogues of a shape stages their inputs into a static Batch and replays ONE graph for the encoder-side
    pass (DecodeSession.load) from the third dialogue on.  Five different dialogues through one session (eager, staged eager, captured,
    replayed, replayed) against each dialogue through a fresh session: identical n-best lists and scores — and the caller's tensors untouched."""
    from mtn_amd import decode as D
    from mtn_amd import make_model
    from mtn_amd.synthetic import CONFIGS, synthetic_batch
    cfg = dict(CONFIGS["cfg2"])
    torch.manual_seed(0)
    model = make_model(cfg["vocab"], cfg["vocab"], N=2, d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"], dropout=0.1,
                       ft_sizes=cfg["ft_sizes"], diff_encoder=True, auto_encoder_ft="query", compute_dtype=torch.bfloat16).to(dev).eval()
    bs = [synthetic_batch(cfg["vocab"], 1, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=200 + i, ragged=True)
          for i in range(5)]
    keep = [b.query.clone() for b in bs]
    D._SESSIONS.clear()
    reused = [D.beam_search_decode(model, b, 12, 2, 0, 3, 1, beam=4) for b in bs]
    sess = [s[0] for s in D._SESSIONS.values()]
    assert len(sess) == 1 and sess[0]._load_graph not in (None, False), "the encoder-side pass was not captured"
    fresh = []
    for b in bs:
        D._SESSIONS.clear()
        fresh.append(D.beam_search_decode(model, b, 12, 2, 0, 3, 1, beam=4))
    D._SESSIONS.clear()
    for (n1, b1), (n0, b0) in zip(reused, fresh):
        assert [h[0] for h in n1] == [h[0] for h in n0]
        assert b1 == b0 and [h[1] for h in n1] == [h[1] for h in n0]
    assert all(torch.equal(b.query, k) for b, k in zip(bs, keep))


@pytest.mark.parametrize("D_,beam", [(1, 4), (1, 5), (2, 4), (1, 8), (4, 4), (3, 5), (2, 8)])
def test_device_side_beam_bookkeeping_equals_host_loop_exactly(dev, D_, beam, monkeypatch):
    """The whole search as one graph replay (mtn_beam_advance keeps the hypotheses on the device) against the same session driven step by
    step from the host (the reference's bookkeeping, data_utils.py:209-240, in Python): the per-step kernels are the same, so the n-best
    lists, every score and the best score must be EQUAL — penalty, min_len and the <unk> / <eos> skips included."""
    from mtn_amd import decode as D
    from mtn_amd import make_model
    from mtn_amd.synthetic import CONFIGS, synthetic_batch
    cfg = dict(CONFIGS["cfg2"])
    torch.manual_seed(1)
    model = make_model(cfg["vocab"], cfg["vocab"], N=2, d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"], dropout=0.1,
                       ft_sizes=cfg["ft_sizes"], diff_encoder=True, auto_encoder_ft="query", compute_dtype=torch.bfloat16).to(dev).eval()
    for seed, (penalty, min_len) in enumerate([(1.0, 1), (0.3, 3), (2.0, 0)]):
        b = synthetic_batch(cfg["vocab"], D_, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=300 + seed, ragged=True)
        D._SESSIONS.clear()
        dev_res = D.beam_search_decode_many(model, b, 14, 2, 0, 3, 1, beam=beam, penalty=penalty, nbest=5, min_len=min_len)
        sess = [s[0] for s in D._SESSIONS.values()]
        assert len(sess) == 1 and isinstance(sess[0], D.MegaDecodeSession) and getattr(sess[0], "_search_key", None) is not None, "the device-side search did not run"
        with monkeypatch.context() as mp:
            mp.setattr(D.MegaDecodeSession, "search", lambda self, *a, **k: None)
            host_res = D.beam_search_decode_many(model, b, 14, 2, 0, 3, 1, beam=beam, penalty=penalty, nbest=5, min_len=min_len)
        assert dev_res == host_res
    D._SESSIONS.clear()


def test_device_side_greedy_equals_host_loop(dev, monkeypatch):
    """greedy_decode as one graph replay (a beam of one on the device) against the same session stepped from the host: same tokens."""
    from mtn_amd import decode as D
    from mtn_amd import make_model
    from mtn_amd.synthetic import CONFIGS, synthetic_batch
    cfg = dict(CONFIGS["cfg2"])
    torch.manual_seed(2)
    model = make_model(cfg["vocab"], cfg["vocab"], N=2, d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"], dropout=0.1,
                       ft_sizes=cfg["ft_sizes"], diff_encoder=True, auto_encoder_ft="query", compute_dtype=torch.bfloat16).to(dev).eval()
    b = synthetic_batch(cfg["vocab"], 1, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=400, ragged=True)
    D._SESSIONS.clear()
    a = D.greedy_decode(model, b, 16, 2)
    sess = [s[0] for s in D._SESSIONS.values()]
    assert isinstance(sess[0], D.MegaDecodeSession) and getattr(sess[0], "_search_key", None) is not None
    with monkeypatch.context() as mp:
        mp.setattr(D.MegaDecodeSession, "greedy", lambda self, *a, **k: None)
        h = D.greedy_decode(model, b, 16, 2)
    os.environ["MTN_DECODE_MEGA"] = "0"
    try:
        D._SESSIONS.clear()
        launch = D.greedy_decode(model, b, 16, 2)
    finally:
        del os.environ["MTN_DECODE_MEGA"]
        D._SESSIONS.clear()
    assert torch.equal(a, h) and a.shape == (1, 16)
    assert (a == launch).float().mean() > 0.8          # (the launch path's log-probabilities differ in the last bf16 bits: an argmax may flip)


@pytest.mark.parametrize("d,h,dff,width", [(1024, 16, 4096, 2), (256, 4, 1024, 6)])
def test_persistent_decode_step_other_widths(dev, d, h, dff, width):
    """The persistent decode step at the edges of its shape range: d_model 1024 with d_ff 4096 (contraction steps beyond the prefetched
    fragments, the largest activation image), d_model 256 with four heads of 64 — against the launch-per-sublayer pass."""
    from mtn_amd import make_model
    from mtn_amd.synthetic import synthetic_batch
    torch.manual_seed(3)
    model = make_model(500, 500, N=2, d_model=d, d_ff=dff, h=h, dropout=0.1, ft_sizes=[64, 32], diff_encoder=True, auto_encoder_ft="query",
                       compute_dtype=torch.bfloat16).to(dev).eval()
    b = synthetic_batch(500, 1, 12, 40, 16, 10, [8, 8], [64, 32], device=dev, seed=11, ragged=True)
    worst = _mega_vs_launch_pass(dev, model, b, width, 9, 6)
    print(f"d_model {d}, {h} heads, d_ff {dff}, width {width}: worst relative log-probability difference {worst:.2e}")


def test_persistent_decode_falls_back_when_compute_units_are_taken(dev):
    """The persistent step needs every workgroup of its launch resident at once.  Here 128 compute units are held by another kernel on a
    second stream (mtn_debug_hold_cus: 128 workgroups x 150 KiB of LDS for 0.6 s; the step's launch is 192 workgroups at beam 4) while a beam search starts: the step's polls time out
    (50 ms, once — later launches of the captured search leave at entry), beam_search_decode notices, recovers the session and re-runs the
    search on the launch-per-sublayer pass: the result EQUALS that pass's own (MTN_DECODE_MEGA=0), FALLBACKS counts it, nothing raises —
    and once the compute units are free again the same session decodes on the persistent step as before."""
    import time
    from mtn_amd import decode as D
    from mtn_amd import lib as L
    from mtn_amd import make_model
    from mtn_amd.synthetic import CONFIGS, synthetic_batch
    cfg = dict(CONFIGS["cfg2"])
    torch.manual_seed(4)
    model = make_model(cfg["vocab"], cfg["vocab"], N=2, d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"], dropout=0.1,
                       ft_sizes=cfg["ft_sizes"], diff_encoder=True, auto_encoder_ft="query", compute_dtype=torch.bfloat16).to(dev).eval()
    b = synthetic_batch(cfg["vocab"], 1, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=500, ragged=True)
    search = lambda: D.beam_search_decode(model, b, 12, 2, 0, 3, 1, beam=4)
    D._SESSIONS.clear()
    clean = search()                                            # builds + captures the persistent session, nothing in its way
    sess = [s_[0] for s_ in D._SESSIONS.values() if isinstance(s_[0], D.MegaDecodeSession)]
    assert len(sess) == 1 and not sess[0].timed_out()
    os.environ["MTN_DECODE_MEGA"] = "0"
    try:
        launch = search()                                       # the launch-per-sublayer pass's own result (its session stays cached)
    finally:
        del os.environ["MTN_DECODE_MEGA"]
    before = D.MegaDecodeSession.FALLBACKS
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        L.check(L.load().mtn_debug_hold_cus(128, 150 * 1024, 600000, L.stream_ptr()))
    time.sleep(0.05)                                            # (the holders are running)
    t0 = time.time()
    held = search()
    dt = time.time() - t0
    assert D.MegaDecodeSession.FALLBACKS == before + 1, "the persistent step did not time out (were 128 compute units really held?)"
    assert held == launch
    assert dt < 1.5, f"fallback took {dt:.2f} s: one 50 ms timeout + one search on the launch pass expected"
    torch.cuda.synchronize()                                    # the holders are gone
    again = search()
    assert D.MegaDecodeSession.FALLBACKS == before + 1 and not sess[0].timed_out()
    assert again == clean
    D._SESSIONS.clear()
