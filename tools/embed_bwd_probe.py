# time the embedding backward on a ragged cfg2-like token set (25 % pads) atomic vs deterministic kernel; second line per B = uniform tokens
import math, os, sys, torch
sys.path.insert(0, "/root/repo")
from mtn_amd import lib as L
lib = L.load(); dev = torch.device("cuda")
V, d = 3000, 512
g = torch.Generator().manual_seed(3)
for B, RAGGED in ((32, True), (32, False), (128, True), (128, False)):
    toks, dxs = [], []
    for Lq in (20, 128, 40, 20):
        t = torch.randint(4, V, (B, Lq), generator=g)
        if RAGGED:
            lens = torch.randint(Lq // 2, Lq + 1, (B,), generator=g)
            t[torch.arange(Lq)[None, :] >= lens[:, None]] = 1
        toks.append(t.to(dev).reshape(-1).contiguous()); dxs.append(torch.randn(B * Lq, d, generator=g).to(dev))
    for mode in ("atomic", "det"):
        os.environ["MTN_EMBED_DETERMINISTIC"] = "0" if mode == "atomic" else "1"
        L.reload_env()
        dlut = torch.zeros(V, d, device=dev)
        descs = (L.EmbedBwdDesc * 4)()
        for E, t, x in zip(descs, toks, dxs):
            E.rows, E.d, E.tokens, E.dx, E.emb_scale = t.numel(), d, t.data_ptr(), x.data_ptr(), math.sqrt(d)
            E.dlut, E.lut_rows = dlut.data_ptr(), V
        for _ in range(3): L.check(lib.mtn_embed_bwd_group(4, descs, L.stream_ptr()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): L.check(lib.mtn_embed_bwd_group(4, descs, L.stream_ptr()))
        e1.record(); torch.cuda.synchronize()
        print(f"B={B} ragged={RAGGED} {mode:>6}: {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us", flush=True)
