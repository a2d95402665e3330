"""Where one decode step of the persistent launch (csrc/decode.hip) spends its time: workgroup 0's wall-clock stamps per stage
(MTN_DECODE_TIMELINE=1: behind the barrier / operands ready / computed / stores issued), cfg5 shape (cfg2 model, beam 4).
    python tools/decode_timeline.py [dialogues side by side = 1]      (4 = the 16-row launch)"""
import os, sys
os.environ["MTN_DECODE_TIMELINE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtn_amd import make_model
from mtn_amd.decode import MegaDecodeSession
from mtn_amd.synthetic import CONFIGS, synthetic_batch
dev = torch.device("cuda:0")
cfg = dict(CONFIGS["cfg2"]); torch.manual_seed(0)
model = make_model(cfg["vocab"], cfg["vocab"], N=cfg["N"], d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"], dropout=0.1,
                   ft_sizes=cfg["ft_sizes"], diff_encoder=True, auto_encoder_ft="query", compute_dtype=torch.bfloat16).to(dev).eval()
D = int(sys.argv[1]) if len(sys.argv) > 1 else 1
b = synthetic_batch(cfg["vocab"], D, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=100, ragged=False)
sess = MegaDecodeSession(model, b, 20, 4, pad=1, use_graph=False)
plists = [[[2]] for _ in range(D)]
for l in range(6):
    lps = sess.step_many(plists)
    nxt = []
    for prefixes, lp in zip(plists, lps):
        top = lp.topk(4, dim=-1).indices.tolist()
        nxt.append([p + [int(t)] for p, tt in zip(prefixes, top) for t in tt][:4])
    plists = nxt
torch.cuda.synchronize()
st = sess._dbg.view(-1, 4).cpu().numpy().astype("int64")
names = ["EMBED", "SELF_QKV", "SELF_ATT", "OUT", "CROSS", "FFN1", "FFN2", "FINAL", "CROSS_P", "SELF_ATT_P", "XSUM"]
import ctypes as C
from mtn_amd import lib as L
raw = bytes(sess._stages_dev.cpu().numpy())
kinds = [L.DecodeStage.from_buffer_copy(raw[i * C.sizeof(L.DecodeStage):(i + 1) * C.sizeof(L.DecodeStage)]).kind for i in range(sess._n_stages)]
t0 = st[0, 0]
agg = {}
print(f"{D} dialogue(s) x beam 4 = {4 * D} hypothesis rows")
print(f"step total (first stage start -> last stage stores issued): {(st[-1, 3] - t0) / 100:.1f} us over {len(kinds)} stages")
for i, k in enumerate(kinds):
    prev_end = st[i - 1, 3] if i else st[0, 0]
    barrier = (st[i, 0] - prev_end) / 100           # previous stage's stores issued -> this stage behind its barrier (drain + arrive + poll + prefetch issue)
    ready = (st[i, 1] - st[i, 0]) / 100 if st[i, 1] else 0.0
    comp = (st[i, 2] - max(st[i, 1], st[i, 0])) / 100 if st[i, 2] else 0.0
    store = (st[i, 3] - max(st[i, 2], st[i, 1], st[i, 0])) / 100
    a = agg.setdefault(names[k], [0, 0.0, 0.0, 0.0, 0.0])
    a[0] += 1; a[1] += barrier; a[2] += ready; a[3] += comp; a[4] += store
    if i < 24:
        print(f"{i:3d} {names[k]:9s} barrier {barrier:6.2f}  operands {ready:6.2f}  compute {comp:6.2f}  epilogue {store:6.2f} us")
print("per stage kind (count, mean us): barrier | operands | compute | epilogue")
for k, a in agg.items():
    print(f"{k:9s} n={a[0]:3d}  {a[1] / a[0]:6.2f} | {a[2] / a[0]:6.2f} | {a[3] / a[0]:6.2f} | {a[4] / a[0]:6.2f}   total {sum(a[1:]):7.1f} us")
