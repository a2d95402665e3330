"""How the workgroups of ONE fused_head_bwd_kernel launch are spread over the chip and over time (a launch of more workgroups
than CUs: batch 64).  Development build of the library (fused_bwd.hip compiled with -DFB_TIMELINE):
    SRC=fused_bwd bash tools/build_variant.sh fbtl -DFB_TIMELINE
    MTN_HIP_LIB=tools/libmtn_hip_fbtl.so MTN_FB_TL_LAUNCH=<n> python tools/fb_rounds.py [batch]
Per workgroup: entry and end stamps (100 MHz wall clock) and the XCC / SE / CU it ran on."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mtn_amd import make_model, lib as L
from mtn_amd.synthetic import CONFIGS, synthetic_batch
from mtn_amd.train_step import TrainStep
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
cfg = dict(CONFIGS["cfg2"]); torch.manual_seed(0)
model = make_model(cfg["vocab"], cfg["vocab"], N=cfg["N"], d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"], dropout=0.1,
                   ft_sizes=cfg["ft_sizes"], diff_encoder=True, auto_encoder_ft="query", compute_dtype=torch.bfloat16).to(dev).train()
batch = synthetic_batch(cfg["vocab"], B, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=1)
ts = TrainStep(model, batch, cfg["vocab"], use_graph=False)
for _ in range(3):
    ts._fwd_bwd()
torch.cuda.synchronize()
lib = L.load()
buf = (C.c_ulonglong * (1024 * 16))()
lib.mtn_fb_timeline_read.restype = C.c_int
assert lib.mtn_fb_timeline_read(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16)
live = t[:, 0] > 0
n = int(live.sum())
t = t[live]
st = t[:, 0].astype(np.int64)
en = t[:, 11].astype(np.int64)
base = st.min()
st_us, en_us = (st - base) / 100.0, (en - base) / 100.0
hw = t[:, 15]
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
hwid = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 0x1; se = (hwid >> 13) & 0x7
place = xcc * 1000 + se * 100 + sh * 50 + cu
print(f"batch {B}: {n} workgroups; launch span {en_us.max():.2f} us; workgroup duration median {np.median(en_us - st_us):.2f} us, max {(en_us - st_us).max():.2f}")
print(f"distinct (XCC, SE, SH, CU) places used: {len(set(place.tolist()))}")
first = st_us < 2.0
print(f"workgroups entering within 2 us of the first: {int(first.sum())}; their duration median {np.median((en_us - st_us)[first]):.2f} us, end median {np.median(en_us[first]):.2f} max {en_us[first].max():.2f}")
late = ~first
if late.any():
    print(f"later workgroups: {int(late.sum())}; entry median {np.median(st_us[late]):.2f} (min {st_us[late].min():.2f}, max {st_us[late].max():.2f}); duration median {np.median((en_us - st_us)[late]):.2f} us; end median {np.median(en_us[late]):.2f} max {en_us[late].max():.2f}")
per = {}
for p_, s_, e_ in zip(place.tolist(), st_us.tolist(), en_us.tolist()):
    per.setdefault(p_, []).append((s_, e_))
cnt = np.bincount([len(v) for v in per.values()])
print("workgroups per place: " + ", ".join(f"{k}: {c} places" for k, c in enumerate(cnt) if c))
ov = sum(1 for v in per.values() if len(v) >= 2 and sorted(v)[1][0] < sorted(v)[0][1])
print(f"places where the second workgroup entered BEFORE the first ended (co-resident): {ov}")
names = ["entry", "issued", "landed", "dO mfma", "dO image", "frags+D", "S,dP mfma", "elementwise", "dV,dK", "dQ mfma", "tiles done", "end"]
for label, sel in (("first round", first), ("later", late)):
    if not sel.any():
        continue
    print(label + ": median stamp since the workgroup's own entry")
    for k, nme in enumerate(names):
        ok = sel & (t[:, k] > 0)
        if ok.any():
            print(f"   {k:2d} {nme:12s} {np.median((t[ok, k].astype(np.int64) - st[ok]) / 100.0):6.2f}")
