"""In-kernel timeline of fused_head_bwd_kernel inside a real train step.

Needs the development build of the library (fused_bwd.hip compiled with -DFB_TIMELINE, see tools/README.md):
    MTN_HIP_LIB=tools/libmtn_hip_fbtl.so MTN_FB_TL_LAUNCH=<n> python tools/fb_timeline.py
The n-th fused backward launch of the process records 12 wall-clock stamps (100 MHz) from wave 0 of every workgroup."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mtn_amd import make_model, lib as L
from mtn_amd.synthetic import CONFIGS, synthetic_batch
from mtn_amd.train_step import TrainStep
dev = torch.device("cuda:0")
cfg = dict(CONFIGS["cfg2"]); torch.manual_seed(0)
model = make_model(cfg["vocab"], cfg["vocab"], N=cfg["N"], d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"], dropout=0.1,
                   ft_sizes=cfg["ft_sizes"], diff_encoder=True, auto_encoder_ft="query", compute_dtype=torch.bfloat16).to(dev).train()
batch = synthetic_batch(cfg["vocab"], 32, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=1)
ts = TrainStep(model, batch, cfg["vocab"], use_graph=False)
for _ in range(3):
    ts._fwd_bwd()
torch.cuda.synchronize()
lib = L.load()
buf = (C.c_ulonglong * (256 * 16))()
lib.mtn_fb_timeline_read.restype = C.c_int
assert lib.mtn_fb_timeline_read(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 16).astype(np.int64)
live = t[:, 0] > 0
t = t[live]
names = ["entry", "issued", "landed", "dO mfma", "dO image", "frags+D", "S,dP mfma", "elementwise", "dV,dK", "dQ mfma", "tiles done", "end"]
base = t[:, 0].min()
print(f"{live.sum()} workgroups; stamps in us after the first workgroup's entry (median / max over workgroups); delta = median step")
prev = None
for k, nme in enumerate(names):
    col = (t[:, k] - base) / 100.0
    ok = t[:, k] > 0
    if not ok.any():
        continue
    med, mx = np.median(col[ok]), col[ok].max()
    print(f"{k:2d} {nme:12s} median {med:6.2f}  max {mx:6.2f}" + (f"  (+{med - prev:5.2f})" if prev is not None else ""))
    prev = med
