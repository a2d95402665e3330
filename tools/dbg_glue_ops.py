"""Which PyTorch (non-library) ops still run inside one train step, and from where: one eager fused step under torch.profiler with stacks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtn_amd import make_model
from mtn_amd.synthetic import CONFIGS, synthetic_batch
from mtn_amd.train_step import TrainStep
dev = torch.device("cuda:0")
cfg = dict(CONFIGS["cfg2"]); torch.manual_seed(0)
model = make_model(cfg["vocab"], cfg["vocab"], N=cfg["N"], d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"], dropout=0.1,
                   ft_sizes=cfg["ft_sizes"], diff_encoder=True, auto_encoder_ft="query", compute_dtype=torch.bfloat16).to(dev).train()
batch = synthetic_batch(cfg["vocab"], 32, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=1)
ts = TrainStep(model, batch, cfg["vocab"], use_graph=False)
for _ in range(2): ts()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    ts(); torch.cuda.synchronize()
seen = {}
LAUNCHING = ("aten::add", "aten::add_", "aten::copy_", "aten::zero_", "aten::fill_", "aten::sum", "aten::mul", "aten::mul_", "aten::div", "aten::ne",
             "aten::index_select", "aten::embedding", "aten::cat", "aten::stack", "aten::clone", "aten::masked_fill_", "aten::sub", "aten::neg")
for e in prof.events():
    if e.name in LAUNCHING:
        st = [f for f in (e.stack or []) if "mtn_amd" in f or "train_step" in f][:3]
        key = (e.name, tuple(st))
        c = seen.setdefault(key, [0, 0.0]); c[0] += 1; c[1] += e.cpu_time_total
for (name, st), (n, us) in sorted(seen.items(), key=lambda kv: -kv[1][0]):
    print(f"{name:22s} n={n:3d}  " + " <- ".join(s.split('/')[-1] for s in st))
