import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtn_amd import make_model, ops
from mtn_amd.synthetic import CONFIGS, synthetic_batch
from mtn_amd.train_step import TrainStep
dev = torch.device("cuda:0")
cfg = dict(CONFIGS["cfg2"]); torch.manual_seed(0)
model = make_model(cfg["vocab"], cfg["vocab"], N=cfg["N"], d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"], dropout=0.1,
                   ft_sizes=cfg["ft_sizes"], diff_encoder=True, auto_encoder_ft="query", compute_dtype=torch.bfloat16).to(dev).train()
batch = synthetic_batch(cfg["vocab"], 32, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=1)
ts = TrainStep(model, batch, cfg["vocab"], use_graph=False)
hits = [0, 0]
orig = ops.SublayerGroupFn.forward
def fwd(ctx, members, *tensors):
    out = orig(ctx, members, *tensors)
    hits[0] += sum(1 for x in ctx.xaccs if x is not None); hits[1] += len(ctx.xaccs)
    return out
ops.SublayerGroupFn.forward = staticmethod(fwd)
ts._fwd_bwd(); torch.cuda.synchronize()
print("x-role hits", hits)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=False) as prof:
    ts._fwd_bwd(); torch.cuda.synchronize()
evs = [e for e in prof.key_averages() if e.key in ("aten::add", "aten::add_", "aten::mul", "aten::sum", "aten::copy_", "aten::zero_", "aten::fill_")]
for e in evs: print(e.key, e.count)
