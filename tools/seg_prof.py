import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtn_amd import make_model
from mtn_amd.synthetic import CONFIGS, synthetic_batch
from mtn_amd.train_step import TrainStep
class NoSync:
    world = 1
    def all_reduce_scalars(self, t): return t
    def reduce_range(self, lo, hi): return None
    def wait(self, h): pass
    def __call__(self): pass
dev = torch.device("cuda:0"); cfg = dict(CONFIGS["cfg2"]); torch.manual_seed(0)
model = make_model(cfg["vocab"], cfg["vocab"], N=cfg["N"], d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"], dropout=0.1,
                   ft_sizes=cfg["ft_sizes"], diff_encoder=True, auto_encoder_ft="query", compute_dtype=torch.bfloat16).to(dev).train()
batch = synthetic_batch(cfg["vocab"], 32, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=1)
ts = TrainStep(model, batch, cfg["vocab"], grad_sync=NoSync(), overlap=True)
for _ in range(8): ts()
torch.cuda.synchronize()
