"""Cost of the layer-segmented schedule on one GPU: single graph vs N+3 graphs without any exchange (the single-rank RCCL variant is `MTN_FORCE_DIST=1 python bench.py`)."""
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


dev = torch.device("cuda:0")
cfg = dict(CONFIGS["cfg2"])
def build(sync, overlap):
    torch.manual_seed(0)
    model = make_model(cfg["vocab"], cfg["vocab"], N=cfg["N"], d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"], dropout=0.1,
                       ft_sizes=cfg["ft_sizes"], diff_encoder=True, auto_encoder_ft="query", compute_dtype=torch.bfloat16).to(dev).train()
    batch = synthetic_batch(cfg["vocab"], 32, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=1)
    return TrainStep(model, batch, cfg["vocab"], grad_sync=sync, overlap=overlap)

def timeit(ts, n=40):
    for _ in range(5): ts()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): ts()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

print("single graph            %.3f ms" % timeit(build(None, False)))
print("segmented, no exchange  %.3f ms" % timeit(build(NoSync(), True)))
print("two graphs, no exchange %.3f ms" % timeit(build(NoSync(), False)))
