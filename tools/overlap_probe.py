"""How much does a concurrent optimiser pass (HBM-bound Adam + transposes on a second stream) slow the launch-latency-bound
forward+backward graph?  Decides whether per-layer optimiser work is worth moving under the backward pass."""
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
torch.manual_seed(0)
model = make_model(cfg["vocab"], cfg["vocab"], N=cfg["N"], d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"], dropout=0.1,
                   ft_sizes=cfg["ft_sizes"], diff_encoder=True, auto_encoder_ft="query", compute_dtype=torch.bfloat16).to(dev).train()
batch = synthetic_batch(cfg["vocab"], 32, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=1)
ts = TrainStep(model, batch, cfg["vocab"], grad_sync=NoSync(), overlap=False)
ts()
torch.cuda.synchronize()
fb, opt = ts._g_fb, ts._g_opt
side = torch.cuda.Stream()


def timeit(fn, n=30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


t_fb = timeit(fb.replay)
t_opt = timeit(opt.replay)
def both():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        opt.replay()
    fb.replay()
    torch.cuda.current_stream().wait_stream(side)
t_both = timeit(both)
print(f"fwd+bwd graph alone {t_fb:.3f} ms, optimiser graph alone {t_opt:.3f} ms, serial {t_fb + t_opt:.3f} ms, concurrent {t_both:.3f} ms")
