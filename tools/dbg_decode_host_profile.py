import cProfile, pstats, sys, torch
sys.path.insert(0, ".")
from mtn_amd import make_model
from mtn_amd.synthetic import CONFIGS, synthetic_batch
from mtn_amd import decode as D
dev = torch.device("cuda:0")
cfg = dict(CONFIGS["cfg2"]); torch.manual_seed(0)
model = make_model(cfg["vocab"], cfg["vocab"], N=cfg["N"], d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"], dropout=0.1,
                   ft_sizes=cfg["ft_sizes"], diff_encoder=True, auto_encoder_ft="query", compute_dtype=torch.bfloat16).to(dev).eval()
b = synthetic_batch(cfg["vocab"], 1, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=1)
def run(n):
    for _ in range(n):
        D.beam_search_decode(model, b, 20, 2, 0, 3, 1, beam=4, penalty=1.0, nbest=4, min_len=1)
run(3); torch.cuda.synchronize()
import time
t=time.time(); run(10); torch.cuda.synchronize(); print("ms per dialogue", (time.time()-t)*100)
pr = cProfile.Profile(); pr.enable(); run(10); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
