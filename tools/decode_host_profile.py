"""Where the HOST time of a beam-search step goes (cProfile over bench.decode_measure's one-dialogue loop)."""
import cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from mtn_amd import make_model
from mtn_amd.synthetic import CONFIGS

cfg = dict(CONFIGS["cfg2"])
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = make_model(cfg["vocab"], cfg["vocab"], N=cfg["N"], d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"], dropout=0.1,
                   ft_sizes=cfg["ft_sizes"], diff_encoder=True, auto_encoder_ft="query", compute_dtype=torch.bfloat16).to(dev)
bench.decode_measure(model, cfg, dev, cpu=False)          # warm: sessions and graphs built
pr = cProfile.Profile()
pr.enable()
out = bench.decode_measure(model, cfg, dev, cpu=False)
pr.disable()
print({k: v for k, v in out.items() if k != "roofline"})
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
