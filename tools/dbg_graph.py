import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtn_amd import make_model
from mtn_amd.synthetic import CONFIGS, synthetic_batch
from mtn_amd.train_step import TrainStep
dev = torch.device("cuda:0")
cfg = dict(CONFIGS["cfg1"])
def run(use_graph, steps=4):
    torch.manual_seed(0)
    m = make_model(cfg["vocab"], cfg["vocab"], N=cfg["N"], d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"], dropout=0.0,
                   ft_sizes=cfg["ft_sizes"], diff_encoder=True, auto_encoder_ft="query", compute_dtype="bf16", attn_dropout=0.0).to(dev).train()
    b = synthetic_batch(cfg["vocab"], cfg["B"], cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=1)
    st = TrainStep(m, b, cfg["vocab"], warmup=10, use_graph=use_graph)
    losses = []
    for _ in range(steps):
        losses.append(float(st()))
    torch.cuda.synchronize()
    return losses, m._flat.clone(), st.opt.optimizer.state.clone()
le, pe, se = run(False)
lg, pg, sg = run(True)
print("eager", le, se[:4].tolist())
print("graph", lg, sg[:4].tolist())
print("param diff", float((pe - pg).abs().max()), float(pe.abs().max()))
