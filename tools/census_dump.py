"""List every GEMM launch of one train step (library launch census): kernel, workgroups, shapes, time."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
ts._fwd_bwd(); torch.cuda.synchronize()
lib = L.load(); lib.mtn_census_begin(); ts._fwd_bwd(); torch.cuda.synchronize(); n = lib.mtn_census_end()
st = torch.cuda.current_stream(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(n):
    info = L.CensusLaunch(); lib.mtn_census_info(i, C.byref(info))
    lib.mtn_census_replay(i, 2, st.cuda_stream); e0.record(st); lib.mtn_census_replay(i, 10, st.cuda_stream); e1.record(st); e1.synchronize()
    us = e0.elapsed_time(e1) * 100
    shapes = " ".join(f"{info.M[k]}x{info.N[k]}x{info.K[k]}" for k in range(min(4, info.count)))
    print(f"{i:3d} {lib.mtn_census_variant_name(info.variant).decode():34s} n={info.count:2d} wgs={info.workgroups:5d} {us:7.2f} us {info.flops/us/1e6:7.1f} TF  {shapes}")
