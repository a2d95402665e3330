"""Host-side cost of the layer-segmented DP schedule with a single-rank RCCL group: time to ENQUEUE a step vs time to run it."""
import os, sys, time
os.environ["MTN_FORCE_DIST"] = "1"; os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("RANK", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtn_amd import make_model, dp
from mtn_amd.synthetic import CONFIGS, synthetic_batch
from mtn_amd.train_step import TrainStep
dp.init_distributed()
dev = torch.device("cuda:0")
cfg = dict(CONFIGS["cfg2"]); torch.manual_seed(0)
model = make_model(cfg["vocab"], cfg["vocab"], N=cfg["N"], d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"], dropout=0.1,
                   ft_sizes=cfg["ft_sizes"], diff_encoder=True, auto_encoder_ft="query", compute_dtype=torch.bfloat16).to(dev).train()
batch = synthetic_batch(cfg["vocab"], 32, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=1)
for overlap in (True, False):
    sync = dp.GradSync(lambda: model.flat_buffers()[2])
    ts = TrainStep(model, batch, cfg["vocab"], grad_sync=sync, overlap=overlap)
    for _ in range(5): ts()
    torch.cuda.synchronize()
    n = 40; t0 = time.perf_counter()
    for _ in range(n): ts()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    print(f"overlap={overlap}: enqueue {1e3*t_enq/n:.3f} ms/step, total {1e3*t_all/n:.3f} ms/step")
    if overlap:
        # the pieces
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            for run, _r in ts._g_seg: run()
            ts._g_opt.replay()
        t_enq = time.perf_counter() - t0; torch.cuda.synchronize(); t_all = time.perf_counter() - t0
        print(f"  graphs only (no exchange calls): enqueue {1e3*t_enq/n:.3f}, total {1e3*t_all/n:.3f} ms/step")
torch.distributed.destroy_process_group()
