"""What the data-parallel schedule costs on ONE GPU when every collective really goes through RCCL (a one-rank "nccl" group,
MTN_FORCE_DIST=1), and where that time goes.  Prints
  * ms/step of the layer-segmented schedule with the collectives issued (dp.ShardedOptimizerSync.collective = True) and skipped;
  * the stand-alone duration of the two collectives on a slice of one decoder layer (16.8 M floats), in place, as the chain
    issues them, and of a plain device copy of the same bytes for scale;
run it under `rocprofv3 --kernel-trace --stats` to get the ncclDevKernel rows (tools/prof_summary.py).

    MTN_FORCE_DIST=1 python tools/dp_rccl_probe.py [--batch 32] [--steps 20]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MTN_FORCE_DIST", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    from mtn_amd import dp, lib, make_model
    from mtn_amd.synthetic import CONFIGS, synthetic_batch
    from mtn_amd.train_step import TrainStep
    dp.init_distributed()
    dev = torch.device("cuda:0")
    lib.load()
    cfg = dict(CONFIGS["cfg2"])
    torch.manual_seed(0)
    model = make_model(cfg["vocab"], cfg["vocab"], N=cfg["N"], d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"], dropout=0.1,
                       ft_sizes=cfg["ft_sizes"], diff_encoder=True, auto_encoder_ft="query", compute_dtype=torch.bfloat16, attn_dropout=0.1).to(dev).train()
    model.prepare()
    batch = synthetic_batch(cfg["vocab"], args.batch, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=1)

    def timed(step, n):
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    res = {}
    for name, coll, sharded in (("sharded, collectives through RCCL", True, "1"), ("sharded, collectives skipped", False, "1"),
                                ("all-reduce per slice + full Adam, through RCCL", True, "0")):
        os.environ["MTN_DP_SHARDED"] = sharded
        sync = dp.GradSync(lambda: model.flat_buffers()[2])
        st = TrainStep(model, batch, cfg["vocab"], pad=1, warmup=4000, grad_sync=sync)
        if st.sharded is not None:
            st.sharded.collective = coll
        res[name] = timed(st, args.steps)
        del st
    st = TrainStep(model, batch, cfg["vocab"], pad=1, warmup=4000, grad_sync=None)
    res["one rank, optimiser in the dW launch (bench line)"] = timed(st, args.steps)
    for k, v in res.items():
        print(f"{k:60s} {v:8.3f} ms/step")

    # the collectives by themselves
    n = model._layer_slices[0][1] - model._layer_slices[0][0]
    buf = torch.zeros(n, device=dev)
    other = torch.zeros(n, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def ev(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    print(f"slice = one decoder layer = {n} floats = {n * 4 / 1e6:.1f} MB, world = {dist.get_world_size()}, backend = {dist.get_backend()}")
    print(f"  reduce_scatter_tensor in place   {ev(lambda: dist.reduce_scatter_tensor(buf, buf)):9.1f} us")
    print(f"  all_gather_into_tensor in place  {ev(lambda: dist.all_gather_into_tensor(buf, buf)):9.1f} us")
    print(f"  all_reduce                       {ev(lambda: dist.all_reduce(buf)):9.1f} us")
    print(f"  torch copy_ of the same bytes    {ev(lambda: other.copy_(buf)):9.1f} us")
    small = torch.zeros(1024, device=dev)
    print(f"  all_reduce of 4 KB               {ev(lambda: dist.all_reduce(small)):9.1f} us")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
