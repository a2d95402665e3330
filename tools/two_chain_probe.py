"""Would two half-batch chains side by side beat one full-batch chain?  The train step is a chain of ~240 dependent, latency-bound
launches of <= 256 one-per-CU workgroups.  Two INDEPENDENT half-batch steps (two models, batch 16 each, one captured graph each)
replayed on two streams at the same time would, if the runtime co-schedules them on disjoint halves of the chip, finish 32 samples
in the time of one half-batch step.  This probe measures the ceiling of that idea before anything is built on it:

    one step, batch 32            (the bench line)
    one step, batch 16            (how much of the step is latency: a half batch should take half the time if it were throughput)
    two batch-16 steps at once    (two streams, graphs replayed alternately)

    python tools/two_chain_probe.py [--reps 30]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    from mtn_amd import lib, make_model
    from mtn_amd.synthetic import CONFIGS, synthetic_batch
    from mtn_amd.train_step import TrainStep
    dev = torch.device("cuda:0")
    lib.load()
    cfg = dict(CONFIGS["cfg2"])

    def build(B, seed):
        torch.manual_seed(seed)
        m = make_model(cfg["vocab"], cfg["vocab"], N=cfg["N"], d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"], dropout=0.1,
                       ft_sizes=cfg["ft_sizes"], diff_encoder=True, auto_encoder_ft="query", compute_dtype=torch.bfloat16, attn_dropout=0.1).to(dev).train()
        m.prepare()
        b = synthetic_batch(cfg["vocab"], B, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=seed)
        st = TrainStep(m, b, cfg["vocab"], pad=1, warmup=4000)
        for _ in range(3):
            st()
        torch.cuda.synchronize()
        return st

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    s32 = build(32, 0)
    t32 = timed(s32, args.reps)
    del s32
    a, b = build(16, 1), build(16, 2)
    t16 = timed(a, args.reps)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def both():
        with torch.cuda.stream(s1):
            a._g_fb.replay()
        with torch.cuda.stream(s2):
            b._g_fb.replay()

    t2 = timed(both, args.reps)
    print(f"one step, batch 32                 : {t32:7.3f} ms  = {32 / t32 * 1e3:8.0f} samples/s")
    print(f"one step, batch 16                 : {t16:7.3f} ms  = {16 / t16 * 1e3:8.0f} samples/s  ({t16 / t32:.2f} of the batch-32 step for half the samples)")
    print(f"two batch-16 steps on two streams  : {t2:7.3f} ms  = {32 / t2 * 1e3:8.0f} samples/s  ({t32 / t2:.2f}x the batch-32 step; 2.00x a batch-16 step would be perfect overlap: {2 * t16 / t2:.2f}x)")


if __name__ == "__main__":
    main()
