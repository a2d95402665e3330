"""Can an HBM-bound pass hide under the latency-bound backward chain of the train step?  

The step's parameter-gradient + optimiser launch is HBM-bound (28 B per weight behind every 2K flops) and runs strictly after
a forward/backward chain of ~240 dependent launches whose waves wait most of the time.  Round 1 tried an unmasked second stream
(-8 %).  This probe measures the ingredients with a CU-MASKED side stream (mtn_stream_create_cu_masked):

  main : the captured forward+backward graph of the cfg2 step (optimiser excluded), replayed on the main stream;
  side : a pure HBM stream of the optimiser's size on the side stream — mtn_adam_step over a second set of buffers
         (mtn_adam_step with the bf16 copy: 30 B/param over 106.65 M parameters = 3.2 GB; the fused launch moves 3.36 GB);
         it is data-independent of the main graph, so the two can be overlapped at will, which is the BEST case for overlap.

For n in {256 (no mask), 128, 64, 32} CUs on the side stream it reports: the side pass alone, the main graph alone, and both
started together (main duration and side duration by HIP events on their streams, makespan = host wall around both) against
running one after the other on the whole chip.  Overlap pays only if makespan < main alone + side alone at 256 CUs.

    python tools/overlap_cu_mask_probe.py [--batch 32] [--reps 20]
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    from mtn_amd import lib as L, make_model
    from mtn_amd.synthetic import CONFIGS, synthetic_batch
    from mtn_amd.train_step import TrainStep
    dev = torch.device("cuda:0")
    lib = L.load()
    cfg = dict(CONFIGS["cfg2"])
    torch.manual_seed(0)
    model = make_model(cfg["vocab"], cfg["vocab"], N=cfg["N"], d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"], dropout=0.1,
                       ft_sizes=cfg["ft_sizes"], diff_encoder=True, auto_encoder_ft="query", compute_dtype=torch.bfloat16, attn_dropout=0.1).to(dev).train()
    model.prepare()
    batch = synthetic_batch(cfg["vocab"], args.batch, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=1)
    st = TrainStep(model, batch, cfg["vocab"], pad=1, warmup=4000, fuse_optimizer=False)
    # main graph: forward + loss + backward + the parameter-gradient launches (no optimiser)
    side_warm = torch.cuda.Stream()
    side_warm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side_warm):
        for _ in range(2):
            st._fwd_bwd()
    torch.cuda.current_stream().wait_stream(side_warm)
    torch.cuda.synchronize()
    g_main = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_main):
        st._fwd_bwd()
    # side pass: Adam over an independent set of buffers of the model's size
    n = model._flat.numel()
    p, g, m, v = (torch.zeros(n, device=dev) for _ in range(4))
    g.normal_()
    p_lp = torch.zeros(n, device=dev, dtype=torch.bfloat16)
    state = torch.tensor([1.0, 1e-4, 0.1, 0.02, 0, 0, 0, 0], device=dev)

    def side_pass(stream_ptr):
        L.check(lib.mtn_adam_step(L.MTN_BF16, n, p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p_lp.data_ptr(), state.data_ptr(),
                                  None, 0.9, 0.98, 1e-9, stream_ptr))

    main_s = torch.cuda.current_stream()
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def run(n_cus, mode):
        """mode: 'main', 'side', 'both' -> (main ms, side ms, makespan ms), averaged over reps"""
        sp = C.c_void_p()
        if n_cus >= 256:
            side = torch.cuda.Stream()
            side_ptr = side.cuda_stream
        else:
            L.check(lib.mtn_stream_create_cu_masked(n_cus, 0, C.byref(sp)))
            side = torch.cuda.ExternalStream(sp.value)
            side_ptr = sp.value
        tm = ts = tt = 0.0
        offs = [0.0, 0.0, 0.0]
        for rep in range(args.reps + 2):
            torch.cuda.synchronize()
            e0, e1, s0, s1, ref = ev(), ev(), ev(), ev(), ev()
            t0 = time.perf_counter()
            ref.record(main_s)
            if mode in ("side", "both"):
                s0.record(side)
                side_pass(side_ptr)
                s1.record(side)
            if mode in ("main", "both"):
                e0.record(main_s)
                g_main.replay()
                e1.record(main_s)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
            if rep >= 2:
                tm += e0.elapsed_time(e1) if mode != "side" else 0.0
                ts += s0.elapsed_time(s1) if mode != "main" else 0.0
                tt += wall
                if mode == "both":       # where the side pass sat relative to the main graph (same clock: offsets from one reference event)
                    offs[0] += ref.elapsed_time(s0); offs[1] += ref.elapsed_time(s1); offs[2] += ref.elapsed_time(e1)
        if n_cus < 256:
            torch.cuda.synchronize()
            L.check(lib.mtn_stream_destroy(sp))
        r = args.reps
        return tm / r, ts / r, tt / r, [o / r for o in offs]

    print(f"# cfg2 batch {args.batch}: main = captured forward+backward graph; side = adam_kernel over {n} elements (30 B/param = {30 * n / 1e9:.2f} GB)")
    main_alone = run(256, "main")
    print(f"main alone                         : main {main_alone[0]:7.3f} ms   (host wall {main_alone[2]:.3f})")
    base_side = None
    for n_cus in (256, 128, 64, 32):
        a = run(n_cus, "side")
        b = run(n_cus, "both")
        if n_cus == 256:
            base_side = a[1]
        seq = main_alone[2] + base_side          # host wall of the main graph alone + the side pass alone on the whole chip
        print(f"side on {n_cus:3d} CUs: side alone {a[1]:7.3f} ms = {30 * n / 1e9 / (a[1] * 1e-3) / 1e3:5.2f} TB/s | together: main {b[0]:7.3f} ms, "
              f"side {b[1]:7.3f} ms, makespan (host wall) {b[2]:7.3f} ms | one after the other on the whole chip {seq:7.3f} ms -> "
              f"{'gain' if b[2] < seq else 'loss'} {100 * (seq - b[2]) / seq:+.1f} %   [side ran from +{b[3][0]:.3f} to +{b[3][1]:.3f} ms, main ended at +{b[3][2]:.3f} ms]")


if __name__ == "__main__":
    main()
