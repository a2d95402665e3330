#!/usr/bin/env python
"""bench.py — train-step throughput of the MTN hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...            (spawns its own N ranks: re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the reference's batch loop body (train.py:29-40 + data_utils.py:133-156) over one synthetic
batch already resident in HBM: forward -> generator + label-smoothed loss (main + 2 auto-encoder terms) -> backward ->
[RCCL all-reduce of the flat gradient] -> Adam/Noam update.  Workload at every N: BASELINE.json configs[1] shapes
(d_model=512, 6 layers, 8 heads, d_ff=2048, |V|=3000, Q/H/C/T=20/128/40/20, 32 I3D(2048)+32 VGGish(128) frames); N = 1:
32 samples (cfg2 = BASELINE configs[1], the configuration the metric is quoted on); N > 1: 64 samples PER GPU (cfg3 = BASELINE
configs[2], weak scaling); bf16 compute with fp32 master weights / residual stream / statistics, dropout 0.1 on (in-kernel),
random-init weights, synthetic data.  Rank 0 prints ONE JSON line — the TERSE form of the record (< 7 KB: the driver keeps an 8 KB tail of
stdout; keys explained in README.md "Reading the bench line"; `--full-record PATH` writes the verbose form).  `value` comes from EXACTLY --steps steps between two
barrier + synchronize brackets; four more windows of the same length are timed afterwards and reported beside it
(config.window_ms_per_step, median) because one window of a 4-5 ms step is a short sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md; AMD's 5 PF figure is 2:1 sparse)
PEAK_FP32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
SURVEY_GFLOP_PER_SAMPLE = {"cfg2": 15.90, "cfg3": 15.90, "cfg4": 33.22, "cfg1": 0.315}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None, choices=["cfg1", "cfg2", "cfg3", "cfg4", "avsd32"],
                    help="default: cfg2 (batch 32) on one GPU, cfg3 (batch 64 per GPU) on several")
    ap.add_argument("--batch-per-gpu", type=int, default=None)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--bf16-grad-allreduce", action="store_true")
    ap.add_argument("--windows", type=int, default=4, help="extra timed windows of --steps steps after the one `value` comes from")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements (batch-64 figure, one-rank RCCL "
                    "schedule and decode figures at N=1, exchange-free single-rank figure at N>1)")
    ap.add_argument("--pmc-calibration", action="store_true", help="after the timed steps, one whole-buffer mtn_adam_step over a scratch "
                    "buffer of 2^24 elements: a launch of exactly known HBM traffic (16 B read + 14 B written per element) for "
                    "tools/pmc_summary.py to calibrate the WRITE_SIZE counter on (the fused step itself no longer has one)")
    ap.add_argument("--full-record", default=None, metavar="PATH", help="also write the VERBOSE record (every figure with its prose description, "
                    "per-slice exchange timeline, full kernel tables) to PATH as JSON; the printed line is the terse form of it (README.md, "
                    "\"Reading the bench line\")")
    ap.add_argument("--dp-one-rank-probe", action="store_true", help=argparse.SUPPRESS)     # child process of the N = 1 secondary measurement
    ap.add_argument("--no-record", action="store_true", help="development probe: allows MTN_DP_EMULATE_WORLD (a rank updates 1/N of "
                    "every slice as in an N-GPU job — WRONG parameters, timing only); the line is then marked \"record\": false")
    args = ap.parse_args()
    if args.workload is None:
        args.workload = "cfg2" if args.gpus == 1 else "cfg3"
    return args


def spawn_ranks(args):
    """`python bench.py --gpus N` without a torchrun environment: re-execute under torch.distributed.run, one rank per GPU
    (RCCL over xGMI), rendezvous on 127.0.0.1; the ranks' output (rank 0's JSON line last) passes through."""
    import socket
    import subprocess
    backend = os.environ.get("MTN_DIST_BACKEND") or "nccl"
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and n_dev < args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs for RCCL, this node shows {n_dev} "
                 f"(MTN_DIST_BACKEND=gloo runs the multi-rank control flow on fewer devices, as a dry run only)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.call(cmd, env=env))


def cpu_baseline(model, cfg, B, steps):
    """The CPU oracle (PyTorch-CPU fp32 restatement of the reference, oracle/mtn_oracle.py) running the same step on
    the host cores of this box: forward -> loss -> backward -> Adam/Noam.  Bounded sample: 2 warm-ups + `steps` steps (median)."""
    from oracle import fixtures as fx
    from oracle.mtn_oracle import OracleConfig, OracleMTN, noam_rate
    ocfg = OracleConfig(vocab=cfg["vocab"], n_layers=cfg["N"], d_model=cfg["d_model"], d_ff=cfg["d_ff"], heads=cfg["h"],
                        ft_sizes=tuple(cfg["ft_sizes"]), diff_encoder=True, auto_encoder_ft="query")
    sd = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items() if not k.endswith(".pe")}
    om = OracleMTN(ocfg, sd)
    raw = fx.det_batch(cfg["vocab"], B, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], seed=1, ragged=False)
    ob = fx.oracle_batch(raw)
    opt = torch.optim.Adam(list(sd.values()), lr=0.0, betas=(0.9, 0.98), eps=1e-9)
    step_no = [0]

    def one_step():
        t0 = time.perf_counter()
        out, ae = om.forward(ob)
        loss = om.loss(ob, out, ae)
        opt.zero_grad()
        loss.backward()
        step_no[0] += 1
        for gparam in opt.param_groups:
            gparam["lr"] = noam_rate(step_no[0], cfg["d_model"], 4000)
        opt.step()
        return time.perf_counter() - t0

    # The reference's own CPU path does not get faster with every core of a 128-core host (intra-op threading of these small
    # matrices: BASELINE.md has 17.8 samples/s on 8 threads): time one step at several thread counts and keep the best, so that
    # the baseline is the CPU path at ITS best setting, not at torch's default of one thread per core.
    nproc = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    cands = sorted({t for t in (8, 16, 32, 64, nproc // 2, default_threads) if 1 <= t <= nproc})
    one_step()                                         # first touch: allocator, lazy initialisation
    sweep = {}
    for t in cands:
        torch.set_num_threads(t)
        one_step()
        sweep[t] = one_step()
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    times = sorted(one_step() for _ in range(steps))
    torch.set_num_threads(default_threads)
    med = times[len(times) // 2]
    return {"value": B / med, "unit": "samples/s", "cores": best, "kind": "port",
            "sample": f"{steps} train steps of the same workload, batch {B}, fp32, dropout off, on the best of {cands} intra-op threads "
                      f"(one step each: {', '.join(f'{t}: {sweep[t]:.2f} s' for t in cands)}); median step {med:.3f} s; nproc={nproc}"}


def decode_cpu_baseline(model, cfg, max_len, beam, live, SOS, UNK, EOS):
    """cpu_baseline leg of bench_decode.py: the same beam search on the CPU oracle (oracle/mtn_oracle.py), one dialogue."""
    from oracle import fixtures as fx
    from oracle.mtn_oracle import OracleConfig, OracleMTN, beam_search
    ocfg = OracleConfig(vocab=cfg["vocab"], n_layers=cfg["N"], d_model=cfg["d_model"], d_ff=cfg["d_ff"], heads=cfg["h"],
                        ft_sizes=tuple(cfg["ft_sizes"]), diff_encoder=True, auto_encoder_ft="query")
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items() if not k.endswith(".pe")}
    om = OracleMTN(ocfg, sd)
    raw = fx.det_batch(cfg["vocab"], 1, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], seed=1, ragged=False)
    ob = fx.oracle_batch(raw)
    # as in cpu_baseline: the CPU path at its best intra-op thread count, not at one thread per core of a 128-core host
    default_threads = torch.get_num_threads()
    times = {}
    with torch.no_grad():
        for t in sorted({t for t in (8, 16, 32, default_threads) if 1 <= t <= (os.cpu_count() or 1)}):
            torch.set_num_threads(t)
            t0 = time.perf_counter()
            beam_search(om, ob, max_len, SOS, UNK, EOS, beam=beam, nbest=beam)
            times[t] = time.perf_counter() - t0
    torch.set_num_threads(default_threads)
    best = min(times, key=times.get)
    t_cpu = times[best]
    return {"value": round(live / t_cpu, 1), "unit": "hypothesis-tokens/s", "cores": best,
            "kind": "port", "sample": f"1 dialogue, same search, fp32 oracle, best of {', '.join(f'{t} threads: {v:.2f} s' for t, v in times.items())}"}


def decode_measure(model, cfg, dev, beam=4, max_len=20, dialogues=4, batch_dialogues=8, cpu_steps=5, use_graph=True, cpu=True):
    """BASELINE configs[4] (cfg5) on the model of the train-step line: beam search exactly as generate.py -> data_utils.py:188-242
    drives it (every step extends `beam` live hypotheses; the loop always runs max_len steps), one synthetic dialogue at a time and
    `batch_dialogues` side by side; greedy; and, as cpu_baseline, the FIRST `cpu_steps` steps of the same search on the CPU oracle
    (a bounded sample: the whole 20-step search takes ~40 s per dialogue on the host).  hypothesis-tokens = decode() calls of the
    reference's loop = 1 + (steps - 1) * beam per dialogue."""
    from mtn_amd.decode import beam_search_decode, beam_search_decode_many, greedy_decode
    from mtn_amd.synthetic import synthetic_batch
    SOS, EOS, UNK, PAD = 2, 3, 0, 1
    was_training = model.training
    model.eval()
    out = {}
    try:
        batches = [synthetic_batch(cfg["vocab"], 1, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev,
                                   seed=100 + i, ragged=False) for i in range(dialogues)]
        run = lambda b: beam_search_decode(model, b, max_len, SOS, UNK, EOS, PAD, beam=beam, nbest=beam, use_graph=use_graph)
        for i in range(3):                                 # warm-up: allocator, the search's graph, and (third dialogue of a shape) the
            run(batches[i % dialogues])                    # capture of the per-dialogue encoder-side pass — the steady state is what is timed
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in batches:
            run(b)
        torch.cuda.synchronize()
        t_beam = time.perf_counter() - t0
        live = 1 + (max_len - 1) * beam
        out["beam"] = {"hypothesis_tokens_per_s": round(dialogues * live / t_beam, 1), "dialogues_per_s": round(dialogues / t_beam, 2),
                       "ms_per_step": round(1e3 * t_beam / dialogues / max_len, 3)}
        for i in range(3):
            greedy_decode(model, batches[i % dialogues], max_len, SOS, PAD, use_graph=use_graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in batches:
            greedy_decode(model, b, max_len, SOS, PAD, use_graph=use_graph)
        torch.cuda.synchronize()
        t_g = time.perf_counter() - t0
        out["greedy"] = {"tokens_per_s": round(dialogues * (max_len - 1) / t_g, 1), "ms_per_step": round(1e3 * t_g / dialogues / (max_len - 1), 3)}
        # two / four dialogues side by side = 8 / 16 hypothesis rows in ONE persistent decode-step launch (16 = the rows of its MFMA tiles)
        for D_, key in ((2, "beam_two_side_by_side"), (4, "beam_four_side_by_side")):
            if D_ * beam > 16:
                continue
            try:
                groups = [synthetic_batch(cfg["vocab"], D_, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev,
                                          seed=200 + 10 * D_ + i, ragged=False) for i in range(max(2, dialogues // D_))]
                many_of = lambda b: beam_search_decode_many(model, b, max_len, SOS, UNK, EOS, PAD, beam=beam, nbest=beam, use_graph=use_graph)
                for i in range(3):
                    many_of(groups[i % len(groups)])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for b in groups:
                    many_of(b)
                torch.cuda.synchronize()
                t_g = time.perf_counter() - t0
                out[key] = {"hypothesis_tokens_per_s": round(D_ * len(groups) * live / t_g, 1), "dialogues_per_s": round(D_ * len(groups) / t_g, 2),
                            "ms_per_step": round(1e3 * t_g / len(groups) / max_len, 3),
                            "what": f"{D_} dialogues per search on the persistent decode-step kernel ({D_ * beam} hypothesis rows per launch)"}
            except Exception as e:  # pragma: no cover
                out[key] = {"error": f"{type(e).__name__}: {e}"}
        if batch_dialogues > 0:
            D = batch_dialogues
            big = synthetic_batch(cfg["vocab"], D, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=300, ragged=True)
            many = lambda: beam_search_decode_many(model, big, max_len, SOS, UNK, EOS, PAD, beam=beam, nbest=beam, use_graph=use_graph)
            many()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                many()
            torch.cuda.synchronize()
            t_many = (time.perf_counter() - t0) / 2
            out["beam_batched"] = {"dialogues_side_by_side": D, "hypothesis_tokens_per_s": round(D * live / t_many, 1),
                                   "dialogues_per_s": round(D / t_many, 2), "ms_per_step": round(1e3 * t_many / max_len, 3)}
        # roofline of a decode step: what it must stream per token is the target stream's weights (per layer: self-attention q|k|v|o,
        # the q and o projections of the five cross-attentions — their memories' K|V are projected once per dialogue —, the FFN) and
        # the generator, in the compute dtype: bytes / step time against HBM.  (The auto-encoder chains run once per dialogue.)
        wbytes = 0
        for layer in model.decoder.layers:
            d = layer.size
            esz = model._flat_lp.element_size()
            n_cross = 3 + len(layer.auto_encoder_attn)
            wbytes += esz * (4 * d * d + n_cross * 2 * d * d + sum(p.numel() for p in (layer.feed_forward.w_1.weight, layer.feed_forward.w_2.weight)))
        wbytes += model._flat_lp.element_size() * model.generator.proj.weight.numel()
        step_ms = out["beam"]["ms_per_step"]
        out["roofline"] = {"bound": "hbm", "what": "weights one decode step streams (target-stream sublayers + generator, compute dtype) / measured "
                                                   "time per step of the one-dialogue beam search.  For <= 16 live hypotheses the step is ONE persistent launch (csrc/decode.hip) whose "
                                                   "92 stages are a chain of all-to-all hand-offs (each >= one fabric round trip): it is bound by that chain's latency, far below the "
                                                   "stream rate — which is why dialogues are batched where throughput matters (beam_batched: the launch path, 8 x beam rows per weight pass)",
                           "weight_bytes_per_step": int(wbytes), "achieved": round(wbytes / (step_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                           "frac": round(wbytes / (step_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 5)}
        widest = out.get("beam_four_side_by_side") or out.get("beam_two_side_by_side") or {}
        if "ms_per_step" in widest:
            rows = (4 if "beam_four_side_by_side" in out and "ms_per_step" in out["beam_four_side_by_side"] else 2) * beam
            out["roofline"]["widest_persistent"] = {"hypothesis_rows_per_launch": rows, "achieved": round(wbytes / (widest["ms_per_step"] * 1e-3) / 1e9, 1),
                                                    "frac": round(wbytes / (widest["ms_per_step"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 5),
                                                    "hypothesis_tokens_per_weight_pass": rows}
        from mtn_amd.decode import MegaDecodeSession
        out["persistent_step_fallbacks"] = MegaDecodeSession.FALLBACKS       # searches re-run on the launch pass after a poll timeout: 0 expected
        if "beam_batched" in out:
            bms = out["beam_batched"]["ms_per_step"]
            out["roofline"]["batched"] = {"dialogues_side_by_side": out["beam_batched"]["dialogues_side_by_side"],
                                          "achieved": round(wbytes / (bms * 1e-3) / 1e9, 1), "frac": round(wbytes / (bms * 1e-3) / 1e9 / PEAK_HBM_GBS, 5),
                                          "hypothesis_tokens_per_weight_pass": out["beam_batched"]["dialogues_side_by_side"] * beam}
        out.update({"what": f"cfg5: beam-{beam} / greedy decode of the cfg2 model, max_len {max_len}, one dialogue at a time "
                            f"({dialogues} dialogues) and {batch_dialogues} side by side; eval mode, same weights as the train step",
                    "unit": "hypothesis-tokens/s", "beam_width": beam, "max_len": max_len, "hip_graph": use_graph})
        if cpu:
            live_cpu = 1 + (cpu_steps - 1) * beam
            out["cpu_baseline"] = decode_cpu_baseline(model, cfg, cpu_steps, beam, live_cpu, SOS, UNK, EOS)
            out["cpu_baseline"]["sample"] = f"first {cpu_steps} steps of the same beam-{beam} search ({live_cpu} decode() calls), " + out["cpu_baseline"]["sample"]
    finally:
        model.train(was_training)
    return out


def hbm_peak_measured(dev):
    """GB/s of a plain 16-byte-per-lane streaming copy (1 GiB read + 1 GiB written: past the 256 MiB Infinity Cache) on this box:
    the measured ceiling beside the 8 TB/s spec for the HBM-bound launches (include/mtn_hip.h mtn_measure_hbm_peak)."""
    import ctypes as C
    from mtn_amd import lib as L
    n = 1 << 30
    src = torch.empty(n, device=dev, dtype=torch.uint8)
    dst = torch.empty(n, device=dev, dtype=torch.uint8)
    src.zero_()
    g = C.c_double(0.0)
    L.check(L.load().mtn_measure_hbm_peak(src.data_ptr(), dst.data_ptr(), n, torch.cuda.current_stream().cuda_stream, C.byref(g)))
    del src, dst
    return round(g.value, 1)


def gemm_census_roofline(step, peak_tflops):
    """Live roofline of the step's dominant kernel family, the grouped MFMA GEMMs.  One eager pass of the step's
    forward+backward is recorded by the library's launch census (include/mtn_hip.h: mtn_census_*), then EVERY recorded GEMM
    launch is re-issued in step order between its own pair of HIP events on the launch stream (median of 5 passes): duration per launch, algorithmic
    FLOPs (2*M*N*K) and algorithmic bytes (operands once + outputs once) per launch, aggregated per kernel.  The dominant
    kernel is the one with the largest summed duration; its average duration is what rocprofv3 --stats reports for the
    same kernel name (profiles/)."""
    import ctypes as C
    from mtn_amd import lib as L
    lib = L.load()
    lib.mtn_census_begin()
    if step._fused():
        step._step_fused()          # the step as captured: the optimiser rides on the parameter-gradient launch (a real update)
    else:
        step._fwd_bwd()
    torch.cuda.synchronize()
    n = lib.mtn_census_end()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # every launch is timed IN STEP ORDER (launch i runs right after launch i-1, as in the step, so its operands are as cold as
    # in the step — re-issuing one launch back-to-back keeps its weights in L2/Infinity Cache and reads ~15 % too fast), one event
    # pair per launch, the median of `reps` passes over the whole sequence
    reps = 5
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    samples = [[] for _ in range(n)]
    for rep in range(reps + 1):
        for i in range(n):
            evs[i][0].record(st)
            L.check(lib.mtn_census_replay(i, 1, st.cuda_stream))
            evs[i][1].record(st)
        torch.cuda.synchronize()
        if rep:                                  # first pass = warm-up
            for i in range(n):
                samples[i].append(evs[i][0].elapsed_time(evs[i][1]) * 1e3)
    # the MEDIAN over the passes: an event pair brackets a host-side launch call, and a host hiccup between the first record and the launch
    # (seen once: 30 ms inside one pair, which made a 24 us kernel the "dominant" one) is idle time, not kernel time
    tot = [sorted(x)[len(x) // 2] * reps for x in samples]
    # an event pair costs a few microseconds of marker hand-over that a kernel between the records only partly hides: time the
    # whole sequence once more between ONE pair (overhead amortised over all launches) and take the per-pair overhead as the
    # difference, so that the per-launch durations add up to the measured duration of the sequence
    seqs = []
    for rep in range(reps):
        evs[0][0].record(st)
        for i in range(n):
            L.check(lib.mtn_census_replay(i, 1, st.cuda_stream))
        evs[0][1].record(st)
        torch.cuda.synchronize()
        seqs.append(evs[0][0].elapsed_time(evs[0][1]) * 1e3)
    seq = sorted(seqs)[len(seqs) // 2]
    overhead = max((sum(tot) / reps - seq) / n, 0.0)
    per = {}
    for i in range(n):
        info = L.CensusLaunch()
        L.check(lib.mtn_census_info(i, C.byref(info)))
        us = max(tot[i] / reps - overhead, 0.5)
        k = per.setdefault(info.variant, {"launches": 0, "flops": 0.0, "bytes": 0.0, "us": 0.0, "workgroups": 0})
        k["launches"] += 1; k["flops"] += info.flops; k["bytes"] += info.bytes; k["us"] += us; k["workgroups"] += info.workgroups
    table = {}
    for v, k in per.items():
        name = lib.mtn_census_variant_name(v).decode()
        table[name] = {"launches_per_step": k["launches"], "avg_us": round(k["us"] / k["launches"], 2),
                       "total_us_per_step": round(k["us"], 1), "avg_workgroups": round(k["workgroups"] / k["launches"]),
                       "gflop_per_launch": round(k["flops"] / k["launches"] / 1e9, 3),
                       "algorithmic_MB_per_launch": round(k["bytes"] / k["launches"] / 1e6, 3),
                       "achieved_TFLOPs": round(k["flops"] / k["us"] / 1e6, 1),
                       "achieved_GBps_algorithmic": round(k["bytes"] / k["us"] / 1e3, 1)}
    dom = max(table, key=lambda nm: table[nm]["total_us_per_step"])
    tot_us = sum(k["us"] for k in per.values())
    tot_fl = sum(k["flops"] for k in per.values())
    return dom, table, {"event_pair_overhead_us": round(overhead, 2), "launches_per_step": n, "total_us_per_step": round(tot_us, 1), "gflop_per_step": round(tot_fl / 1e9, 1),
                        "achieved_TFLOPs": round(tot_fl / tot_us / 1e6, 1), "frac": round(tot_fl / tot_us / 1e6 / peak_tflops, 4)}


def pmc_source():
    """Which committed PMC summary the `traffic` figures come from: file name, its git blob id (sha1 of "blob <len>\\0" + bytes:
    `git hash-object`), and the library source revision it was collected on when the summary records one — so that a stale
    summary (kernels changed, PMC passes not re-run) is visible in the line instead of silently quoted."""
    import glob
    import hashlib
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files:
        return None
    raw = open(files[-1], "rb").read()
    blob = hashlib.sha1(b"blob %d\0" % len(raw) + raw).hexdigest()
    try:
        meta = json.loads(raw).get("collected_on")
    except Exception:
        meta = None
    return {"file": os.path.relpath(files[-1], ROOT), "git_blob": blob, "collected_on": meta,
            "csrc_sha16_now": csrc_sha16(), "stale": (meta or {}).get("csrc_sha16") not in (None, csrc_sha16())}


def csrc_sha16():
    """sha256 (first 16 hex digits) over the kernel sources: identifies the build a profile belongs to."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "mtn_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC passes (profiles/*pmc_traffic.json, written by
    tools/pmc_summary.py from separate FETCH_SIZE / WRITE_SIZE runs with the gfx950 corrections of MI355X_MICROARCH.md)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files:
        return None
    try:
        data = json.load(open(files[-1]))
        for k, v in data.get("families", {}).items():
            if kernel_name.split("<")[0] in k and kernel_name.split("<")[-1].split(">")[0].replace(" ", "") in k.replace(" ", ""):
                return v.get("hbm_bytes_per_launch")
    except Exception:
        return None
    return None


def pmc_step_traffic():
    """Whole-step HBM-side GB from the latest committed PMC summary (profiles/*pmc_traffic.json)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    try:
        return float(json.load(open(files[-1]))["step"]["total_GB"]) if files else None
    except Exception:
        return None


def corpus_loop_measure(model, cfg, dev, videos=24, epochs=2):
    """The REAL loop (a step starts with "batch build on device"; train.py:29-40): `train_step.BucketedTrainer` over a
    synthetic ragged `DeviceCorpus` at AVSD lengths (answers <= 52, questions <= 42 tokens, captions 10-40, histories growing turn by
    turn, 20-40 frames per video), batch 32 planned by `make_batch_indices` exactly as the reference plans its batches, every step =
    device-side batch assembly (two grouped launches refilling the static batch of the padded shape) + one replay of that shape's
    captured train step.  Epoch 0 captures the shapes (untimed); `epochs` further epochs are timed.  Reports samples/s and the
    reference's own unit, target tokens/s (train.py:47-48)."""
    from mtn_amd.data_handler import DeviceCorpus, make_batch_indices
    from mtn_amd.train import synthetic_corpus
    from mtn_amd.train_step import BucketedTrainer
    data = synthetic_corpus(videos, cfg["vocab"], cfg["ft_sizes"], 5, max_answer=52, max_question=42)
    indices, n_samples = make_batch_indices(data, batchsize=32, max_length=256, separate_caption=True)
    corpus = DeviceCorpus(data, dev)
    trainer = BucketedTrainer(model, corpus, cfg["vocab"], pad=1, warmup=4000, bucket=8)
    t_cap = time.perf_counter()
    for idx in indices:
        trainer.step(idx)
    torch.cuda.synchronize()
    t_cap = time.perf_counter() - t_cap
    tok = torch.zeros((), device=dev, dtype=torch.float64)
    samples = 0
    t0 = time.perf_counter()
    for _ in range(epochs):
        for idx in indices:
            _, b = trainer.step(idx)
            tok += b._norms_global[0].double()
            samples += int(idx[-1])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = epochs * len(indices)
    return {"samples_per_s": round(samples / dt, 1), "target_tokens_per_s": round(float(tok.item()) / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4),
            "steps_timed": steps, "batches_per_epoch": len(indices), "dialog_turns": n_samples, "padded_shapes_captured": len(trainer.steps),
            "capture_epoch_s": round(t_cap, 1),
            "what": f"BucketedTrainer over a synthetic ragged DeviceCorpus ({videos} videos x 10 turns, AVSD lengths: answers <= 52, questions <= 42 tokens), batch 32 "
                    "as make_batch_indices plans it (long histories shrink the batch), lengths padded to multiples of 8; every timed step = device-side batch "
                    "assembly + one replay of the padded shape's captured train step (forward, loss, backward, fused Adam); dropout on, bf16"}


def exchange_timeline(step, ref_ms):
    """One more step with HIP events on the exchange chain (dp.ShardedOptimizerSync.timeline): per slice, when its gradients were ready,
    reduce-scattered, updated and all-gathered, relative to the step's start; where the compute chain ended; the step's end."""
    step()
    torch.cuda.synchronize()
    e_start, e_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    step.sharded.timeline = []
    e_start.record()
    step()
    e_end.record()
    torch.cuda.synchronize()
    tl, step.sharded.timeline = step.sharded.timeline, None
    us = lambda e: round(e_start.elapsed_time(e) * 1e3, 1)
    slices = [{"flat_range": [t["lo"], t["hi"]], "MB_reduced": round(t["bytes_reduced"] / 1e6, 1), "MB_gathered": round(t["bytes_gathered"] / 1e6, 1),
               **{k + "_us": us(t[k]) for k in ("ready", "reduced", "update_begins", "updated", "gathered") if k in t}} for t in tl if "lo" in t]
    chain_end = [us(t["chain_end"]) for t in tl if "chain_end" in t]
    return {"what": "this rank, one step, offsets from the step's start (HIP events): a slice's gradients ready (its segment graph done, "
                    "reduce-scatter issued) -> reduce-scatter done (stamped on a separate stream) -> the compute stream, one segment later, "
                    "begins Adam on the shard -> done, all-gather issued -> all-gather done (stamped); compute_chain_end = the last segment "
                    "graph finished, step_end = the exchange drained and the copies refreshed.  exposed = step_end - compute_chain_end (the "
                    "last slices' update + gather); the chain's stretch = compute_chain_end / the exchange-free single-rank step of the "
                    "same batch (optimiser inside its dW launch): it contains the shard updates of all but the last slices",
            "slices": slices, "compute_chain_end_us": chain_end[0] if chain_end else None, "step_end_us": us(e_end),
            "exposed_exchange_us": round(us(e_end) - chain_end[0], 1) if chain_end else None,
            "chain_stretch_vs_single_rank_step": round(chain_end[0] / (ref_ms * 1e3), 3) if (chain_end and ref_ms) else None}


def dp_schedule_one_rank(args):
    """The data-parallel schedule of an N-GPU job on ONE rank with every collective through RCCL, measured in a FRESH process
    (`bench.py --dp-one-rank-probe`): the process group is created first, the model and its flat buffers afterwards — the order
    of a real multi-rank job (main() below).  (Creating the one-rank group late, inside this process, next to the graphs and
    buffers of the main measurement, made every collective ~0.7 ms slower: 14.8 instead of 5.1 ms per step, r03 profiles.)"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MTN_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.abspath(__file__), "--dp-one-rank-probe", "--steps", str(args.steps), "--workload", args.workload,
           "--dtype", args.dtype, "--dropout", str(args.dropout)] + (["--no-graph"] if args.no_graph else []) + \
          (["--batch-per-gpu", str(args.batch_per_gpu)] if args.batch_per_gpu else [])
    def child(env, extra=()):
        out = subprocess.run(cmd + list(extra), env=env, capture_output=True, text=True, timeout=240)
        for ln in reversed(out.stdout.strip().splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return {"error": (out.stderr or out.stdout)[-400:]}

    try:
        res = child(env)
        # the same schedule with the update of a rank of 8: Adam on 1/8 of every slice (the parameters come out WRONG: a timing probe,
        # `--no-record`); the one-rank collectives move nothing, so this is a rank's own cost in an 8-GPU job without any link time
        r8 = child(dict(env, MTN_DP_EMULATE_WORLD="8"), ("--no-record",))
        if "ms_per_step" in r8 and "ms_per_step" in res:
            res["ms_per_step_update_of_a_rank_of_8"] = r8["ms_per_step"]
            res["update_of_a_rank_of_8_what"] = ("same probe, Adam on 1/8 of every slice as on a rank of an 8-GPU job (results wrong: timing only); "
                                                 "collectives still one-rank: no link time in it")
        return res
    except Exception as e:  # pragma: no cover
        return {"error": f"{type(e).__name__}: {e}"}


def dp_one_rank_probe(args):
    """Child of dp_schedule_one_rank(): prints one JSON object."""
    import torch.distributed as dist
    from mtn_amd import dp, lib, make_model
    from mtn_amd.synthetic import CONFIGS, synthetic_batch
    from mtn_amd.train_step import TrainStep
    dp.init_distributed(force=True)
    dp.ALLOW_EMULATION = bool(args.no_record)
    emulating = args.no_record and os.environ.get("MTN_DP_EMULATE_WORLD", "1") not in ("", "0", "1")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    lib.load()
    cfg = dict(CONFIGS[args.workload])
    B = args.batch_per_gpu or cfg["B"]
    lp = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    model = make_model(cfg["vocab"], cfg["vocab"], N=cfg["N"], d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"], dropout=args.dropout,
                       ft_sizes=cfg["ft_sizes"], diff_encoder=True, diff_embed=False, diff_gen=False, auto_encoder_ft="query", compute_dtype=lp,
                       attn_dropout=0.1 if args.dropout > 0 else 0.0).to(dev).train()
    model.prepare()
    sync = dp.GradSync(lambda: model.flat_buffers()[2], force=True)
    sync.broadcast_(model._flat)
    batch = synthetic_batch(cfg["vocab"], B, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=1, ragged=False)

    def timed(st):
        for _ in range(3):
            st()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            st()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps * 1e3

    st = TrainStep(model, batch, cfg["vocab"], pad=1, warmup=4000, grad_sync=sync, use_graph=not args.no_graph)
    ms = timed(st)
    sh = st.sharded
    out = {"ms_per_step": round(ms, 4), "samples_per_s": round(B / ms * 1e3, 1), "rccl_ranks": dist.get_world_size(), "dist_backend": dist.get_backend(),
           "schedule": ("layer-segmented backward (N+3 hipGraphs), " if st.overlap else "two-graph schedule, ") +
                       ("reduce-scatter -> Adam on the shard (compute stream, one segment later) -> all-gather, per slice" if sh is not None else "all-reduce per slice + full Adam"),
           "collectives_issued": dict(sh.calls) if sh is not None else {}, "slices_per_step": len(st._slices()), "batch_per_gpu": B,
           "what": "the step of a data-parallel rank on this one GPU in a fresh process, every collective through a one-rank RCCL group "
                   "(shard = the whole slice): the schedule's cost without peers (segmentation, the separate optimiser pass, RCCL "
                   "calls); same batch as `value`"}
    if sh is not None and not emulating:     # the same schedule with the collectives skipped: what the RCCL calls themselves cost
        st2 = TrainStep(model, batch, cfg["vocab"], pad=1, warmup=4000, grad_sync=sync, use_graph=not args.no_graph)
        st2.sharded.collective = False
        out["ms_per_step_collectives_skipped"] = round(timed(st2), 4)
        st2.sharded.collective = True
        if st.overlap and not args.no_graph:
            try:
                out["timeline"] = exchange_timeline(st, None)
            except Exception as e:  # pragma: no cover
                out["timeline"] = {"error": str(e)}
    import ctypes
    ctypes.CDLL(None).fflush(None)
    print(json.dumps(out), flush=True)
    dist.destroy_process_group()


KERNEL_COLS = ["launches_per_step", "avg_us", "achieved_TFLOPs", "achieved_GBps_algorithmic", "avg_workgroups"]


def _kernels_terse(table):
    """{kernel: [launches, avg us, TFLOP/s, algorithmic GB/s, avg workgroups]} (columns: KERNEL_COLS), largest summed duration first"""
    order = sorted(table, key=lambda k: -table[k]["total_us_per_step"])
    return {k: [table[k][c] for c in KERNEL_COLS] for k in order}


def terse_line(full):
    """The one JSON line the driver records keeps only an 8 KB tail: this is the verbose record with the prose removed (it lives in
    README.md, "Reading the bench line") and the tables packed, < 7 KB, the contract's keys first.  `--full-record PATH` writes the verbose form."""
    g = lambda d, *ks: (g(d.get(ks[0]), *ks[1:]) if (isinstance(d, dict) and len(ks) > 1) else (d.get(ks[0]) if isinstance(d, dict) else None))
    keep = lambda d, ks: {k: d[k] for k in ks if isinstance(d, dict) and d.get(k) is not None}
    line = {k: full[k] for k in ("metric", "value", "unit", "record", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data") if k in full}
    cfg = full["config"]
    line["config"] = keep(cfg, ["workload", "batch_per_gpu", "global_batch", "parallelism", "dropout", "attn_dropout", "hip_graph", "rccl_ranks",
                                "dist_backend", "window_ms_per_step", "median_window_ms_per_step", "normalised_loss_last_step"])
    r = full["roofline"]
    roof = keep(r, ["bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launches_per_step", "avg_us_per_launch", "gflop_per_launch",
                    "algorithmic_bytes_per_launch", "achieved_TFLOPs", "peak_measured", "frac_of_measured_peak", "peak_guide_copy", "frac_of_guide_copy", "error"])
    ts = r.get("traffic_source") or {}
    roof["traffic_source"] = {"file": ts.get("file"), "stale": ts.get("stale")}
    if r.get("next_kernel"):
        roof["next_kernel"] = keep(r["next_kernel"], ["bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launches_per_step", "avg_us_per_launch",
                                                      "algorithmic_bytes_per_launch", "frac_of_measured_peak"])
    st = r.get("step") or {}
    roof["step"] = keep(st, ["achieved_TFLOPs", "frac", "frac_of_measured_peak", "gflop_per_sample", "hip_event_ms_per_step", "optimiser_bytes_per_step",
                             "hbm_GB_per_step_pmc", "hbm_frac_of_8TBps"])
    if st.get("mfma_peak_measured_by_shape"):
        roof["step"]["mfma_peak_measured"] = {"16x16x32": st["mfma_peak_measured_by_shape"].get("v_mfma_f32_16x16x32_bf16"),
                                              "32x32x16": st["mfma_peak_measured_by_shape"].get("v_mfma_f32_32x32x16_bf16")}
    if r.get("all_gemm_kernels"):
        roof["all_gemm_kernels"] = keep(r["all_gemm_kernels"], ["launches_per_step", "total_us_per_step", "gflop_per_step", "achieved_TFLOPs", "frac"])
    if r.get("kernels"):
        roof["kernel_cols"] = KERNEL_COLS
        roof["kernels"] = _kernels_terse(r["kernels"])
    line["roofline"] = roof
    if "cpu_baseline" in full:
        cb = dict(full["cpu_baseline"])
        if isinstance(cb.get("value"), float):
            cb["value"] = round(cb["value"], 2)
        line["cpu_baseline"] = cb
    # secondary figures LAST (the driver's record keeps the tail of the line): batch 64 first, then DP, sweep, cfg4, corpus loop, decode
    sec = cfg.get("secondary") or {}
    out = {}
    b64 = sec.get("batch64_one_gpu")
    if b64:
        stp = g(b64, "roofline", "step") or {}
        out["batch64_one_gpu"] = {**keep(b64, ["samples_per_s", "ms_per_step"]),
                                  **{k: stp.get(k) for k in ("achieved_TFLOPs", "frac", "frac_of_measured_peak") if stp.get(k) is not None},
                                  "dominant_kernel": g(b64, "roofline", "dominant_kernel"),
                                  "all_gemm_kernels": keep(g(b64, "roofline", "all_gemm_kernels") or {}, ["launches_per_step", "total_us_per_step", "achieved_TFLOPs"]),
                                  "kernels": _kernels_terse(g(b64, "roofline", "kernels") or {})}
    dpr = sec.get("dp_schedule_one_rank")
    if dpr:
        out["dp_schedule_one_rank"] = {**keep(dpr, ["ms_per_step", "samples_per_s", "ms_per_step_collectives_skipped", "ms_per_step_update_of_a_rank_of_8",
                                                    "rccl_ranks", "dist_backend", "slices_per_step", "collectives_issued", "error"]),
                                       **keep(dpr.get("timeline") or {}, ["compute_chain_end_us", "step_end_us", "exposed_exchange_us"])}
    if sec.get("batch_sweep"):
        out["batch_sweep"] = {b: [v["samples_per_s"], v["ms_per_step"], v["step_TFLOPs"], v["frac_of_2.5PF"]] for b, v in sec["batch_sweep"]["batches"].items()}
        out["batch_sweep_cols"] = ["samples_per_s", "ms_per_step", "step_TFLOPs", "frac_of_2.5PF"]
    if sec.get("cfg4_long_context"):
        out["cfg4_long_context"] = keep(sec["cfg4_long_context"], ["samples_per_s", "ms_per_step", "step_TFLOPs"])
    if sec.get("corpus_loop"):
        out["corpus_loop"] = keep(sec["corpus_loop"], ["samples_per_s", "target_tokens_per_s", "ms_per_step", "steps_timed", "padded_shapes_captured"])
    dec = sec.get("decode")
    if dec:
        d_ = {k: keep(dec[k], ["hypothesis_tokens_per_s", "tokens_per_s", "dialogues_per_s", "ms_per_step", "dialogues_side_by_side", "error"])
              for k in ("beam", "greedy", "beam_two_side_by_side", "beam_four_side_by_side", "beam_batched") if k in dec}
        d_["roofline"] = keep(dec.get("roofline") or {}, ["bound", "weight_bytes_per_step", "achieved", "peak", "unit", "frac"])
        if g(dec, "roofline", "widest_persistent"):
            d_["roofline"]["widest_persistent"] = dec["roofline"]["widest_persistent"]
        if g(dec, "roofline", "batched"):
            d_["roofline"]["batched"] = keep(dec["roofline"]["batched"], ["achieved", "frac", "hypothesis_tokens_per_weight_pass"])
        d_.update(keep(dec, ["beam_width", "max_len", "persistent_step_fallbacks"]))
        if dec.get("cpu_baseline"):
            d_["cpu_baseline"] = keep(dec["cpu_baseline"], ["value", "unit", "cores", "kind"])
        out["decode"] = d_
    for k in ("one_rank_no_exchange",):
        if sec.get(k):
            out[k] = keep(sec[k], ["samples_per_s", "ms_per_step"])
    ex = sec.get("exchange")
    if ex:
        tl = ex.get("timeline") or {}
        out["exchange"] = {**keep(ex, ["step_minus_one_rank_no_exchange_ms", "bytes_on_the_links_per_rank_per_step", "scheme", "collectives_per_step"]),
                           "timeline": {**keep(tl, ["compute_chain_end_us", "step_end_us", "exposed_exchange_us", "chain_stretch_vs_single_rank_step"]),
                                        "slices_us": [[sl.get(k + "_us") for k in ("ready", "reduced", "update_begins", "updated", "gathered")] for sl in tl.get("slices", [])],
                                        "slices_cols": ["ready", "reduced", "update_begins", "updated", "gathered"]}}
    if sec.get("error"):
        out["error"] = sec["error"]
    line["secondary"] = out
    return line


def main():
    args = parse()
    if args.dp_one_rank_probe:
        return dp_one_rank_probe(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    from mtn_amd import dp, lib, make_model
    from mtn_amd.synthetic import CONFIGS, flops_per_sample, synthetic_batch
    from mtn_amd.train_step import TrainStep

    if args.bf16_grad_allreduce:
        os.environ["MTN_DP_SHARDED"] = "0"     # bf16 compression exists for the all-reduce scheme only (the sharded optimiser reduce-scatters fp32 in place)
    if args.no_record:
        dp.ALLOW_EMULATION = True
    elif os.environ.get("MTN_DP_EMULATE_WORLD", "1") not in ("", "0", "1"):
        sys.exit("bench.py: MTN_DP_EMULATE_WORLD produces wrong parameters (timing probe): pass --no-record to run it; such a line "
                 "is marked \"record\": false and must not be quoted")
    rank, world, local = dp.init_distributed()
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    local = local % torch.cuda.device_count()      # (several ranks may share a GPU in a gloo dry run of the multi-rank control flow)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    lib.load()
    cfg = dict(CONFIGS[args.workload])
    B = args.batch_per_gpu or cfg["B"]
    lp = torch.bfloat16 if args.dtype == "bf16" else torch.float32

    torch.manual_seed(0)                       # identical replicas: same seed -> same make_model init on every rank
    model = make_model(cfg["vocab"], cfg["vocab"], N=cfg["N"], d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"],
                       dropout=args.dropout, ft_sizes=cfg["ft_sizes"], diff_encoder=True, diff_embed=False, diff_gen=False,
                       auto_encoder_ft="query", compute_dtype=lp, attn_dropout=0.1 if args.dropout > 0 else 0.0)
    model.to(dev).train()
    model.prepare()
    sync = None
    if world > 1 or os.environ.get("MTN_FORCE_DIST") == "1":
        sync = dp.GradSync(lambda: model.flat_buffers()[2], compress_bf16=args.bf16_grad_allreduce)
        sync.broadcast_(model._flat)
        model._flat_version = -1
        model.prepare()
    batch = synthetic_batch(cfg["vocab"], B, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"],
                            device=dev, seed=1 + rank, ragged=False)
    step = TrainStep(model, batch, cfg["vocab"], pad=1, warmup=4000, grad_sync=sync, use_graph=not args.no_graph)

    def timed_window(st, n_steps):
        """n_steps steps between two (barrier + synchronize) brackets; wall seconds, max over ranks; HIP-event ms of this rank."""
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        out = None
        for _ in range(n_steps):
            out = st()
        e1.record()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        if world > 1:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item()), e0.elapsed_time(e1), out

    for _ in range(max(1, args.warmup)):
        loss = step()
    elapsed, ev_total_ms, loss = timed_window(step, args.steps)              # <- `value`
    loss_val = float(loss.item())            # TrainStep returns the normalised loss: sum of the three KL terms per target / query token
    windows = [elapsed / args.steps * 1e3]
    for _ in range(max(0, args.windows)):
        windows.append(timed_window(step, args.steps)[0] / args.steps * 1e3)
    # secondary measurements (never `value`)
    secondary = {}
    if not args.no_secondary and args.workload in ("cfg2", "cfg3"):
        try:
            if world == 1 and B != 64:
                # BASELINE configs[2] / north_star quote the utilisation target at batch 64 per GPU: the same step at that batch
                b64 = synthetic_batch(cfg["vocab"], 64, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=7, ragged=False)
                st64 = TrainStep(model, b64, cfg["vocab"], pad=1, warmup=4000, grad_sync=None, use_graph=not args.no_graph)
                for _ in range(3):
                    st64()
                dt64, ev64, _ = timed_window(st64, args.steps)
                sps64 = 64 * args.steps / dt64
                tf64 = sps64 * SURVEY_GFLOP_PER_SAMPLE["cfg3"] / 1e3
                peak64 = PEAK_BF16_TFLOPS if lp == torch.bfloat16 else PEAK_FP32_TFLOPS
                opt_bytes = (26 if st64._fused() else 38) * sum(p.numel() for p in model.parameters())
                r64 = {"step": {"what": "whole captured train step at batch 64 — the batch north_star states its MFMA-utilisation target on: "
                                        "15.90 GFLOP/sample x samples/s against the dense bf16 MFMA peak; optimiser bytes / step time against HBM",
                                "achieved_TFLOPs": round(tf64, 2), "peak": peak64, "frac": round(tf64 / peak64, 5),
                                "hip_event_ms_per_step": round(ev64 / args.steps, 4),
                                "optimiser_bytes_per_step": opt_bytes,
                                "optimiser_GBps_over_whole_step": round(opt_bytes / (dt64 / args.steps) / 1e9, 1)}}
                try:
                    dom64, table64, allg64 = gemm_census_roofline(st64, peak64)
                    r64.update({"dominant_kernel": dom64, "all_gemm_kernels": allg64, "kernels": table64})
                except Exception as e:  # pragma: no cover
                    r64["error"] = str(e)
                secondary["batch64_one_gpu"] = {"samples_per_s": round(sps64, 1), "ms_per_step": round(dt64 / args.steps * 1e3, 4),
                                                "what": "same model and step at batch 64 (BASELINE configs[2] per-GPU batch) on this one GPU",
                                                "roofline": r64}
                del st64, b64
            if world == 1:
                # the data-parallel schedule of an N-GPU job on this ONE rank, every collective through RCCL (a one-rank "nccl"
                # group): layer-segmented backward, per-slice reduce-scatter -> Adam on the shard -> all-gather on side streams
                # (dp.ShardedOptimizerSync).  What a rank of the scaling run computes besides waiting for its peers.
                secondary["dp_schedule_one_rank"] = dp_schedule_one_rank(args)
                # BASELINE configs[4]: decode on the same weights (never `value`)
                secondary["decode"] = decode_measure(model, cfg, dev, use_graph=not args.no_graph, cpu=not args.no_cpu_baseline)
                # where this design saturates: the same captured step at growing batch (samples/s, whole-step TFLOP/s against 2.5 PF) — the
                # evidence that separates "the problem is too small for the chip" from "the kernels are slow"
                sweep = {}
                peak_sw = PEAK_BF16_TFLOPS if lp == torch.bfloat16 else PEAK_FP32_TFLOPS
                for Bs in (128, 256):
                    bb = synthetic_batch(cfg["vocab"], Bs, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=11, ragged=False)
                    stb = TrainStep(model, bb, cfg["vocab"], pad=1, warmup=4000, grad_sync=None, use_graph=not args.no_graph)
                    for _ in range(3):
                        stb()
                    dtb, _, _ = timed_window(stb, max(5, args.steps // 2))
                    spsb = Bs * max(5, args.steps // 2) / dtb
                    sweep[str(Bs)] = {"samples_per_s": round(spsb, 1), "ms_per_step": round(dtb / max(5, args.steps // 2) * 1e3, 4),
                                      "step_TFLOPs": round(spsb * SURVEY_GFLOP_PER_SAMPLE["cfg2"] / 1e3, 1),
                                      "frac_of_2.5PF": round(spsb * SURVEY_GFLOP_PER_SAMPLE["cfg2"] / 1e3 / peak_sw, 4)}
                    del stb, bb
                secondary["batch_sweep"] = {"what": "same model, same captured step, batch per GPU 32 (= `value`) / 64 (= batch64_one_gpu) / 128 / 256: 15.90 GFLOP/sample x "
                                                    "samples/s against the dense bf16 MFMA peak", "batches": sweep}
                # BASELINE configs[3] (cfg4: 512-token history, 256 + 256 frames, batch 8) on the same model
                from mtn_amd.synthetic import CONFIGS as _C
                c4 = dict(_C["cfg4"])
                b4 = synthetic_batch(c4["vocab"], c4["B"], c4["Q"], c4["H"], c4["C"], c4["T"], c4["frames"], c4["ft_sizes"], device=dev, seed=13, ragged=False)
                st4 = TrainStep(model, b4, c4["vocab"], pad=1, warmup=4000, grad_sync=None, use_graph=not args.no_graph)
                for _ in range(3):
                    st4()
                dt4, _, _ = timed_window(st4, args.steps)
                secondary["cfg4_long_context"] = {"samples_per_s": round(c4["B"] * args.steps / dt4, 1), "ms_per_step": round(dt4 / args.steps * 1e3, 4),
                                                  "step_TFLOPs": round(c4["B"] * args.steps / dt4 * SURVEY_GFLOP_PER_SAMPLE["cfg4"] / 1e3, 1),
                                                  "what": f"BASELINE configs[3]: batch {c4['B']}, history {c4['H']} tokens, frames {c4['frames']}, same model and captured step"}
                del st4, b4
                # the real loop: ragged corpus, per-step device batch assembly, one captured graph per padded shape
                secondary["corpus_loop"] = corpus_loop_measure(model, cfg, dev)
            if world > 1:
                # the same per-GPU batch on ONE rank without any exchange (graph of forward+backward+optimiser epilogue): the
                # figure a DP rank is compared with, measured in the same job; the other ranks idle at the barriers
                st1 = TrainStep(model, batch, cfg["vocab"], pad=1, warmup=4000, grad_sync=None, use_graph=not args.no_graph) if rank == 0 else None
                if rank == 0:
                    for _ in range(3):
                        st1()
                dt1, _, _ = timed_window(st1 if rank == 0 else (lambda: None), args.steps)
                secondary["one_rank_no_exchange"] = {"samples_per_s": round(B * args.steps / dt1, 1), "ms_per_step": round(dt1 / args.steps * 1e3, 4),
                                                     "what": "rank 0 alone, same per-GPU batch, no gradient exchange, optimiser fused into the dW launch"}
                # what data parallelism adds to a rank's step: everything that is not the single-rank step of the same batch — the
                # schedule's own cost (layer-segmented graphs, separate shard update: bench.py --gpus 1 reports it as
                # secondary.dp_schedule_one_rank) plus the exchange that backward does not hide
                nparams = sum(p.numel() for p in model.parameters())
                lp_gather = step.sharded is not None and step.sharded.lp_mode()
                timeline = exchange_timeline(step, dt1 / args.steps * 1e3) if (step.sharded is not None and step.overlap) else None
                secondary["exchange"] = {
                    "step_minus_one_rank_no_exchange_ms": round(elapsed / args.steps * 1e3 - dt1 / args.steps * 1e3, 4),
                    "bytes_on_the_links_per_rank_per_step": int(nparams * (4 + (2 if lp_gather else 4)) * (world - 1) / world),
                    "scheme": ("reduce-scatter fp32 gradients + all-gather " + ("bf16 weight copies (matrices) / fp32 (glue, vectors)" if lp_gather else "fp32 masters")
                               if step.sharded is not None else "all-reduce fp32 gradients"),
                    "collectives_per_step": {k: round(v / max(1, args.steps + args.warmup + args.steps * args.windows), 1) for k, v in step.sharded.calls.items()} if step.sharded is not None else None,
                    "timeline": timeline}
        except Exception as e:  # pragma: no cover
            secondary["error"] = str(e)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        gflop_formula = flops_per_sample(**cfg) / 1e9
        gflop = SURVEY_GFLOP_PER_SAMPLE.get(args.workload, round(gflop_formula, 2))
        step_tf = value * gflop / 1e3                                 # whole job
        peak = (PEAK_BF16_TFLOPS if lp == torch.bfloat16 else PEAK_FP32_TFLOPS) * world
        ev_ms = ev_total_ms / args.steps
        step_info = {"what": "whole captured train step (one hipGraph launch = one step): algorithmic GFLOP/sample x samples "
                             "per launch / HIP-event time per launch on the launch stream",
                     "achieved_TFLOPs": round(step_tf, 2), "frac": round(step_tf / peak, 5),
                     "gflop_per_sample": gflop, "gflop_per_sample_closed_form": round(gflop_formula, 3),
                     "hip_event_ms_per_step": round(ev_ms, 4),
                     # fused: p, m, v read + written, the bf16 copy written (the transposed copy only for the W_o matrices);
                     # separate pass: + the gradient written and read back, + every transposed copy read and written
                     "optimiser_bytes_per_step": (26 if step._fused() else 38) * sum(p.numel() for p in model.parameters())}
        hbm = pmc_step_traffic()
        if hbm is not None and world == 1 and args.workload == "cfg2" and B == 32:
            # whole-step HBM-side traffic from the committed PMC passes (FETCH_SIZE + WRITE_SIZE, gfx950 corrections) / measured step time
            step_info["hbm_GB_per_step_pmc"] = hbm
            step_info["hbm_GBps_measured"] = round(hbm / (ms * 1e-3), 1)
            step_info["hbm_frac_of_8TBps"] = round(hbm / (ms * 1e-3) / PEAK_HBM_GBS, 4)
        try:
            import ctypes as C
            scratch = torch.empty(2048 * 256, device=dev, dtype=torch.float32)
            tf16, tf32 = C.c_double(0.0), C.c_double(0.0)
            lib.check(lib.load().mtn_measure_mfma_peak_shapes(20000, scratch.data_ptr(), torch.cuda.current_stream().cuda_stream, C.byref(tf16), C.byref(tf32)))
            tf = C.c_double(max(tf16.value, tf32.value))
            step_info["mfma_peak_measured_TFLOPs"] = round(tf.value, 1)
            step_info["mfma_peak_measured_by_shape"] = {"v_mfma_f32_16x16x32_bf16": round(tf16.value, 1), "v_mfma_f32_32x32x16_bf16": round(tf32.value, 1),
                                                        "guide_32x32x16": 2495.0,
                                                        "note": "register-only issue on independent accumulator chains, this box, this run; `frac` is against the 2.5 PFLOP/s "
                                                                "spec, `frac_of_measured_peak` against the better of the two shapes"}
            step_info["frac_of_measured_peak"] = round(step_tf / world / tf.value, 5) if lp == torch.bfloat16 else None
            b64s = secondary.get("batch64_one_gpu", {}).get("roofline", {}).get("step")
            if b64s is not None and lp == torch.bfloat16:
                b64s["mfma_peak_measured_TFLOPs"] = round(tf.value, 1)
                b64s["frac_of_measured_peak"] = round(b64s["achieved_TFLOPs"] / tf.value, 5)
        except Exception as e:  # pragma: no cover
            step_info["mfma_peak_measured_TFLOPs"] = None
        try:
            dom, table, allg = gemm_census_roofline(step, peak / world)

            def mfma_roof(name):
                d = table[name]
                return {"bound": "mfma", "kernel": name, "achieved": d["achieved_TFLOPs"], "peak": peak / world, "unit": "TFLOP/s",
                        "frac": round(d["achieved_TFLOPs"] / (peak / world), 4), "traffic": pmc_traffic(name),
                        "launches_per_step": d["launches_per_step"], "avg_us_per_launch": d["avg_us"],
                        "gflop_per_launch": d["gflop_per_launch"], "algorithmic_bytes_per_launch": int(d["algorithmic_MB_per_launch"] * 1e6),
                        "peak_measured": step_info.get("mfma_peak_measured_TFLOPs"),
                        "frac_of_measured_peak": (round(d["achieved_TFLOPs"] / step_info["mfma_peak_measured_TFLOPs"], 4)
                                                  if step_info.get("mfma_peak_measured_TFLOPs") and lp == torch.bfloat16 else None)}

            try:
                hbm_peak = hbm_peak_measured(dev)
            except Exception:  # pragma: no cover
                hbm_peak = None
            TABLE = "gemm_tt_dma128_table_kernel"
            if dom == TABLE:
                # the parameter-gradient + optimiser launch: 26-28 B of parameter streams per weight behind every 2*K flops -> HBM-bound
                d = table[dom]
                roof = {"bound": "hbm", "kernel": dom, "achieved": d["achieved_GBps_algorithmic"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(d["achieved_GBps_algorithmic"] / PEAK_HBM_GBS, 4), "traffic": pmc_traffic(dom),
                        "launches_per_step": d["launches_per_step"], "avg_us_per_launch": d["avg_us"],
                        "gflop_per_launch": d["gflop_per_launch"], "algorithmic_bytes_per_launch": int(d["algorithmic_MB_per_launch"] * 1e6),
                        "achieved_TFLOPs": d["achieved_TFLOPs"],
                        "peak_measured": hbm_peak, "peak_guide_copy": 6290.0,
                        "frac_of_measured_peak": round(d["achieved_GBps_algorithmic"] / hbm_peak, 4) if hbm_peak else None,
                        "frac_of_guide_copy": round(d["achieved_GBps_algorithmic"] / 6290.0, 4),
                        "peak_measured_what": "peak_guide_copy = the 6.29 TB/s float4 copy of MI355X_MICROARCH.md; peak_measured = GB/s of a 16-byte-per-lane streaming copy (1 GiB read + 1 GiB written) on this box: best over non-temporal / plain accesses, 1 / 4 loads in flight per thread, 8-32 workgroups per CU"}
                second = max((n for n in table if n != TABLE), key=lambda nm: table[nm]["total_us_per_step"])
                roof["next_kernel"] = mfma_roof(second)
            else:
                roof = mfma_roof(dom)
            roof["what"] = ("dominant kernel of the step (largest summed duration): algorithmic bytes (operands once + per weight the "
                            "optimiser epilogue's 26 B: p, m, v read and written, the bf16 copy written; + 2 B for the W_o matrices' transposed copy) or FLOPs per launch / HIP-event "
                            "duration per launch, averaged over all of its launches in one step (library launch census, each "
                            "launch replayed in step order, one HIP-event pair each, median of 5 passes); traffic = HBM bytes per launch from the "
                            "committed rocprofv3 PMC passes")
            roof["traffic_source"] = pmc_source()
            if roof["traffic_source"] and roof["traffic_source"].get("stale"):
                print("bench.py: WARNING: roofline.traffic comes from PMC passes collected on OTHER kernel sources (" +
                      str((roof["traffic_source"].get("collected_on") or {}).get("csrc_sha16")) + " vs " + roof["traffic_source"]["csrc_sha16_now"] +
                      "): re-run tools/prof_round.sh and commit profiles/*_pmc_traffic.json", file=sys.stderr, flush=True)
            roof.update({"all_gemm_kernels": allg, "kernels": table, "step": step_info})
        except Exception as e:  # pragma: no cover
            roof = {"bound": "mfma", "achieved": round(step_tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(step_tf / peak, 5),
                    "traffic": None, "error": str(e), "step": step_info}
        line = {"metric": "train-step samples/sec (d_model=512, 6L MTN)", "value": round(value, 2), "unit": "samples/s",
                **({"record": False} if args.no_record else {}),
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": f"{args.workload}: d_model={cfg['d_model']} N={cfg['N']} h={cfg['h']} d_ff={cfg['d_ff']} |V|={cfg['vocab']} "
                                       f"Q/H/C/T={cfg['Q']}/{cfg['H']}/{cfg['C']}/{cfg['T']} frames={cfg['frames']} ft={cfg['ft_sizes']}",
                           "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}",
                           "dropout": args.dropout, "attn_dropout": 0.1 if args.dropout > 0 else 0.0,
                           "hip_graph": not args.no_graph, "weights": "random-init (xavier), fp32 master + bf16 compute copy",
                           "rccl_ranks": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
                           "dist_backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else None,
                           "window_ms_per_step": [round(w, 4) for w in windows],
                           "median_window_ms_per_step": round(sorted(windows)[len(windows) // 2], 4),
                           "normalised_loss_last_step": round(loss_val, 4),
                           **({"per_gpu_work": f"{B} samples per GPU at every N > 1 (BASELINE configs[2] = cfg3: 64 per GPU); the N = 1 default line is "
                                               "configs[1] (cfg2, batch 32): the single-GPU figure of THIS per-GPU batch is secondary.one_rank_no_exchange "
                                               "(same job) / the N = 1 line's secondary.batch64_one_gpu"} if world > 1 else {}),
                           "secondary": secondary},
                "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(model, cfg, B, args.cpu_steps)
            except Exception as e:  # pragma: no cover
                line["cpu_baseline"] = {"error": str(e)}
    if args.pmc_calibration and rank == 0:
        from mtn_amd import lib as L
        n = 1 << 24
        p_, g_, m_, v_ = (torch.full((n,), x, device=dev) for x in (0.5, 1e-3, 0.0, 0.0))
        lp_ = torch.zeros(n, device=dev, dtype=torch.bfloat16)
        st_ = torch.tensor([1.0, 1e-3, 0.1, 0.02, 0, 0, 0, 0], device=dev)           # step, lr, 1 - beta1^t, 1 - beta2^t
        L.check(L.load().mtn_adam_step(L.dtype_code(torch.bfloat16), n, p_.data_ptr(), g_.data_ptr(), m_.data_ptr(), v_.data_ptr(),
                                       lp_.data_ptr(), st_.data_ptr(), None, 0.9, 0.98, 1e-9, L.stream_ptr()))
        torch.cuda.synchronize()
    # RCCL writes its version banner through C stdio, which is block-buffered on a pipe and would surface after our line at
    # exit: flush every rank's C buffers first, so that the JSON line is the last thing the job prints
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    if world > 1:
        torch.distributed.barrier()
    if rank == 0:
        if args.full_record:
            with open(args.full_record, "w") as f:
                json.dump(line, f, indent=1)
        print(json.dumps(terse_line(line), separators=(",", ":")), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
