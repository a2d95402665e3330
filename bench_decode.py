#!/usr/bin/env python
"""bench_decode.py — decode-path throughput (BASELINE.json configs[4]: greedy / beam-4 decode, 1 GPU) next to the CPU oracle.

    python bench_decode.py [--beam 4] [--max-len 20] [--dialogues 8] [--dtype bf16]

Workload: the cfg2 model (d_model=512, 6 layers, 8 heads, |V|=3000; random init, eval mode), one synthetic dialogue at a
time (Q/H/C = 20/128/40 tokens, 32 I3D + 32 VGGish frames), beam search exactly as data_utils.py:188-242 drives it
(every step extends `beam` live hypotheses; the loop always runs max_len steps).  Reported: generated tokens/s =
dialogues x max_len x live hypotheses / wall time (hypothesis-tokens, what the reference's per-hypothesis decode() calls
count), dialogues/s, ms per decode step; the same search on the CPU oracle on a bounded sample beside it.  One JSON line.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--beam", type=int, default=4)
    ap.add_argument("--max-len", type=int, default=20)
    ap.add_argument("--dialogues", type=int, default=8)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch-dialogues", type=int, default=8, help="also time D dialogues decoded side by side (0 = skip)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--kv-cache", default="auto", choices=["auto", "on", "off"],
                    help="prefix K/V cache of the target self-attention (auto: on beyond 32 tokens, mtn_amd.decode.KV_CACHE_FROM)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    from mtn_amd import lib, make_model
    from mtn_amd.decode import beam_search_decode, beam_search_decode_many, greedy_decode
    KV = {"auto": None, "on": True, "off": False}[args.kv_cache]
    from mtn_amd.synthetic import CONFIGS, synthetic_batch
    assert torch.cuda.is_available(), "bench_decode.py needs a GPU (the HIP path has no CPU fallback)"
    dev = torch.device("cuda", 0)
    lib.load()
    cfg = dict(CONFIGS["cfg2"])
    lp = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    model = make_model(cfg["vocab"], cfg["vocab"], N=cfg["N"], d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"],
                       dropout=0.1, ft_sizes=cfg["ft_sizes"], diff_encoder=True, diff_embed=False, diff_gen=False,
                       auto_encoder_ft="query", compute_dtype=lp).to(dev).eval()
    SOS, EOS, UNK, PAD = 2, 3, 0, 1
    batches = [synthetic_batch(cfg["vocab"], 1, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"],
                               device=dev, seed=100 + i, ragged=False) for i in range(args.dialogues)]

    def run_beam(b):
        return beam_search_decode(model, b, args.max_len, SOS, UNK, EOS, PAD, beam=args.beam, nbest=args.beam, kv_cache=KV,
                                  use_graph=not args.no_graph)

    for i in range(3):                                     # warm-up: allocator, the search graph, and (third dialogue of a shape) the capture
        run_beam(batches[i % len(batches)])                # of the per-dialogue encoder-side pass: the steady state is timed
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches:
        nbest, _ = run_beam(b)
    torch.cuda.synchronize()
    t_beam = time.perf_counter() - t0
    live = 1 + (args.max_len - 1) * args.beam              # step 0 extends <sos> only, later steps `beam` hypotheses
    tok_beam = args.dialogues * live / t_beam

    for i in range(3):
        greedy_decode(model, batches[i % len(batches)], args.max_len, SOS, PAD, use_graph=not args.no_graph, kv_cache=KV)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches:
        greedy_decode(model, b, args.max_len, SOS, PAD, use_graph=not args.no_graph, kv_cache=KV)
    torch.cuda.synchronize()
    t_greedy = time.perf_counter() - t0

    line = {"metric": f"decode tokens/sec (beam-{args.beam}, d_model=512, 6L MTN, max_len {args.max_len})",
            "value": round(tok_beam, 1), "unit": "hypothesis-tokens/s", "n_gpus": 1, "higher_is_better": True,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "cfg5: beam search over the cfg2 model, one dialogue at a time", "beam": args.beam,
                       "max_len": args.max_len, "dialogues": args.dialogues, "hip_graph": not args.no_graph},
            "beam": {"dialogues_per_s": round(args.dialogues / t_beam, 2), "ms_per_dialogue": round(1e3 * t_beam / args.dialogues, 2),
                     "ms_per_step": round(1e3 * t_beam / args.dialogues / args.max_len, 3)},
            "greedy": {"tokens_per_s": round(args.dialogues * (args.max_len - 1) / t_greedy, 1),
                       "ms_per_step": round(1e3 * t_greedy / args.dialogues / (args.max_len - 1), 3)}}
    # two / four dialogues per search = 8 / 16 hypothesis rows in the persistent decode-step kernel (16 = the rows of its MFMA tiles: the
    # widest launch it takes; round 5 stopped at 8)
    many_of = lambda b: beam_search_decode_many(model, b, args.max_len, SOS, UNK, EOS, PAD, beam=args.beam, nbest=args.beam, use_graph=not args.no_graph, kv_cache=KV)
    for D_, key in ((2, "beam_two_side_by_side"), (4, "beam_four_side_by_side")):
        if D_ * args.beam > 16:
            continue
        groups = [synthetic_batch(cfg["vocab"], D_, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=200 + 10 * D_ + i, ragged=False)
                  for i in range(max(2, args.dialogues // D_))]
        for i in range(3):
            many_of(groups[i % len(groups)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in groups:
            many_of(b)
        torch.cuda.synchronize()
        t_g = time.perf_counter() - t0
        line[key] = {"hypothesis_tokens_per_s": round(D_ * len(groups) * live / t_g, 1), "dialogues_per_s": round(D_ * len(groups) / t_g, 2),
                     "ms_per_step": round(1e3 * t_g / len(groups) / args.max_len, 3)}
    from mtn_amd.decode import MegaDecodeSession
    line["persistent_step_fallbacks"] = MegaDecodeSession.FALLBACKS          # searches re-run on the launch pass after a poll timeout: 0 expected
    if args.batch_dialogues > 0:
        D = args.batch_dialogues
        big = synthetic_batch(cfg["vocab"], D, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=300, ragged=True)
        many = lambda: beam_search_decode_many(model, big, args.max_len, SOS, UNK, EOS, PAD, beam=args.beam, nbest=args.beam, use_graph=not args.no_graph, kv_cache=KV)
        many()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            many()
        torch.cuda.synchronize()
        t_many = (time.perf_counter() - t0) / 3
        line["beam_batched"] = {"dialogues_side_by_side": D, "hypothesis_tokens_per_s": round(D * live / t_many, 1),
                                "dialogues_per_s": round(D / t_many, 2), "ms_per_step": round(1e3 * t_many / args.max_len, 3)}
    if not args.no_cpu_baseline:
        from bench import decode_cpu_baseline          # the only place outside tests/ that runs the oracle is bench.py's cpu_baseline leg
        line["cpu_baseline"] = decode_cpu_baseline(model, cfg, args.max_len, args.beam, live, SOS, UNK, EOS)
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
