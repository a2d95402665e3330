"""Where the ragged-corpus training loop (bench.py secondary.corpus_loop: BucketedTrainer over a synthetic DeviceCorpus) spends its time:
host time per step (enqueue only), GPU time per step by shape (HIP events around the replay), samples per step — and the same shapes'
captured steps replayed back to back without the per-step host work (assembly, norms refresh), i.e. the loop's GPU floor."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtn_amd import make_model
from mtn_amd.data_handler import DeviceCorpus, make_batch_indices
from mtn_amd.synthetic import CONFIGS
from mtn_amd.train import synthetic_corpus
from mtn_amd.train_step import BucketedTrainer

dev = torch.device("cuda:0")
cfg = dict(CONFIGS["cfg2"]); torch.manual_seed(0)
model = make_model(cfg["vocab"], cfg["vocab"], N=cfg["N"], d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"], dropout=0.1, ft_sizes=cfg["ft_sizes"],
                   diff_encoder=True, auto_encoder_ft="query", compute_dtype=torch.bfloat16, attn_dropout=0.1).to(dev).train()
model.prepare()
data = synthetic_corpus(24, cfg["vocab"], cfg["ft_sizes"], 5, max_answer=52, max_question=42)
indices, n_samples = make_batch_indices(data, batchsize=32, max_length=256, separate_caption=True)
corpus = DeviceCorpus(data, dev)
tr = BucketedTrainer(model, corpus, cfg["vocab"], pad=1, warmup=4000, bucket=8)
for idx in indices:
    tr.step(idx)
torch.cuda.synchronize()
print(f"{len(indices)} batches per epoch, {n_samples} dialog turns, {len(tr.steps)} padded shapes")
# (a) the loop as bench.py times it
host, t0 = [], time.perf_counter()
for _ in range(2):
    for idx in indices:
        h0 = time.perf_counter(); tr.step(idx); host.append(time.perf_counter() - h0)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
steps = 2 * len(indices)
print(f"loop: {dt / steps * 1e3:.3f} ms per step wall, host enqueue {sum(host) / steps * 1e3:.3f} ms per step (median {sorted(host)[len(host) // 2] * 1e3:.3f}), {2 * n_samples / dt:.0f} samples/s")
# (b) GPU time of each shape's captured step alone (5 replays back to back), weighted by how often the epoch uses it
per_shape = {}
for idx in indices:
    pidx = tr._padded(idx); key = (tuple(pidx[2]),) + tuple(pidx[3:])
    per_shape.setdefault(key, []).append(idx[-1])
tot = 0.0
for key, ns in per_shape.items():
    ts = tr.steps[key][1]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts(); torch.cuda.synchronize()
    e0.record()
    for _ in range(5): ts()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    tot += ms * len(ns)
    print(f"  shape frames {key[0]} H {key[1]} Q {key[2]} A {key[3]} C {key[4]} B {key[5]}: {ms:.3f} ms per captured step x {len(ns)} batches ({sum(ns)} samples)")
print(f"GPU floor of an epoch (captured steps only): {tot / len(indices):.3f} ms per step -> {n_samples / (tot * 1e-3):.0f} samples/s")
