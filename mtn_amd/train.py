#!/usr/bin/env python
"""Training entry point with the reference's model/optimiser flags (train.py:57-96), so that the `python train.py ...`
line of run.sh:109-140 can launch this implementation:  python -m mtn_amd.train --nb-blocks 6 --d-model 512 ...

Dataset I/O (data_handler.py: DSTC7-AVSD json + .npy features) is out of the hot-path scope; without
``--train-set`` the loop runs on synthetic batches of the reference's shapes.  Dataset flags are accepted and ignored
with a notice, so existing command lines keep working.  One process per GPU under torchrun gives data parallelism
(RCCL all-reduce of the flat gradient buffer).
"""
import argparse
import logging
import time

import torch

from . import dp, make_model
from .synthetic import synthetic_batch
from .train_step import TrainStep


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpu", "-g", default=0, type=int)
    # data flags of the reference (accepted; synthetic data is used when --train-set is empty)
    for f in ("--fea-type",):
        p.add_argument(f, nargs="+", type=str, default=["i3d_rgb", "vggish"])
    for f in ("--train-path", "--train-set", "--valid-path", "--valid-set", "--model"):
        p.add_argument(f, default="", type=str)
    p.add_argument("--include-caption", default="none", type=str)
    p.add_argument("--separate-caption", default=1, type=int)
    p.add_argument("--cut-a", default=0, type=int)
    p.add_argument("--merge-source", default=0, type=int)
    p.add_argument("--exclude-video", action="store_true")
    p.add_argument("--fixed-word-emb", default=0, type=int)
    # model (train.py:75-84)
    p.add_argument("--nb-blocks", default=6, type=int)
    p.add_argument("--d-model", default=512, type=int)
    p.add_argument("--d-ff", default=2048, type=int)
    p.add_argument("--att-h", default=8, type=int)
    p.add_argument("--dropout", default=0.1, type=float)
    p.add_argument("--separate-his-embed", default=0, type=int)
    p.add_argument("--separate-cap-embed", default=0, type=int)
    p.add_argument("--diff-encoder", default=1, type=int)
    p.add_argument("--diff-embed", default=0, type=int)
    p.add_argument("--diff-gen", default=0, type=int)
    p.add_argument("--auto-encoder-ft", default="query", type=str)
    # training (train.py:86-93)
    p.add_argument("--num-epochs", "-e", default=1, type=int)
    p.add_argument("--rand-seed", "-s", default=1, type=int)
    p.add_argument("--batch-size", "-b", default=32, type=int)
    p.add_argument("--max-length", default=20, type=int)
    p.add_argument("--max-history-length", default=-1, type=int)
    p.add_argument("--report-interval", default=100, type=int)
    p.add_argument("--warmup-steps", default=4000, type=int)
    p.add_argument("--loss-l", default=1.0, type=float)
    p.add_argument("--verbose", "-v", default=0, type=int)
    # this implementation
    p.add_argument("--compute-dtype", default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--steps-per-epoch", default=200, type=int, help="synthetic mode: batches per epoch")
    p.add_argument("--vocab-size", default=3000, type=int, help="synthetic mode")
    p.add_argument("--ft-sizes", default=[2048, 128], nargs="+", type=int, help="synthetic mode: feature dims")
    p.add_argument("--lens", default=[20, 128, 40, 20, 32], nargs=5, type=int, help="synthetic mode: Q H C T frames")
    return p.parse_args(argv)


def main(argv=None):
    args = parse(argv)
    logging.basicConfig(level=logging.DEBUG if args.verbose else logging.INFO, format="%(asctime)s %(levelname)s: %(message)s")
    rank, world, local = dp.init_distributed()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if args.train_set:
        logging.warning("dataset loading is outside this implementation's scope; running on synthetic batches instead")
    torch.manual_seed(args.rand_seed)
    model = make_model(args.vocab_size, args.vocab_size, N=args.nb_blocks, d_model=args.d_model, d_ff=args.d_ff, h=args.att_h,
                       dropout=args.dropout, separate_his_embed=bool(args.separate_his_embed), separate_cap_embed=bool(args.separate_cap_embed),
                       ft_sizes=args.ft_sizes, diff_encoder=bool(args.diff_encoder), diff_embed=bool(args.diff_embed),
                       diff_gen=bool(args.diff_gen), auto_encoder_ft=args.auto_encoder_ft, compute_dtype=args.compute_dtype)
    model.to(dev).train()
    model.prepare()
    sync = None
    if world > 1:
        sync = dp.GradSync(lambda: model.flat_buffers()[2])
        sync.broadcast_(model._flat)
        model._flat_version = -1
        model.prepare()
    Q, H, C, T, V = args.lens
    batch = synthetic_batch(args.vocab_size, args.batch_size, Q, H, C, T, [V] * len(args.ft_sizes), args.ft_sizes, device=dev, seed=1 + rank)
    step = TrainStep(model, batch, args.vocab_size, pad=1, warmup=args.warmup_steps, lam=args.loss_l, grad_sync=sync)
    ntok = int(batch.ntokens) * world
    for epoch in range(args.num_epochs):
        t0, tokens = time.time(), 0
        for j in range(args.steps_per_epoch):
            loss = step()
            tokens += ntok
            if (j + 1) % args.report_interval == 0 and rank == 0:
                torch.cuda.synchronize()
                dt = time.time() - t0
                print("Epoch: %d Step: %d Loss: %f Tokens per Sec: %f" % (epoch + 1, j + 1, float(loss), tokens / dt))   # train.py:46
                t0, tokens = time.time(), 0
        if args.model and rank == 0:
            torch.save(model.state_dict(), f"{args.model}_{epoch + 1}.pth.tar")      # reference key schema
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
