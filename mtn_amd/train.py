#!/usr/bin/env python
"""Training entry point with the reference's model/optimiser flags (train.py:57-96), so that the `python train.py ...`
line of run.sh:109-140 can launch this implementation:  python -m mtn_amd.train --nb-blocks 6 --d-model 512 ...

Three modes:
  * ``--train-set <json> --train-path <.../<FeaType>/<ImageID>.npy>`` (what run.sh passes): the reference's pipeline —
    vocabulary from the training annotations, DSTC7-AVSD json + per-video .npy features loaded once and uploaded whole
    (mtn_amd.data_handler), length-bucketed batches, an epoch loop on captured graphs per padded shape (or ``--eager``),
    validation loss after every epoch, <model>.conf / <model>_params.txt / <model>_N.pth.tar / <model>_best.pth.tar written
    as train.py:166-225 does (checkpoints are state_dicts in the reference's key schema);
  * ``--corpus-videos N``: the same loop on a synthetic ragged corpus;
  * neither: one fixed-shape synthetic batch replayed (throughput runs).
One process per GPU under torchrun gives data parallelism (RCCL all-reduce of the flat gradient buffer).
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
    # data flags of the reference (train.py:59-73); synthetic data is used when --train-set is empty
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
    p.add_argument("--valid-videos", default=0, type=int,
                   help="corpus mode: a held-out synthetic corpus of that many videos; its mean loss per token is evaluated in eval() mode "
                        "after every epoch and the best model is kept as <model>_best (train.py:201-224)")
    p.add_argument("--resume", default="", type=str,
                   help="corpus mode: checkpoint prefix to continue from (<prefix>.pth.tar = model state_dict in the reference's key "
                        "schema, <prefix>_opt.pth.tar = optimiser moments + schedule state)")
    p.add_argument("--eager", action="store_true", help="corpus mode: one eager step per batch instead of captured graphs per padded shape")
    p.add_argument("--bucket", default=8, type=int, help="corpus mode: batch lengths are rounded up to multiples of this")
    p.add_argument("--corpus-max-answer", default=18, type=int, help="synthetic corpus: longest answer (AVSD's reach 52 tokens)")
    p.add_argument("--corpus-max-question", default=20, type=int, help="synthetic corpus: longest question (AVSD's reach 42 tokens)")
    p.add_argument("--corpus-videos", default=0, type=int,
                   help="> 0: train on a synthetic ragged CORPUS of that many videos (10 turns each) through the reference's "
                        "epoch loop — batch planning, device-side batch assembly, one eager step per batch — instead of "
                        "replaying one fixed-shape batch")
    return p.parse_args(argv)


def synthetic_corpus(n_videos, vocab, ft_sizes, seed, turns=10, max_answer=18, max_question=20):
    """A ragged corpus in the layout data_handler.load returns (dialogs + per-video feature arrays), lengths in the ranges
    of the AVSD data: questions/answers 3-20 tokens, captions 10-40, 20-40 frames per video."""
    import numpy as np
    rs = np.random.RandomState(seed)
    tok = lambda lo, hi: rs.randint(4, vocab, size=rs.randint(lo, hi + 1)).astype(np.int64)
    dialogs, qa = [], 0
    vids = [f"v{v:05d}" for v in range(n_videos)]
    for v in vids:
        cap, hist = tok(10, 40), np.zeros(0, np.int64)
        for _ in range(turns):
            q, a = tok(3, max_question), tok(3, max_answer)
            ans = np.concatenate([[2], a, [3]]).astype(np.int64)                     # <sos> ... <eos>
            dialogs.append([v, qa, hist.copy() if len(hist) else np.array([1], np.int64), q, ans[:-1], ans[1:], cap])
            hist = np.concatenate([hist, q, a])
            qa += 1
    feats = [{v: rs.randn(rs.randint(20, 41), F).astype(np.float32) for v in vids} for F in ft_sizes]
    return {"dialogs": dialogs, "features": feats, "vocab": {"<blank>": 1}}


def run_epoch_graphed(trainer, indices, epoch, report_interval, rank, rng):
    """The same epoch on captured graphs (train_step.BucketedTrainer): lengths rounded up to the bucket, one graph per padded
    shape, batches assembled in place on the device; the host only syncs at the report interval."""
    order = list(range(len(indices)))
    rng.shuffle(order)
    dev = trainer.corpus.device
    loss_sum = torch.zeros((), device=dev)
    tok_sum = torch.zeros((), device=dev, dtype=torch.int64)
    t0, tok0 = time.time(), 0
    for j, k in enumerate(order):
        loss, b = trainer.step(indices[k])
        # the step's loss is already divided by the token count of the step over ALL ranks (train.py:36; dp.py): weight it
        # with that same count; the ranks' sums add up to the epoch mean of the global batches
        n_glob = b._norms_global[0].to(torch.int64)
        loss_sum += loss * n_glob
        tok_sum += n_glob
        if (j + 1) % report_interval == 0 and rank == 0:
            ntok = int(tok_sum)
            dt = time.time() - t0
            print("Epoch: %d Step: %d Loss: %f Tokens per Sec: %f" % (epoch + 1, j + 1, float(loss), (ntok - tok0) / dt))
            t0, tok0 = time.time(), ntok
    if torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        torch.distributed.all_reduce(loss_sum)
    return float(loss_sum) / max(1, int(tok_sum))


def validate(corpus, indices, model, criterion, ae_ft, lam):
    """train.py:201-210: the epoch loop with the model in eval() mode and no optimiser — mean loss per target token."""
    from .data_handler import make_batch
    from .data_utils import SimpleLossCompute
    lc = SimpleLossCompute(model.generator, model.auto_encoder_generator, criterion, opt=None, l=lam, sync=False)
    was_training = model.training
    model.eval()
    total = torch.zeros((), device=corpus.device)
    tokens = torch.zeros((), device=corpus.device, dtype=torch.int64)
    with torch.no_grad():
        for ix in indices:
            b = make_batch(corpus, ix, 1, separate_caption=True)
            out, ae_out = model.forward(b)
            ae_y = b.cap if ae_ft in ("caption", "summary") else b.query
            total += lc(out, b.trg_y, b.ntokens, ae_out, ae_y, (ae_y != 1).sum())
            tokens += b.ntokens
    model.train(was_training)
    return float(total) / max(1, int(tokens))


def global_norms(b, ae_y, sync):
    """[target tokens, auto-encoder tokens] of this step over ALL ranks (device tensor), and this rank's target tokens."""
    norms = torch.stack([b.ntokens, (ae_y != 1).sum()]).float()
    local_ntok = float(norms[0])
    if sync is not None and getattr(sync, "world", 1) > 1:
        sync.all_reduce_scalars(norms)
    return norms, local_ntok


def run_epoch(corpus, indices, model, loss_compute, ae_ft, epoch, report_interval, rank, rng):
    """train.py:23-50: shuffle the planned batches, assemble each on the device, forward, loss + backward + optimiser.
    Data parallel: the loss normalisers are the GLOBAL token counts of the step (all ranks' batches), so that the summed
    gradients are those of one rank on the concatenated batch (dp.py) — as the captured schedule does (TrainStep._norms)."""
    from .data_handler import make_batch
    order = list(range(len(indices)))
    rng.shuffle(order)
    t0, tokens, total_loss, total_tokens = time.time(), 0, 0.0, 0
    sync = loss_compute.grad_sync
    for j, k in enumerate(order):
        b = make_batch(corpus, indices[k], 1, separate_caption=True)
        out, ae_out = model.forward(b)
        ae_y = b.cap if ae_ft in ("caption", "summary") else b.query
        norms, local_ntok = global_norms(b, ae_y, sync)
        loss = loss_compute(out, b.trg_y, norms[0], ae_out, ae_y, norms[1])     # = normalised loss x global target tokens
        loss = loss * local_ntok / float(norms[0])                              # this rank's share, for the epoch mean below
        total_loss += loss
        total_tokens += int(b.ntokens)
        tokens += int(b.ntokens)
        if (j + 1) % report_interval == 0 and rank == 0:
            dt = time.time() - t0
            print("Epoch: %d Step: %d Loss: %f Tokens per Sec: %f" % (epoch + 1, j + 1, loss / float(b.ntokens), tokens / dt))
            t0, tokens = time.time(), 0
    return total_loss / max(1, total_tokens)


def main(argv=None):
    args = parse(argv)
    logging.basicConfig(level=logging.DEBUG if args.verbose else logging.INFO, format="%(asctime)s %(levelname)s: %(message)s")
    rank, world, local = dp.init_distributed()
    local = local % torch.cuda.device_count()      # (several ranks may share a GPU in a gloo dry run)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    train_data = valid_data = vocab = None
    if args.train_set:
        from . import data_handler as dh
        sep_cap = bool(args.separate_caption) and args.include_caption != "none"
        logging.info("Extracting words from " + args.train_set)
        vocab = dh.get_vocabulary(args.train_set, include_caption=args.include_caption)
        kw = dict(include_caption=args.include_caption, separate_caption=bool(args.separate_caption), vocab=vocab,
                  max_history_length=args.max_history_length, merge_source=bool(args.merge_source))
        logging.info("Loading training data from " + args.train_set)
        train_data = dh.load(args.fea_type, args.train_path, args.train_set, **kw)
        if args.valid_set:
            logging.info("Loading validation data from " + args.valid_set)
            valid_data = dh.load(args.fea_type, args.valid_path or args.train_path, args.valid_set, **kw)
        if not sep_cap:
            raise SystemExit("mtn_amd.train: the model needs the caption as its own stream (--separate-caption 1 with "
                             "--include-caption caption|summary|caption,summary), as run.sh sets it; the reference crashes "
                             "without it too (mtn.py:52)")
        args.vocab_size = len(vocab)
        args.ft_sizes = dh.feature_shape(train_data)
        logging.info("Detected feature dims: %s  #vocab = %d", args.ft_sizes, len(vocab))
        if args.model and rank == 0:                                  # train.py:166-172
            import os
            import pickle
            os.makedirs(os.path.dirname(args.model) or ".", exist_ok=True)
            with open(args.model + ".conf", "wb") as f:
                pickle.dump((vocab, args), f, -1)
            with open(args.model + "_params.txt", "w") as f:
                for arg in vars(args):
                    f.write("{}={}\n".format(arg, getattr(args, arg)))
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
    if args.corpus_videos > 0 or train_data is not None:
        import random
        from .data_handler import DeviceCorpus, make_batch_indices
        from .data_utils import FusedAdam, LabelSmoothing, NoamOpt, SimpleLossCompute
        synthetic = train_data is None
        max_len = 256 if synthetic else args.max_length         # (run.sh passes 256, the parser's default is the reference's 20)
        data = (synthetic_corpus(args.corpus_videos, args.vocab_size, args.ft_sizes, args.rand_seed, max_answer=args.corpus_max_answer,
                                 max_question=args.corpus_max_question) if synthetic else train_data)
        indices, n_samples = make_batch_indices(data, batchsize=args.batch_size, max_length=max_len, separate_caption=True)  # train.py:126
        indices = indices[:len(indices) // world * world][rank::world]   # data parallel: equal shares of the planned batches (every
                                                                         # rank must issue the same number of gradient exchanges)
        corpus = DeviceCorpus(data, dev)
        logging.info("corpus: %d dialogs in %d batches, %.1f MB resident on the device", n_samples, len(indices), corpus.nbytes() / 1e6)
        rng = random.Random(args.rand_seed)
        means = []
        if args.eager:
            opt = NoamOpt(args.d_model, 1, args.warmup_steps, FusedAdam(model))
            lc = SimpleLossCompute(model.generator, model.auto_encoder_generator, LabelSmoothing(args.vocab_size, 1, 0.1), opt=opt,
                                   l=args.loss_l, grad_sync=sync)
        else:
            from .train_step import BucketedTrainer
            trainer = BucketedTrainer(model, corpus, args.vocab_size, pad=1, warmup=args.warmup_steps, lam=args.loss_l,
                                      bucket=args.bucket, grad_sync=sync)
        valid = None
        if args.valid_videos > 0 or valid_data is not None:
            vdata = valid_data if valid_data is not None else synthetic_corpus(args.valid_videos, args.vocab_size, args.ft_sizes, args.rand_seed + 1000)
            vidx, vn = make_batch_indices(vdata, batchsize=args.batch_size, max_length=max_len, separate_caption=True)
            valid = (DeviceCorpus(vdata, dev), vidx)
            logging.info("#validation sample = %d  #validation batch = %d", vn, len(vidx))
        min_valid = 1.0e10
        the_opt = opt if args.eager else trainer.opt
        if args.resume:
            model.load_state_dict(torch.load(args.resume + ".pth.tar", map_location=dev), strict=False)
            model.prepare()
            the_opt.load_state_dict(torch.load(args.resume + "_opt.pth.tar"))
            logging.info("resumed from %s at optimiser step %d", args.resume, the_opt.step_count())
        for epoch in range(args.num_epochs):
            if args.eager:
                mean = run_epoch(corpus, indices, model, lc, args.auto_encoder_ft, epoch, args.report_interval, rank, rng)
            else:
                mean = run_epoch_graphed(trainer, indices, epoch, args.report_interval, rank, rng)
            means.append(mean)
            if rank == 0:
                print("epoch %d mean train loss per token: %f" % (epoch + 1, mean))
            if args.model:
                # EVERY rank builds the optimiser state: with the sharded optimiser (dp.ShardedOptimizerSync) a rank holds the
                # Adam moments of its shards only (and, with the compute-dtype gather, the current fp32 masters of its shards
                # only) and state_dict() gathers them with collectives; rank 0 alone writes the files
                opt_state = the_opt.state_dict()
                model_state = model.state_dict()              # (collective under the compute-dtype gather: fp32 masters)
                if rank == 0:
                    torch.save(model_state, f"{args.model}_{epoch + 1}.pth.tar")
                    torch.save(opt_state, f"{args.model}_{epoch + 1}_opt.pth.tar")
                del opt_state, model_state
            if valid is not None:
                best_state = model.state_dict() if args.model else None    # (collective) written by rank 0 alone if validation improves
                from .data_utils import LabelSmoothing as _LS
                vloss = validate(valid[0], valid[1], model, _LS(args.vocab_size, 1, 0.1), args.auto_encoder_ft, args.loss_l)
                if rank == 0:
                    print("epoch: %d validation loss: %f" % (epoch + 1, vloss))          # train.py:210
                    if vloss < min_valid:
                        print("validation loss reduced %.4f -> %.4f" % (min_valid, vloss))
                        min_valid = vloss
                        if args.model:
                            torch.save(best_state, f"{args.model}_best.pth.tar")
        if world > 1:
            torch.distributed.destroy_process_group()
        return means
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
        if args.model:
            model_state = model.state_dict()      # every rank: under the sharded optimiser's compute-dtype gather this gathers the fp32
            if rank == 0:                         # masters of the other ranks' shards (a collective) — rank 0 alone writes
                torch.save(model_state, f"{args.model}_{epoch + 1}.pth.tar")      # reference key schema
            del model_state
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
