"""Data parallelism with the real HIP path: two ranks sharing cuda:0 (gloo backend, the only one that allows two ranks on one
device) run TrainStep (hipGraph fwd+bwd | eager all-reduce of the flat gradient | hipGraph Adam) on half batches and must
end with the parameters of ONE rank stepping on the concatenated batch."""
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(c, dev, dtype):
    from mtn_amd import make_model
    from oracle import fixtures as fx
    m = make_model(c["vocab"], c["vocab"], N=c["N"], d_model=c["d_model"], d_ff=c["d_ff"], h=c["h"], dropout=0.0, ft_sizes=c["ft_sizes"],
                   diff_encoder=c["diff_encoder"], diff_embed=c["diff_embed"], diff_gen=c["diff_gen"], auto_encoder_ft=c["auto_encoder_ft"],
                   compute_dtype=dtype, attn_dropout=0.0)
    m.load_state_dict(fx.det_state_dict(fx.state_shapes(**c)), strict=False)
    return m.to(dev).train()


def _batch(c, raw, dev):
    from mtn_amd import Batch
    from oracle import fixtures as fx
    t = torch.from_numpy
    return Batch(t(raw["query"]), t(raw["his"]), None, [t(f) for f in raw["fts"]], t(raw["cap"]), t(raw["trg"]), t(raw["trg_y"]), pad=fx.PAD, device=dev)


def _worker(rank, world, port, out_dir, use_graph, sharded=True):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      MTN_DP_SHARDED="1" if sharded else "0")
    import torch.distributed as dist
    from mtn_amd import dp
    from mtn_amd.train_step import TrainStep
    from oracle import fixtures as fx
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = dict(fx.GOLDEN_CONFIGS["cfg1_query"], B=4)
    raw = fx.det_batch(c["vocab"], c["B"], c["Q"], c["H"], c["C"], c["T"], c["frames"], c["ft_sizes"], seed=3)
    s, e = dp.shard_range(c["B"], rank, world)
    shard = {k: (v[s:e] if k != "fts" else [f[s:e] for f in v]) for k, v in raw.items()}
    model = _make(c, dev, "fp32")
    sync = dp.GradSync(lambda: model.flat_buffers()[2], n_buckets=3)
    step = TrainStep(model, _batch(c, shard, dev), c["vocab"], pad=fx.PAD, warmup=10, grad_sync=sync, use_graph=use_graph)
    assert (step.sharded is not None) == sharded
    losses = [float(step())]
    torch.cuda.synchronize()
    grad1 = model._flat_grad.cpu().clone()          # reduced gradient of step 1 (the optimiser does not touch it)
    losses += [float(step()) for _ in range(2)]
    torch.cuda.synchronize()
    torch.save({"flat": model._flat.cpu(), "grad1": grad1, "losses": losses}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("sharded", [True, False], ids=["sharded-optimiser", "allreduce-full-optimiser"])
@pytest.mark.parametrize("use_graph", [True, False])
def test_two_rank_dp_equals_single_rank_on_concatenated_batch(tmp_path, use_graph, sharded):
    """Both exchange schemes: reduce-scatter + optimiser on the own shard + all-gather of the master weights
    (dp.ShardedOptimizerSync, the default), and all-reduce + the full optimiser pass on every rank (MTN_DP_SHARDED=0)."""
    import torch.multiprocessing as mp
    from mtn_amd.train_step import TrainStep
    from oracle import fixtures as fx
    assert torch.cuda.is_available()
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, str(tmp_path), use_graph, sharded), nprocs=2, join=True, start_method="spawn")
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert torch.equal(r0["flat"], r1["flat"])                       # replicas stay identical
    dev = torch.device("cuda:0")
    c = dict(fx.GOLDEN_CONFIGS["cfg1_query"], B=4)
    raw = fx.det_batch(c["vocab"], c["B"], c["Q"], c["H"], c["C"], c["T"], c["frames"], c["ft_sizes"], seed=3)
    model = _make(c, dev, "fp32")
    step = TrainStep(model, _batch(c, raw, dev), c["vocab"], pad=fx.PAD, warmup=10, use_graph=False, fuse_optimizer=False)   # keeps .grad
    losses = [float(step())]
    torch.cuda.synchronize()
    gref = model._flat_grad.cpu().clone()
    losses += [float(step()) for _ in range(2)]
    # gradients: exact up to fp32 summation order (2 half-batch GEMMs + all-reduce vs one GEMM)
    err = float((r0["grad1"] - gref).abs().max() / gref.abs().max())
    assert err < 1e-5, err
    # losses: every rank normalises by the GLOBAL token counts, so the single-rank loss is the sum of the rank losses.
    # (Parameters after Adam are not compared element-wise: Adam turns noise-level gradients — the mathematically zero
    # key-bias gradients — into O(lr) updates of arbitrary sign.)
    dp_losses = [a + b for a, b in zip(r0["losses"], r1["losses"])]
    for a, b in zip(dp_losses, losses):
        assert abs(a - b) < 2e-3 * abs(b), (dp_losses, losses)


def _eager_worker(rank, world, port, out_dir):
    """The eager data-parallel loop body of mtn_amd.train.run_epoch (forward, global normalisers, SimpleLossCompute with the
    gradient exchange between backward and the optimiser step) on this rank's half of the batch."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from mtn_amd import FusedAdam, LabelSmoothing, NoamOpt, SimpleLossCompute, dp
    from mtn_amd.train import global_norms
    from oracle import fixtures as fx
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = dict(fx.GOLDEN_CONFIGS["cfg1_query"], B=4)
    raw = fx.det_batch(c["vocab"], c["B"], c["Q"], c["H"], c["C"], c["T"], c["frames"], c["ft_sizes"], seed=3)
    s, e = dp.shard_range(c["B"], rank, world)
    shard = {k: (v[s:e] if k != "fts" else [f[s:e] for f in v]) for k, v in raw.items()}
    model = _make(c, dev, "fp32")
    model.prepare()
    sync = dp.GradSync(lambda: model.flat_buffers()[2], n_buckets=3)
    opt = NoamOpt(c["d_model"], 1, 10, FusedAdam(model))
    lc = SimpleLossCompute(model.generator, model.auto_encoder_generator, LabelSmoothing(c["vocab"], fx.PAD, 0.1), opt=opt, grad_sync=sync)
    b = _batch(c, shard, dev)
    model.zero_glue_grads()
    out, ae_out = model.forward(b)
    norms, _ = global_norms(b, b.query, sync)
    lc(out, b.trg_y, norms[0], ae_out, b.query, norms[1])
    torch.cuda.synchronize()
    torch.save({"grad1": model._flat_grad.cpu().clone(), "flat": model._flat.cpu().clone()}, os.path.join(out_dir, f"eager{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_eager_two_rank_dp_equals_single_rank_on_concatenated_batch(tmp_path):
    """The eager loop (python -m mtn_amd.train --eager under torchrun) normalises by the global token counts: the summed
    gradient of two ranks on half batches is the gradient of one rank on the whole batch."""
    import torch.multiprocessing as mp
    from mtn_amd import FusedAdam, LabelSmoothing, NoamOpt, SimpleLossCompute
    from oracle import fixtures as fx
    port = _free_port()
    mp.start_processes(_eager_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = torch.load(tmp_path / "eager0.pt"), torch.load(tmp_path / "eager1.pt")
    assert torch.equal(r0["flat"], r1["flat"]) and torch.equal(r0["grad1"], r1["grad1"])
    dev = torch.device("cuda:0")
    c = dict(fx.GOLDEN_CONFIGS["cfg1_query"], B=4)
    raw = fx.det_batch(c["vocab"], c["B"], c["Q"], c["H"], c["C"], c["T"], c["frames"], c["ft_sizes"], seed=3)
    model = _make(c, dev, "fp32")
    model.prepare()
    opt = NoamOpt(c["d_model"], 1, 10, FusedAdam(model))
    lc = SimpleLossCompute(model.generator, model.auto_encoder_generator, LabelSmoothing(c["vocab"], fx.PAD, 0.1), opt=opt)
    b = _batch(c, raw, dev)
    model.zero_glue_grads()
    out, ae_out = model.forward(b)
    lc(out, b.trg_y, b.ntokens, ae_out, b.query, (b.query != fx.PAD).sum())
    torch.cuda.synchronize()
    gref = model._flat_grad.cpu()
    err = float((r0["grad1"] - gref).abs().max() / gref.abs().max())
    assert err < 1e-5, err


def _train_cli_worker(rank, world, port, out_dir):
    """python -m mtn_amd.train (corpus mode, captured graphs per padded shape) under two ranks, with --model: the epoch-end
    checkpoint gathers the sharded Adam moments with collectives on EVERY rank, rank 0 writes the files (train.py:215-217)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      MTN_DIST_BACKEND="gloo")
    os.environ.pop("MTN_DP_SHARDED", None)
    from mtn_amd import train
    means = train.main(["--corpus-videos", "4", "--num-epochs", "2", "--batch-size", "8", "--nb-blocks", "1", "--d-model", "64", "--d-ff", "128",
                        "--att-h", "2", "--vocab-size", "60", "--ft-sizes", "16", "8", "--warmup-steps", "10", "--report-interval", "1000",
                        "--compute-dtype", "fp32", "--valid-videos", "2", "--model", os.path.join(out_dir, "ck")])
    torch.save({"means": means}, os.path.join(out_dir, f"cli{rank}.pt"))


@pytest.mark.timeout(900)
def test_two_rank_training_cli_checkpoints_without_deadlock(tmp_path):
    """rank 0 alone used to call the optimiser's state_dict(), whose sharded-moment gather is a collective the
    other ranks never joined.  Two ranks through the CLI with --model and validation: both epochs finish, the files exist, and the
    saved first moments are complete (the halves owned by rank 1 are non-zero too)."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.start_processes(_train_cli_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    for e in (1, 2):
        assert (tmp_path / f"ck_{e}.pth.tar").exists() and (tmp_path / f"ck_{e}_opt.pth.tar").exists()
    assert (tmp_path / "ck_best.pth.tar").exists()
    opt = torch.load(tmp_path / "ck_2_opt.pth.tar")
    assert int(opt["step"]) > 0
    for name, m1 in opt["optimizer"]["exp_avg"].items():
        if m1.dim() == 2 and "linears.1" not in name and m1.numel() >= 64:
            flat = m1.reshape(-1)
            n = flat.numel() // 4
            assert all(float(flat[i * n:(i + 1) * n].abs().max()) > 0 for i in range(4)), name     # every quarter was updated by some rank
    r0, r1 = torch.load(tmp_path / "cli0.pt"), torch.load(tmp_path / "cli1.pt")
    assert r0["means"] == r1["means"] and len(r0["means"]) == 2
