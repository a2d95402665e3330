"""The RCCL-native data-parallel schedule on real hardware: dp.ShardedOptimizerSync's `native` branch —
in-place reduce_scatter_tensor on a view of its own input, Adam on the rank's shard on the side stream, in-place
all_gather_into_tensor, finish() — driven by the layer-segmented TrainStep (N+3 hipGraphs with the collectives issued eagerly
between them).  The insertion point in the reference is data_utils.py:153-155 (between loss.backward() and opt.step()).

  * one rank, backend "nccl" (MTN_FORCE_DIST=1): every collective really goes through RCCL (per = the whole slice, own = lo);
    parameters and both Adam moments must equal, BIT FOR BIT, the same schedule with the collectives skipped, and the losses of
    the fused single-rank step (the bench line's step);
  * two ranks, backend "nccl", one GPU each — skipped unless the box has two GPUs: N ranks x half batch == 1 rank x whole batch.
Each case runs in spawned processes (a process group cannot be re-created with another backend inside the pytest process)."""
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


def _cfg():
    from oracle import fixtures as fx
    return dict(fx.GOLDEN_CONFIGS["wide_n1"], N=2, B=4)          # d_model 512 / 8 heads: the fused kernels are on the path


def _make(c, dev, dtype, dropout):
    from mtn_amd import make_model
    from oracle import fixtures as fx
    m = make_model(c["vocab"], c["vocab"], N=c["N"], d_model=c["d_model"], d_ff=c["d_ff"], h=c["h"], dropout=dropout, ft_sizes=c["ft_sizes"],
                   diff_encoder=c["diff_encoder"], diff_embed=c["diff_embed"], diff_gen=c["diff_gen"], auto_encoder_ft=c["auto_encoder_ft"],
                   compute_dtype=dtype, attn_dropout=dropout)
    m.load_state_dict(fx.det_state_dict(fx.state_shapes(**c)), strict=False)
    return m.to(dev).train()


def _batch(c, raw, dev):
    from mtn_amd import Batch
    from oracle import fixtures as fx
    t = torch.from_numpy
    return Batch(t(raw["query"]), t(raw["his"]), None, [t(f) for f in raw["fts"]], t(raw["cap"]), t(raw["trg"]), t(raw["trg_y"]), pad=fx.PAD, device=dev)


def _run_steps(model, batch, c, sync, use_graph, n=3, collective=None, **kw):
    from mtn_amd.train_step import TrainStep
    from oracle import fixtures as fx
    step = TrainStep(model, batch, c["vocab"], pad=fx.PAD, warmup=10, grad_sync=sync, use_graph=use_graph, **kw)
    if collective is not None:
        assert step.sharded is not None
        step.sharded.collective = collective
    if use_graph:
        step._capture()                      # warm-up passes advance the dropout seed differently per schedule ...
    torch.cuda.synchronize()
    model._seed.fill_(4242)                  # ... so every variant starts its counted steps from the same seed
    losses = [float(step()) for _ in range(n)]
    torch.cuda.synchronize()
    adam = step.opt.optimizer
    return step, losses, dict(flat=model._flat.detach().cpu().clone(), m=adam.m.cpu().clone(), v=adam.v.cpu().clone())


def _one_rank_worker(rank, port, out_dir, use_graph, dtype):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MTN_FORCE_DIST="1",
                      MTN_DIST_BACKEND="nccl", MTN_EMBED_DETERMINISTIC="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.pop("MTN_DP_SHARDED", None)
    os.environ.pop("MTN_DP_OVERLAP", None)
    import torch.distributed as dist
    from mtn_amd import dp
    from oracle import fixtures as fx
    r, w, _ = dp.init_distributed()
    assert (r, w) == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl"
    dev = torch.device("cuda:0")
    c = _cfg()
    raw = fx.det_batch(c["vocab"], c["B"], c["Q"], c["H"], c["C"], c["T"], c["frames"], c["ft_sizes"], seed=3)
    drop = 0.1 if dtype == "bf16" else 0.0
    res = {}
    # (a) the RCCL chain
    torch.manual_seed(0)
    model = _make(c, dev, dtype, drop)
    sync = dp.GradSync(lambda: model.flat_buffers()[2], n_buckets=3)
    assert sync.force
    sync.broadcast_(model._flat)
    step, losses, st = _run_steps(model, _batch(c, raw, dev), c, sync, use_graph)
    sh = step.sharded
    assert sh is not None and sh.native and sh.collective and step.overlap
    res["rccl"] = dict(st, losses=losses, calls=dict(sh.calls), n_slices=len(step._slices()))
    # (b) the same schedule, collectives skipped (what the gloo-free single rank does)
    torch.manual_seed(0)
    model_b = _make(c, dev, dtype, drop)
    sync_b = dp.GradSync(lambda: model_b.flat_buffers()[2], n_buckets=3)
    _, losses_b, st_b = _run_steps(model_b, _batch(c, raw, dev), c, sync_b, use_graph, collective=False)
    res["plain"] = dict(st_b, losses=losses_b)
    # (c) the fused single-rank step (bench.py's N = 1 line)
    torch.manual_seed(0)
    model_c = _make(c, dev, dtype, drop)
    _, losses_c, st_c = _run_steps(model_c, _batch(c, raw, dev), c, None, use_graph)
    res["fused"] = dict(st_c, losses=losses_c)
    # (d) the simple (two-graph) schedule takes the same slices: same owner of every element, same bits
    torch.manual_seed(0)
    model_d = _make(c, dev, dtype, drop)
    sync_d = dp.GradSync(lambda: model_d.flat_buffers()[2], n_buckets=3)
    step_d, losses_d, st_d = _run_steps(model_d, _batch(c, raw, dev), c, sync_d, use_graph, overlap=False)
    assert step_d.sharded.slices == sh.slices
    res["simple"] = dict(st_d, losses=losses_d)
    # (e) the all-reduce scheme (MTN_DP_SHARDED=0) through RCCL: asynchronous slice all-reduces + the full optimiser pass
    os.environ["MTN_DP_SHARDED"] = "0"
    torch.manual_seed(0)
    model_e = _make(c, dev, dtype, drop)
    sync_e = dp.GradSync(lambda: model_e.flat_buffers()[2], n_buckets=3)
    step_e, losses_e, st_e = _run_steps(model_e, _batch(c, raw, dev), c, sync_e, use_graph)
    assert step_e.sharded is None
    res["allreduce"] = dict(st_e, losses=losses_e)
    torch.save(res, os.path.join(out_dir, "one_rank.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
@pytest.mark.parametrize("use_graph", [True, False], ids=["graphs", "eager"])
def test_one_rank_rccl_chain_matches_plain_schedule_bitwise(tmp_path, use_graph, dtype):
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    mp.start_processes(_one_rank_worker, args=(_free_port(), str(tmp_path), use_graph, dtype), nprocs=1, join=True, start_method="spawn")
    res = torch.load(tmp_path / "one_rank.pt")
    rccl, plain, fused, simple, allr = res["rccl"], res["plain"], res["fused"], res["simple"], res["allreduce"]
    # every slice of every step went through RCCL: one reduce-scatter and one all-gather each
    n_sl = rccl["n_slices"]
    assert rccl["calls"]["reduce_scatter"] == 3 * n_sl and rccl["calls"]["all_gather"] == 3 * n_sl, rccl["calls"]
    for k in ("flat", "m", "v"):
        assert torch.equal(rccl[k], plain[k]), f"{k}: RCCL chain differs from the collective-free schedule"
        assert torch.equal(rccl[k], allr[k]), f"{k}: all-reduce + full optimiser pass differs from the sharded chain"
    assert rccl["losses"] == plain["losses"] == allr["losses"]
    # the two-graph schedule (monolithic backward: one parameter-gradient flush) and the fused single-rank step (one table launch
    # with the optimiser in its epilogue) compute the parameter gradients with other tilings -> the same step up to fp32
    # summation order: the losses of three consecutive steps agree and the first-moment vectors coincide.  (Adam turns
    # noise-level gradients into O(lr) parameter steps of either sign, so parameters are compared through a loose bound only.)
    for other in (simple, fused):
        for a, b in zip(rccl["losses"], other["losses"]):
            assert abs(a - b) < (1e-4 if dtype == "fp32" else 3e-3) * abs(b), (rccl["losses"], other["losses"])
        ma, mb = rccl["m"].double(), other["m"].double()
        cos = float((ma * mb).sum() / (ma.norm() * mb.norm()))
        assert cos > (0.999999 if dtype == "fp32" else 0.999), cos
        assert float((rccl["flat"] - other["flat"]).abs().max()) < 1e-2


def _two_rank_worker(rank, world, port, out_dir, use_graph):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MTN_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from mtn_amd import dp
    from oracle import fixtures as fx
    dp.init_distributed()
    dev = torch.device("cuda", rank)
    c = _cfg()
    raw = fx.det_batch(c["vocab"], c["B"], c["Q"], c["H"], c["C"], c["T"], c["frames"], c["ft_sizes"], seed=3)
    s, e = dp.shard_range(c["B"], rank, world)
    shard = {k: (v[s:e] if k != "fts" else [f[s:e] for f in v]) for k, v in raw.items()}
    model = _make(c, dev, "fp32", 0.0)
    sync = dp.GradSync(lambda: model.flat_buffers()[2], n_buckets=3)
    sync.broadcast_(model._flat)
    step, losses, st = _run_steps(model, _batch(c, shard, dev), c, sync, use_graph)
    assert step.sharded is not None and step.sharded.native
    opt_state = step.opt.optimizer.state_dict()               # every rank: gathers the sharded moments (collective)
    torch.save(dict(st, losses=losses, exp_avg=opt_state["exp_avg"]), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL allows one rank per device)")
@pytest.mark.parametrize("use_graph", [True, False], ids=["graphs", "eager"])
def test_two_rank_rccl_equals_single_rank_on_concatenated_batch(tmp_path, use_graph):
    import torch.multiprocessing as mp
    from oracle import fixtures as fx
    mp.start_processes(_two_rank_worker, args=(2, _free_port(), str(tmp_path), use_graph), nprocs=2, join=True, start_method="spawn")
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert torch.equal(r0["flat"], r1["flat"])                                # replicas stay identical
    for k in r0["exp_avg"]:
        assert torch.equal(r0["exp_avg"][k], r1["exp_avg"][k]), k            # gathered moments complete on both ranks
    dev = torch.device("cuda:0")
    c = _cfg()
    raw = fx.det_batch(c["vocab"], c["B"], c["Q"], c["H"], c["C"], c["T"], c["frames"], c["ft_sizes"], seed=3)
    model = _make(c, dev, "fp32", 0.0)
    _, losses, st = _run_steps(model, _batch(c, raw, dev), c, None, False, fuse_optimizer=False)
    dp_losses = [a + b for a, b in zip(r0["losses"], r1["losses"])]          # every rank normalises by the GLOBAL token counts
    for a, b in zip(dp_losses, losses):
        assert abs(a - b) < 2e-3 * abs(b), (dp_losses, losses)
    cos = float((r0["m"].double() * st["m"].double()).sum() / (r0["m"].double().norm() * st["m"].double().norm()))
    assert cos > 0.99999, cos
