"""Data-parallel path on CPU (gloo, world_size 2): the bucketed flat-gradient all-reduce + global-token loss normalisation
make N ranks x local batch equal ONE rank x concatenated batch.  The compute engine here is the CPU oracle
(the HIP kernels need a GPU); the DP plumbing under test (mtn_amd.dp) is the code the GPU path uses."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat_grads(params):
    return torch.cat([p.grad.reshape(-1) for p in params])


def _worker(rank, world, port, out_dir, mode):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from mtn_amd import dp
    from oracle import fixtures as fx
    r, w, _ = dp.init_distributed("gloo")
    assert (r, w) == (rank, world)
    c = dict(fx.GOLDEN_CONFIGS["small_diffall"], B=4)
    raw = fx.det_batch(c["vocab"], c["B"], c["Q"], c["H"], c["C"], c["T"], c["frames"], c["ft_sizes"], seed=5)
    s, e = dp.shard_range(c["B"], rank, world)
    shard = {k: (v[s:e] if k != "fts" else [f[s:e] for f in v]) for k, v in raw.items()}
    model, _ = fx.oracle_from_config(c, requires_grad=True)
    names = sorted(model.p)
    params = [model.p[k] for k in names]
    flat = torch.zeros(sum(p.numel() for p in params))
    off = 0
    for p in params:                                   # gradients live in ONE flat buffer, as in the GPU model
        p.grad = flat[off:off + p.numel()].view(p.shape)
        off += p.numel()
    sync = dp.GradSync(lambda: flat, n_buckets=3)
    b = fx.oracle_batch(shard)
    ae_y = b.cap
    norms = torch.stack([b.ntokens, (ae_y != fx.PAD).sum()]).float()
    sync.all_reduce_scalars(norms)                     # global token counts
    out, ae_out = model.forward(b)
    loss = model.loss(b, out, ae_out, norm=norms[0], ae_norm=norms[1])
    loss.backward()
    if mode == "buckets":
        sync()                                         # whole buffer, after backward
    else:                                              # the overlapped schedule's calls: asynchronous slices, one wait
        n = flat.numel()
        cuts = [0, n // 5, n // 2, n]
        sync.wait([sync.reduce_range(lo, hi) for lo, hi in zip(cuts[:-1], cuts[1:])][::-1] + [sync.reduce_range(7, 7)])
    torch.save({"flat": flat.clone(), "norms": norms}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode", ["buckets", "ranges"])
def test_two_rank_gradients_equal_single_rank_on_concatenated_batch(tmp_path, mode):
    sys.path.insert(0, ROOT)
    from oracle import fixtures as fx
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, str(tmp_path), mode), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    assert torch.equal(r0["flat"], r1["flat"])                       # replicas hold identical reduced gradients
    c = dict(fx.GOLDEN_CONFIGS["small_diffall"], B=4)
    raw = fx.det_batch(c["vocab"], c["B"], c["Q"], c["H"], c["C"], c["T"], c["frames"], c["ft_sizes"], seed=5)
    model, _ = fx.oracle_from_config(c, requires_grad=True)
    b = fx.oracle_batch(raw)
    out, ae_out = model.forward(b)
    model.loss(b, out, ae_out).backward()
    ref = torch.cat([model.p[k].grad.reshape(-1) for k in sorted(model.p)])
    assert torch.allclose(r0["norms"], torch.stack([b.ntokens, (b.cap != fx.PAD).sum()]).float())
    err = float((r0["flat"] - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err


def test_shard_range_and_buckets():
    from mtn_amd import dp
    assert [dp.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert sum(e - s for s, e in (dp.shard_range(64 * 8, r, 8) for r in range(8))) == 512
    gs = dp.GradSync(lambda: None, n_buckets=4)
    bk = gs.buckets(106_650_000)
    assert bk[0][0] == 0 and bk[-1][1] == 106_650_000 and all(a[1] == b[0] for a, b in zip(bk, bk[1:])) and len(bk) == 4


def _ref_adam(p, g, m, v, lr, b1=0.9, b2=0.98, eps=1e-9, t=1):
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    p.sub_(lr / (1 - b1 ** t) * m / (v.sqrt() / (1 - b2 ** t) ** 0.5 + eps))


def _sharded_worker(rank, world, port, out_dir, raise_in_step=0):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from mtn_amd import dp
    dp.init_distributed("gloo")
    n = 10_007                                             # not a multiple of 4 * world: exercises the replicated tail
    g0 = torch.Generator().manual_seed(1)
    flat = torch.randn(n, generator=g0)                    # identical replicas
    m, v = torch.zeros(n), torch.zeros(n)
    cuts = [0, 1000, 1003, 6000, n]                         # slices in backward order (last first), one of them 3 elements long
    sh = dp.ShardedOptimizerSync(lambda: flat, lambda: grad, None)
    for t in range(1, 4):
        grad = torch.randn(n, generator=torch.Generator().manual_seed(100 * t + rank))     # this rank's local gradient
        sh.update = lambda off, cnt, t=t: _ref_adam(flat[off:off + cnt], grad[off:off + cnt], m[off:off + cnt], v[off:off + cnt], 1e-2, t=t)
        if t == raise_in_step:
            # a step that raises between reduce_update() and finish(): its first slice's chain is issued, the deferred
            # wait -> update -> gather is pending.  The error path calls abort(); the step is then run again from its gradients.
            keep = grad.clone()
            sh.reduce_update(cuts[-2], cuts[-1])
            assert sh._pending is not None
            sh.abort()
            assert sh._pending is None and not sh._works
            grad = keep                                     # (the reduce-scatter ran in place on the gradient buffer)
        for lo, hi in list(zip(cuts[:-1], cuts[1:]))[::-1]:
            sh.reduce_update(lo, hi)
        sh.finish()
    sh.gather(m); sh.gather(v)
    torch.save({"flat": flat, "m": m, "v": v}, os.path.join(out_dir, f"sh{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("raise_in_step", [0, 2], ids=["clean", "one-step-aborted"])
def test_sharded_optimizer_exchange_equals_allreduce_plus_full_update(tmp_path, raise_in_step):
    """dp.ShardedOptimizerSync (reduce-scatter -> update of the own shard -> all-gather, with a replicated tail) on two gloo
    ranks, three steps: the replicas are BIT-identical (parameters and, after gather(), both moments) and equal to one process
    that sums the two gradients and updates everything.  Second case: step 2 "raises" after its first reduce_update() on both ranks,
    abort() drops the deferred shard update, and the step is repeated — same results (without abort() the stale closure would run
    at the repeated step's first reduce_update() and the slice would be updated twice)."""
    port = _free_port()
    mp.start_processes(_sharded_worker, args=(2, port, str(tmp_path), raise_in_step), nprocs=2, join=True, start_method="spawn")
    r0, r1 = torch.load(tmp_path / "sh0.pt"), torch.load(tmp_path / "sh1.pt")
    for k in ("flat", "m", "v"):
        assert torch.equal(r0[k], r1[k]), k
    n = 10_007
    flat = torch.randn(n, generator=torch.Generator().manual_seed(1))
    m, v = torch.zeros(n), torch.zeros(n)
    for t in range(1, 4):
        g = sum(torch.randn(n, generator=torch.Generator().manual_seed(100 * t + r)) for r in range(2))
        _ref_adam(flat, g, m, v, 1e-2, t=t)
    assert torch.allclose(r0["flat"], flat, rtol=0, atol=1e-6) and torch.allclose(r0["m"], m, atol=1e-6) and torch.allclose(r0["v"], v, atol=1e-6)


def _lp_gather_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from mtn_amd import dp
    dp.init_distributed("gloo")
    n = 10_007
    flat = torch.randn(n, generator=torch.Generator().manual_seed(1))
    lp = flat.to(torch.bfloat16)
    m, v = torch.zeros(n), torch.zeros(n)
    # slices (lo, hi, mat_hi): [matrices | vectors]; one plain fp32 slice (mat_hi None), one whose matrix part is 3 elements long
    slices = [(6000, n, 9001), (1003, 6000, 5500), (1000, 1003, 1003), (0, 1000, None)]

    def upd(off, cnt, t, with_lp):
        _ref_adam(flat[off:off + cnt], grad[off:off + cnt], m[off:off + cnt], v[off:off + cnt], 1e-2, t=t)
        if with_lp:
            lp[off:off + cnt] = flat[off:off + cnt].to(torch.bfloat16)

    sh = dp.ShardedOptimizerSync(lambda: flat, lambda: grad, None, lp_fn=lambda: lp, update_lp=None)
    for t in range(1, 4):
        grad = torch.randn(n, generator=torch.Generator().manual_seed(100 * t + rank))
        sh.update = lambda off, cnt, t=t: upd(off, cnt, t, False)
        sh.update_lp = lambda off, cnt, t=t: upd(off, cnt, t, True)
        for lo, hi, mh in slices:
            sh.reduce_update(lo, hi, mh)
        sh.finish()
        lp[0:1000] = flat[0:1000].to(torch.bfloat16)         # the fp32 slice's copies: a local cast (FusedAdam.refresh_copies(hi))
    before = flat.clone()
    sh.gather(m); sh.gather(v); sh.gather(flat)
    torch.save({"flat": flat, "flat_before_gather": before, "lp": lp, "m": m, "v": v, "calls": dict(sh.calls)}, os.path.join(out_dir, f"lp{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_compute_dtype_gather_keeps_replicas_identical(tmp_path):
    """dp.ShardedOptimizerSync with the compute-dtype gather (round 4): matrices reduce-scattered, the shard's update writes its
    bf16 copy and THAT is gathered; vectors all-reduced and updated everywhere.  Two gloo ranks, three steps: the bf16 copies
    are BIT-identical on both ranks and equal bf16(reference); the fp32 vectors are identical without any gather; the fp32
    masters of the matrices differ between the ranks until gather() (each rank only has its own shards current) and equal
    the one-process reference afterwards, as do the moments."""
    port = _free_port()
    mp.start_processes(_lp_gather_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = torch.load(tmp_path / "lp0.pt"), torch.load(tmp_path / "lp1.pt")
    n = 10_007
    flat = torch.randn(n, generator=torch.Generator().manual_seed(1))
    m, v = torch.zeros(n), torch.zeros(n)
    for t in range(1, 4):
        g = sum(torch.randn(n, generator=torch.Generator().manual_seed(100 * t + r)) for r in range(2))
        _ref_adam(flat, g, m, v, 1e-2, t=t)
    assert torch.equal(r0["lp"], r1["lp"])
    assert torch.equal(r0["lp"].float(), flat.to(torch.bfloat16).float()) or float((r0["lp"].float() - flat).abs().max()) < 2e-2
    for lo, hi in ((9001, n), (5500, 6000), (0, 1000)):                 # vectors + the fp32 slice: consistent all along
        assert torch.equal(r0["flat_before_gather"][lo:hi], r1["flat_before_gather"][lo:hi])
    assert not torch.equal(r0["flat_before_gather"][6000:9001], r1["flat_before_gather"][6000:9001])     # masters of foreign shards were stale
    for k in ("flat", "m", "v"):
        assert torch.equal(r0[k], r1[k]), k
    assert torch.allclose(r0["flat"], flat, rtol=0, atol=1e-6) and torch.allclose(r0["m"], m, atol=1e-6) and torch.allclose(r0["v"], v, atol=1e-6)
    assert r0["calls"]["all_reduce"] > 0
