"""World-size-2 gloo tests (CPU) of the multi-GPU layer: frame sharding without collectives, single-bucket
gradient all-reduce, parameter broadcast, max-over-ranks timing (the N>1 path of bench.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "second.pytorch_amd"))
    from second_amd import distributed as D
    r, lr, w = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                      # different init per rank
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.BatchNorm1d(16), torch.nn.Linear(16, 2))
    D.broadcast_parameters(net, 0)
    params = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    gathered = [torch.zeros_like(params) for _ in range(world)]
    dist.all_gather(gathered, params)
    same_after_bcast = all(torch.equal(gathered[0], g) for g in gathered)
    # per-rank shard of a global batch -> local grads -> one all-reduce == grads of the global batch
    torch.manual_seed(7)
    x, y = torch.randn(12, 8), torch.randn(12, 2)
    lo, hi = D.shard_frames(12, rank, world)
    net.zero_grad()
    net[1].eval()                                       # BN in eval: the loss decomposes over samples
    ((net(x[lo:hi]) - y[lo:hi]) ** 2).sum().backward()
    nbytes = D.allreduce_gradients(net, average=False)
    g_dist = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    net.zero_grad()
    ((net(x) - y) ** 2).sum().backward()
    g_full = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    tmax = D.max_over_ranks(float(rank + 1))
    # a parameter without a gradient on ONE rank (unused head / empty shard) must not change the bucket layout
    two = torch.nn.ModuleDict({"trunk": torch.nn.Linear(4, 4), "head_a": torch.nn.Linear(4, 1), "head_b": torch.nn.Linear(4, 1)})
    D.broadcast_parameters(two, 0)
    two.zero_grad(set_to_none=True)
    xin = torch.ones(3, 4)
    head = two["head_a"] if rank == 0 else two["head_b"]          # each rank exercises a different head
    head(two["trunk"](xin)).sum().backward()
    assert (two["head_b"].weight.grad is None) == (rank == 0)
    D.allreduce_gradients(two, average=True)
    ok_missing = all(p.grad is not None for p in two.parameters())
    flat = torch.cat([p.grad.reshape(-1) for p in two.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    ok_missing = ok_missing and all(torch.equal(gathered[0], g_) for g_ in gathered)
    # head_a only got a gradient on rank 0: its mean over 2 ranks is half of rank 0's local gradient (3 * trunk output / 2)
    ok_missing = ok_missing and torch.allclose(two["head_a"].bias.grad, torch.tensor([1.5]))
    q.put((rank, same_after_bcast, torch.allclose(g_dist, g_full, rtol=1e-5, atol=1e-6) and ok_missing, nbytes, tmax, (lo, hi)))
    dist.destroy_process_group()


def test_two_rank_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(30) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, same, grads_ok, nbytes, tmax, _ in res:
        assert same and grads_ok and nbytes > 0 and tmax == 2.0
    assert [r[5] for r in res] == [(0, 6), (6, 12)]


def test_shard_frames_is_a_partition():
    from second_amd.distributed import shard_frames
    for n in (0, 1, 7, 8, 64, 1001):
        for w in (1, 2, 3, 8):
            spans = [shard_frames(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
