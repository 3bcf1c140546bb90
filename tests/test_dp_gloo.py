"""CPU, world_size 2, gloo: the flat-arena gradient all-reduce reproduces single-process gradients
on the global batch (DDP semantics), including parameters that receive no gradient on a rank."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ctts_amd.dp import FlatGradArena, shard_batch_indices


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(7)
    return torch.nn.Sequential(torch.nn.Linear(12, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3), torch.nn.Linear(3, 3))


def _loss(m, x, y, use_last):
    h = m[2](m[1](m[0](x)))
    if use_last:
        h = m[3](h)
    return ((h - y) ** 2).mean()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _model()
    arena = FlatGradArena(m.parameters())
    g = torch.Generator().manual_seed(3)
    X, Y = torch.randn(8, 12, generator=g), torch.randn(8, 3, generator=g)
    idx = shard_batch_indices(8, rank, world)
    arena.zero_()
    # rank 1 does not use the last layer -> its gradient there is all zeros on that rank
    _loss(m, X[idx], Y[idx], use_last=(rank == 0)).backward()
    arena.all_reduce_mean()
    q.put((rank, torch.cat([p.grad.flatten() for p in arena.params])))      # the arena's per-tensor views, alignment gaps dropped
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_equals_single_process_average():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.equal(res[0], res[1])                      # every rank ends with the same averaged gradient
    # single-process reference: average of the two per-rank losses' gradients
    m = _model()
    g = torch.Generator().manual_seed(3)
    X, Y = torch.randn(8, 12, generator=g), torch.randn(8, 3, generator=g)
    ref = None
    for r in range(world):
        m.zero_grad()
        idx = shard_batch_indices(8, r, world)
        _loss(m, X[idx], Y[idx], use_last=(r == 0)).backward()
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in m.parameters()])
        ref = flat if ref is None else ref + flat
    ref = ref / world
    assert torch.allclose(res[0], ref, atol=1e-6), (res[0] - ref).abs().max()
    assert shard_batch_indices(7, 1, 2) == [1, 3, 5]


def test_arena_views_alias_param_grads():
    m = _model()
    arena = FlatGradArena(m.parameters())
    m(torch.ones(2, 12)).sum().backward()
    assert arena.flat.abs().sum() > 0
    for p, o in zip(arena.params, arena.offsets):
        assert o % 64 == 0                                   # 256-byte aligned starts (16-byte aligned GEMM operands)
        assert p.grad.data_ptr() == arena.flat[o:o + p.numel()].data_ptr()
    used = sum(p.numel() for p in arena.params)
    assert abs(float(arena.flat.sum()) - float(sum(p.grad.sum() for p in arena.params))) < 1e-4 and arena.flat.numel() >= used
