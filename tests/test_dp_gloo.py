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


@__import__("pytest").mark.parametrize("world", [2, 4, 8])
@__import__("pytest").mark.parametrize("order", ["strided", "snake"])
def test_strong_scaling_shard_covers_the_global_batch_once(world, order):
    """bench.py --scaling strong: the canonical 16 utterances dealt over 2 / 4 / 8 ranks (2 per rank at 8: BatchNorm still sees two
    utterances); every utterance lands on exactly one rank, widths shrink to the shard's own maxima, and the snake order is the
    better balanced one"""
    from ctts_amd.synthetic import make_batch, shard, shard_indices, shard_valid_frames
    gb = make_batch()
    B = gb["texts"].shape[0]
    seen = []
    for r in range(world):
        idx = shard_indices(gb["mel_lens"], r, world, order)
        assert len(idx) == B // world
        sh = shard(gb, r, world, order)
        assert sh["texts"].shape == (B // world, int(gb["src_lens"][idx].max()))
        assert sh["mels"].shape[:2] == (B // world, int(gb["mel_lens"][idx].max())) and sh["max_mel_len"] == sh["mels"].shape[1]
        assert torch.equal(sh["mel_lens"], gb["mel_lens"][idx]) and torch.equal(sh["d_targets"], gb["d_targets"][idx][:, :sh["max_src_len"]])
        seen += idx
    assert sorted(seen) == list(range(B))
    v_str, v_snk = shard_valid_frames(gb, world, "strided"), shard_valid_frames(gb, world, "snake")
    assert sum(v_str) == sum(v_snk) == int(gb["mel_lens"].sum())
    assert max(v_snk) <= max(v_str)
    if world == 8:
        assert max(v_str) / (sum(v_str) / 8) > 1.10 and max(v_snk) / (sum(v_snk) / 8) < 1.05


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


# ---- the product's arena / stage plan / bucketed reducer over the REAL CompTransTTS parameter set --------------------------------
def _real_model(block="transformer_fs2"):
    import ctts_amd
    from ctts_amd.configs import get_configs
    pre, mc, tc = get_configs()
    mc["block_type"] = block
    torch.manual_seed(1234)
    return ctts_amd.CompTransTTS(pre, mc, tc)


def _standin_grad(n, rank):
    """deterministic per-rank stand-in for a backward pass (the product's kernels do not run on the CPU)"""
    i = torch.arange(n, dtype=torch.float32)
    return torch.sin(i * 1e-3) * (rank + 1) + (rank == 1) * 0.25


def _real_worker(rank, world, port, q):
    from ctts_amd.dp import BucketedReducer, stage_plan
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _real_model()
    arena = FlatGradArena(m.named_parameters())
    cuts, stage_of = stage_plan(m, 3)
    red = BucketedReducer(arena, stage_of, len(cuts) + 1)
    arena.flat.copy_(_standin_grad(arena.flat.numel(), rank))
    if rank == 1:      # a parameter that received no gradient on this rank contributes zeros
        m.variance_adaptor.energy_predictor.linear.weight.grad.zero_()
    for s in range(len(cuts) + 1):     # backward order: stage 0 first
        red.launch(s)
    red.finish()
    probe = {n: p.grad.flatten()[:5].clone() for n, p in zip(arena.names, arena.params)
             if n in ("postnet.convolutions.4.0.conv.weight", "decoder.layers.3.op.ffn.ffn_1.weight", "decoder.pos_embed_alpha",
                      "encoder.embed_tokens.weight", "variance_adaptor.energy_predictor.linear.weight")}
    q.put((rank, float(arena.flat.double().sum()), probe, red.bucket_bytes()))
    dist.barrier()
    dist.destroy_process_group()


@__import__("pytest").mark.parametrize("world", [2, 4, 8])
def test_real_model_bucketed_allreduce(world):
    """world 2, 4 and 8 (the node size BASELINE names): stage plan + one collective per bucket over the real parameter set"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_real_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, tot, probe, bb = q.get(timeout=300)
        res[r] = (tot, probe, bb)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process expectation on the same arena layout
    m = _real_model()
    arena = FlatGradArena(m.named_parameters())
    n = arena.flat.numel()
    arena.flat.copy_(_standin_grad(n, 1))
    m.variance_adaptor.energy_predictor.linear.weight.grad.zero_()
    g1 = arena.flat.clone()
    mask = torch.zeros(n, dtype=torch.bool)                    # reduced positions = parameter storage + its alignment tail
    ends = arena.offsets[1:] + [n]
    for o, e in zip(arena.offsets, ends):
        mask[o:e] = True
    assert bool(mask.all())                                    # the buckets tile the whole arena
    expect = g1.clone()
    for r in range(world):
        if r != 1:
            expect += _standin_grad(n, r)
    expect /= world
    assert abs(res[0][0] - float(expect.double().sum())) < 1e-3 * max(1.0, abs(float(expect.double().sum())))
    assert all(res[r][0] == res[0][0] for r in range(world))
    arena.flat.copy_(expect)
    for name, v in res[0][1].items():
        p = dict(zip(arena.names, arena.params))[name]
        assert torch.allclose(v, p.grad.flatten()[:5], atol=1e-5), name
        assert all(torch.equal(v, res[r][1][name]) for r in range(1, world)), name
    bb = res[0][2]
    assert len(bb) == 4 and sum(bb) == 4 * n and min(bb) > 20e6 and max(bb) < 60e6     # ~25-55 MB buckets, nothing dropped


@__import__("pytest").mark.parametrize("block", ["transformer_fs2", "conformer"])
def test_stage_plan_partitions_parameters_in_backward_order(block):
    from ctts_amd.dp import stage_plan, BucketedReducer
    m = _real_model(block)
    arena = FlatGradArena(m.named_parameters())
    cuts, stage_of = stage_plan(m, 3)
    assert cuts[-1] == "decoder.in" and len(cuts) == 3
    st = {n: stage_of(n) for n in arena.names}
    assert st["postnet.convolutions.0.0.conv.weight"] == 0 and st["mel_linear.weight"] == 0
    assert all(v == 3 for k, v in st.items() if k.startswith("encoder.") or k.startswith("variance_adaptor."))
    stack = "layers" if block == "transformer_fs2" else "layer_stack"
    lay = {int(k.split(".")[2]): v for k, v in st.items() if k.startswith(f"decoder.{stack}.")}
    assert [lay[i] for i in range(6)] == [2, 2, 1, 1, 0, 0]
    red = BucketedReducer(arena, stage_of, 4, world=1)
    covered = sorted(r for rs in red.ranges for r in rs)
    assert covered[0][0] == 0 and covered[-1][1] == arena.flat.numel()
    assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))          # contiguous, disjoint, complete


def test_flat_adam_state_dict_roundtrips_in_torch_adam_format():
    from ctts_amd.dp import FlatAdam
    m = _model()
    arena = FlatGradArena(m.named_parameters())
    fa = FlatAdam(arena, 1e-3, current_step=7)
    fa.m.copy_(torch.arange(fa.m.numel(), dtype=torch.float32) * 0.5)
    fa.v.copy_(torch.arange(fa.v.numel(), dtype=torch.float32) * 0.25)
    sd = fa.state_dict()
    # torch.optim.Adam accepts it as its own state
    ref = torch.optim.Adam(arena.params, lr=1e-3, betas=(0.9, 0.98), eps=1e-9)
    ref.load_state_dict(sd)
    assert float(ref.state[arena.params[0]]["step"]) == 7.0
    assert torch.equal(ref.state[arena.params[1]]["exp_avg"], sd["state"][1]["exp_avg"])
    m2 = _model()
    fb = FlatAdam(FlatGradArena(m2.named_parameters()), 1e-3)
    fb.load_state_dict(ref.state_dict())
    for p, o in zip(arena.params, arena.offsets):             # per parameter (the alignment gaps between them carry no state)
        sl = slice(o, o + p.numel())
        assert torch.equal(fb.m[sl], fa.m[sl]) and torch.equal(fb.v[sl], fa.v[sl])
    assert float(fb.state[1]) == 7.0
    # parameters were re-homed into the flat arena without changing their values or the module's state_dict
    for (k, a), (_, b) in zip(m.state_dict().items(), _model().state_dict().items()):
        assert torch.equal(a, b), k


def test_flat_adam_state_dict_is_keyed_like_adam_over_all_parameters_of_the_real_model():
    """The reference builds Adam over ALL model.parameters() (model/optimizer.py:8-14): the frozen variance_adaptor.energy_bins sits at
    position 43 of 171, so every later index differs from an enumeration of the trainable tensors.  FlatAdam's state dict must load
    into torch.optim.Adam(list(model.parameters())) and back."""
    from ctts_amd.dp import FlatAdam
    m = _real_model()
    allp = list(m.named_parameters())
    frozen = [i for i, (_, p) in enumerate(allp) if not p.requires_grad]
    assert frozen and frozen[0] < len(allp) - 1                      # a frozen tensor in the MIDDLE of the list
    arena = FlatGradArena(m.named_parameters())
    assert arena.n_all == len(allp) and len(arena.positions) == len(allp) - len(frozen)
    fa = FlatAdam(arena, 1e-3, current_step=11)
    fa.m.copy_(torch.arange(fa.m.numel(), dtype=torch.float32) % 997 * 1e-3)
    fa.v.copy_(torch.arange(fa.v.numel(), dtype=torch.float32) % 991 * 1e-4)
    sd = fa.state_dict()
    assert sd["param_groups"][0]["params"] == list(range(len(allp))) and all(i not in sd["state"] for i in frozen)
    ref = torch.optim.Adam([p for _, p in allp], lr=1e-3, betas=(0.9, 0.98), eps=1e-9)
    ref.load_state_dict(sd)                                          # torch accepts it as its own
    from ctts_amd.dp import _strided_like
    name_of = {id(p): n for n, p in allp}
    checked = 0
    for p, st in ref.state.items():
        i = arena.names.index(name_of[id(p)])
        o = arena.offsets[i]
        assert torch.equal(st["exp_avg"], _strided_like(fa.m[o:o + p.numel()], p)), arena.names[i]
        assert torch.equal(st["exp_avg_sq"], _strided_like(fa.v[o:o + p.numel()], p)), arena.names[i]
        checked += 1
    assert checked == len(arena.params)
    # and back: a checkpoint written by torch's Adam over all parameters loads without a hand-built index map
    m2 = _real_model()
    fb = FlatAdam(FlatGradArena(m2.named_parameters()), 1e-3)
    fb.load_state_dict(ref.state_dict())
    for p, o in zip(arena.params, arena.offsets):
        sl = slice(o, o + p.numel())
        assert torch.equal(fb.m[sl], fa.m[sl]) and torch.equal(fb.v[sl], fa.v[sl])
    assert float(fb.state[1]) == 11.0


def test_zero_grad_set_to_none_is_detected():
    import pytest
    m = _model()
    arena = FlatGradArena(m.named_parameters())
    arena.check_bound()
    torch.optim.SGD(m.parameters(), lr=0.1).zero_grad(set_to_none=True)
    with pytest.raises(RuntimeError, match="no longer aliases"):
        arena.check_bound()
    arena.bind()
    arena.check_bound()
