"""Data-parallel plumbing (reference: train.py:29-35,44,58 = mp.spawn + DDP over NCCL).

One process per GPU, parameters / Adam state / BN running stats replicated (identical seed), each
rank its own batch, BatchNorm statistics local (the reference does not use SyncBatchNorm).  The only
exchange per step is the gradient average.  Instead of DDP's 25 MiB bucket copies the gradients live
in ONE flat fp32 arena (`p.grad` are views into it), reduced by a single RCCL all-reduce over xGMI
(140.4 MB for fs2) and consumed in place by the fused clip + Adam.  Parameters that received no
gradient contribute zeros (needed before var_start_steps).  Device-agnostic: the same
code runs on gloo/CPU tensors in tests/test_dp_gloo.py.
"""
import torch
import torch.distributed as dist


ARENA_ALIGN = 64          # floats: every tensor starts on a 256-byte boundary (the GEMM loaders need 16-byte aligned operands)


def arena_offsets(params):
    offs, o = [], 0
    for p in params:
        offs.append(o)
        o += (p.numel() + ARENA_ALIGN - 1) // ARENA_ALIGN * ARENA_ALIGN
    return offs, o


def _strided_like(flat_slice, p):
    """view of a flat slice with p's shape AND strides (parameters may be dense permuted views, e.g. the GEMM-major Conv1d weights)"""
    if p.is_contiguous():
        return flat_slice.view_as(p)
    return flat_slice.as_strided(p.size(), p.stride())


class FlatGradArena:
    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.offsets, n = arena_offsets(self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=self.params[0].device)     # alignment gaps stay zero forever
        for p, o in zip(self.params, self.offsets):
            p.grad = _strided_like(self.flat[o:o + p.numel()], p)

    def zero_(self):
        self.flat.zero_()

    def all_reduce_mean(self, world=None, group=None):
        """all-reduce(sum) / world == DDP gradient averaging."""
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        if world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(world)
        return self.flat


def shard_batch_indices(n_items, rank, world):
    """DistributedSampler-like strided shard (train.py:44): rank r takes items r, r+world, ..."""
    return list(range(rank, n_items, world))


class FlatAdam:
    """`clip_grad_norm_(params, max_norm)` + `torch.optim.Adam.step()` (train.py:118-125, model/optimizer.py:22-53) as ONE fused
    streaming update (csrc/optim.hip): parameters are re-homed into a flat fp32 arena (the tensors torch sees become
    views, state_dict()/load_state_dict() keep working), the gradients are the FlatGradArena, the moments are flat too.
    `lr` is a device scalar tensor (share ScheduledOptim's capturable lr so the Noam schedule keeps driving it); the step counter
    lives on the device, so `step()` replays inside a hipGraph."""

    def __init__(self, grad_arena, lr, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.0, max_norm=1.0, current_step=0):
        from . import kernels
        self._k = kernels
        self.arena = grad_arena
        params = grad_arena.params
        dev = params[0].device
        n = grad_arena.flat.numel()
        self.flat_param = torch.zeros(n, dtype=torch.float32, device=dev)      # same (aligned) layout as the gradient arena
        with torch.no_grad():
            for p, o in zip(params, grad_arena.offsets):
                v = _strided_like(self.flat_param[o:o + p.numel()], p)
                v.copy_(p.detach())
                p.data = v
        self.m = torch.zeros_like(self.flat_param)
        self.v = torch.zeros_like(self.flat_param)
        self.lr = lr if torch.is_tensor(lr) else torch.tensor(float(lr), dtype=torch.float32, device=dev)
        self.betas, self.eps, self.weight_decay, self.max_norm = betas, eps, weight_decay, max_norm
        self.state = torch.zeros(3, dtype=torch.float32, device=dev)          # {sum g^2 accumulator, step count, last total norm}
        self.state[1] = float(current_step)

    def step(self):
        self._k.adam_clip_step(self.flat_param, self.arena.flat, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps,
                               self.weight_decay, self.max_norm, self.state)

    @property
    def total_norm(self):
        return self.state[2]
