"""Data-parallel plumbing (reference: train.py:29-35,44,58 = mp.spawn + DDP over NCCL).

One process per GPU, parameters / Adam state replicated (identical seed), each rank its own batch, BatchNorm statistics
local (the reference does not use SyncBatchNorm).  The only exchange per step is the gradient average.

Layout: the gradients live in ONE flat fp32 arena (`p.grad` are views into it), so nothing is copied into buckets.  The arena is
cut into a few BUCKETS by backward *stage*: the model's forward is severed at a handful of activations (`ops.stage_cut`), the
backward pass then runs as S separate `backward()` calls (stage 0 = loss -> last cut, stage s = cut s-1 -> cut s), and after
stage s every parameter that lives downstream of cut s has its final gradient.  Its arena range is all-reduced on a side
stream while the next stage computes - the role DDP's 25 MiB bucket hooks play in the reference (train.py:58), but with
boundaries that survive hipGraph capture: each stage is its own graph, the collectives stay eager launches between replays.

Collective choice (xGMI is point-to-point, 7 links x ~153 GB/s per GPU): every bucket is a single large RCCL all-reduce
(35-55 MB fp32), i.e. large enough that RCCL's multi-ring / direct schedules use all links; the per-link-bound single ring would
need 2*(7/8)*140 MB / 153 GB/s = 1.6 ms per step, a 7-link reduce-scatter + all-gather 0.23 ms.  Either way
only the LAST bucket (encoder + variance adaptor, finished by the final stage) is exposed; the others hide under the remaining
backward stages.  Parameters that received no gradient contribute zeros (needed before var_start_steps).
Device-agnostic: the same code runs on gloo/CPU tensors in tests/test_dp_gloo.py.
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
    """`params`: iterable of parameters, or of (name, parameter) pairs (names are kept for state dicts and stage plans)."""

    def __init__(self, params):
        items = list(params)
        if items and not isinstance(items[0], tuple):
            items = [(str(i), p) for i, p in enumerate(items)]
        named = [(n, p) for n, p in items if p.requires_grad]
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        # index of every trainable tensor in the FULL list that was handed in (frozen ones included): torch.optim.Adam - the reference's
        # and ScheduledOptim's - numbers its state over that full list (model/optimizer.py:8-14, utils/model.py:22-26)
        self.positions = [i for i, (_, p) in enumerate(items) if p.requires_grad]
        self.n_all = len(items)
        self.offsets, n = arena_offsets(self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=self.params[0].device)     # alignment gaps stay zero forever
        self.bind()

    def bind(self):
        """(re)point every p.grad at its arena slice"""
        for p, o in zip(self.params, self.offsets):
            p.grad = _strided_like(self.flat[o:o + p.numel()], p)

    def check_bound(self):
        """raise if some p.grad no longer aliases the arena (e.g. after `zero_grad(set_to_none=True)`): the all-reduce and the fused
        optimizer would silently run on stale zeros otherwise"""
        base = self.flat.data_ptr()
        for n, p, o in zip(self.names, self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != base + 4 * o:
                raise RuntimeError(f"FlatGradArena: the gradient of '{n}' no longer aliases the flat arena (zero_grad(set_to_none=True)?) "
                                   "- call arena.zero_() / arena.bind() instead")

    def zero_(self):
        self.flat.zero_()

    def all_reduce_mean(self, world=None, group=None):
        """blocking path: one all-reduce(sum) / world of the whole arena == DDP gradient averaging."""
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        if world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(world)
        return self.flat

    def ranges_of(self, selected):
        """merged [start, end) arena ranges (in floats, alignment gaps included) covering the parameters whose index is in `selected`"""
        out = []
        sel = sorted(selected)
        ends = self.offsets[1:] + [self.flat.numel()]
        for i in sel:
            a, b = self.offsets[i], ends[i]
            if out and out[-1][1] == a:
                out[-1][1] = b
            else:
                out.append([a, b])
        return [tuple(r) for r in out]


def shard_batch_indices(n_items, rank, world):
    """DistributedSampler-like strided shard (train.py:44): rank r takes items r, r+world, ..."""
    return list(range(rank, n_items, world))


# ---- staged backward + bucketed, overlapped gradient all-reduce -------------------------------------------------------------
def stage_plan(model, n_cuts=3):
    """Where to sever the forward of a CompTransTTS and which parameters are final after each backward stage.

    -> (cut_names, stage_of) : `cut_names` in BACKWARD order (first = closest to the loss); `stage_of(param_name)` = index of the
    stage after which that parameter's gradient is complete (0 .. len(cut_names)).
    The decoder dominates the backward pass (75 % of the FLOPs), so the cuts sit between decoder layers and at the
    decoder input: stage 0 = PostNet + mel_linear + upper decoder layers, ..., last stage = variance adaptor + encoder."""
    dec = model.decoder
    stack_name = "layers" if hasattr(dec, "layers") else "layer_stack"
    L = len(getattr(dec, stack_name))
    n_cuts = max(1, min(int(n_cuts), L))
    # layer indices at whose INPUT the forward is severed, descending; the last cut is always the decoder input
    inner = sorted({(L * k) // n_cuts for k in range(1, n_cuts)} - {0}, reverse=True)
    cut_names = [f"decoder.{stack_name}.{i}" for i in inner] + ["decoder.in"]
    bounds = inner                                    # stage s (< len(inner)) owns decoder layers >= inner[s] (and < inner[s-1])

    def stage_of(name):
        if name.startswith("postnet.") or name.startswith("mel_linear."):
            return 0
        if name.startswith(f"decoder.{stack_name}."):
            li = int(name.split(".")[2])
            for s, lo in enumerate(bounds):
                if li >= lo:
                    return s
            return len(bounds)
        if name.startswith("decoder."):
            # final LayerNorm of the fs2 stack is applied after the last layer (stage 0); everything applied before the first
            # layer (pos_embed_alpha) belongs to the stage that ends at the decoder input
            return 0 if name.startswith("decoder.layer_norm.") else len(bounds)
        return len(bounds) + 1                        # encoder, variance adaptor, speaker embedding: complete after the last stage
    return cut_names, stage_of


class AbiCommunicator:
    """One RCCL communicator per process behind the C ABI (include/ctts.h "Gradient all-reduce"; csrc/comm.hip).  Rank 0 draws the
    unique id, `torch.distributed` (any backend, or nothing for a single rank) carries its 128 bytes to the other ranks, every rank
    creates its communicator on ITS device; `allreduce_mean(t)` averages a dense fp32 CUDA tensor over the ranks in place on the
    current stream (graph-capturable)."""

    def __init__(self, device, group=None):
        import ctypes
        from . import _lib
        self._lib_mod, self._lib = _lib, _lib.load()
        self.device = torch.device(device)
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        uid = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
        if rank == 0:
            _lib.check(self._lib.ctts_comm_unique_id(uid), "ctts_comm_unique_id")
        if world > 1:
            box = [bytes(uid.raw)]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            uid = ctypes.create_string_buffer(box[0], _lib.COMM_ID_BYTES)
        comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.ctts_comm_create(ctypes.byref(comm), world, rank, uid), "ctts_comm_create")
        self.comm, self.world, self.rank = comm, world, rank

    def allreduce_mean(self, t):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.device == self.device
        st = torch.cuda.current_stream(self.device).cuda_stream
        self._lib_mod.check(self._lib.ctts_allreduce_mean(t.data_ptr(), t.numel(), self.comm, st), "ctts_allreduce_mean")

    def close(self):
        if self.comm is not None and self.comm.value:
            torch.cuda.synchronize(self.device)
            self._lib.ctts_comm_destroy(self.comm)
            self.comm = None


class BucketedReducer:
    """All-reduce(sum)/world of the arena, one bucket per backward stage, on a side stream (overlaps the following stages).

    launch(s): call right after backward stage s has been ISSUED on the current stream; finish(): current stream waits for all
    buckets.  With world == 1 both are no-ops - unless `always_reduce` asks for the collectives anyway (a one-rank RCCL group on a
    single GPU: the only way to run the RCCL code path, its stream ordering and ReduceOp.AVG on a 1-GPU box).
    Both calls are capture-safe: under hipGraph capture the side stream forks from and re-joins the capturing stream, so a step
    captured with its collectives inside replays them without any host involvement (trainer.TrainStep graph_collectives)."""

    def __init__(self, arena, stage_of, n_stages, world=None, group=None, always_reduce=False, abi_collective=None):
        """`abi_collective` (None = env CTTS_ABI_COLLECTIVE, default off): route the bucket all-reduces through the library's own C-ABI
        collective (include/ctts.h ctts_comm_create / ctts_allreduce_mean: RCCL bound at run time) instead of torch.distributed - the
        process group is then only the side channel that carries the communicator's unique id.  CUDA arenas only."""
        self.arena, self.group = arena, group
        self._abi_comm = None
        self.world = (dist.get_world_size(group) if dist.is_initialized() else 1) if world is None else int(world)
        self.active = self.world > 1 or (bool(always_reduce) and dist.is_initialized())
        by_stage = [[] for _ in range(n_stages)]
        for i, n in enumerate(arena.names):
            by_stage[stage_of(n)].append(i)
        self.ranges = [arena.ranges_of(ix) for ix in by_stage]
        self.is_cuda = arena.flat.is_cuda
        self.comm = torch.cuda.Stream(device=arena.flat.device) if (self.is_cuda and self.active) else None
        self.launched = 0
        if abi_collective is None:
            import os
            abi_collective = os.environ.get("CTTS_ABI_COLLECTIVE", "0") == "1"
        if abi_collective and self.active and self.is_cuda:
            self._abi_comm = AbiCommunicator(arena.flat.device, group)

    def bucket_bytes(self):
        return [sum(b - a for a, b in r) * 4 for r in self.ranges]

    def _reduce(self, s):
        # one collective per contiguous range (the stage plans of both block types give exactly ONE range per bucket); on RCCL the
        # division by the world size rides inside the collective (ReduceOp.AVG), gloo has no AVG: sum, then scale
        if self._abi_comm is not None:
            for a, b in self.ranges[s]:
                self._abi_comm.allreduce_mean(self.arena.flat[a:b])
            return
        avg = self.is_cuda and dist.get_backend(self.group) == "nccl"
        inv = 1.0 / self.world
        for a, b in self.ranges[s]:
            seg = self.arena.flat[a:b]
            if avg:
                dist.all_reduce(seg, op=dist.ReduceOp.AVG, group=self.group)
            else:
                dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group)
                seg.mul_(inv)

    def launch(self, s):
        self.launched += 1
        if not self.active or not self.ranges[s]:
            return
        if self.comm is None:
            self._reduce(s)
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(ev)
            if self._timing is not None:
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record(self.comm)
                self._reduce(s)
                t1.record(self.comm)
                self._timing["buckets"].append((s, t0, t1))
            else:
                self._reduce(s)

    def finish(self):
        assert self.launched == len(self.ranges), f"BucketedReducer: {self.launched} of {len(self.ranges)} stages were launched"
        self.launched = 0
        if self.comm is not None:
            if self._timing is not None:
                # exposed communication = how long the compute stream stalls here for the side stream (everything before this point ran
                # under the backward stages): two events around the wait on the compute stream
                cur = torch.cuda.current_stream()
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record(cur)
                cur.wait_stream(self.comm)
                t1.record(cur)
                self._timing["exposed"].append((t0, t1))
                return
            torch.cuda.current_stream().wait_stream(self.comm)

    # ---- per-bucket timing (bench.py --gpus N: the first real SCALE run should explain itself)
    _timing = None

    def start_timing(self):
        """from now on every launch() / finish() is bracketed by timing events (eager collectives between stage replays only: the
        whole-step graph of graph_collectives has no host-side launch points)"""
        self._timing = {"buckets": [], "exposed": []}

    def stop_timing(self):
        """-> {"allreduce_ms_per_bucket": [...], "exposed_ms_per_step": x, "steps": n} averaged over the steps since start_timing();
        synchronises the device"""
        t, self._timing = self._timing, None
        if not t or not t["exposed"]:
            return None
        torch.cuda.synchronize()
        n = len(t["exposed"])
        per = [0.0] * len(self.ranges)
        for s, a, b in t["buckets"]:
            per[s] += a.elapsed_time(b)
        return {"allreduce_ms_per_bucket": [v / n for v in per], "bucket_bytes": self.bucket_bytes(),
                "exposed_ms_per_step": sum(a.elapsed_time(b) for a, b in t["exposed"]) / n, "steps": n,
                "note": "all-reduce time of a bucket is measured on the reducer's side stream (it overlaps the next backward stage); "
                        "exposed = stall of the compute stream at the join in front of the optimizer"}


class FlatAdam:
    """`clip_grad_norm_(params, max_norm)` + `torch.optim.Adam.step()` (train.py:118-125, model/optimizer.py:22-53) as ONE fused
    streaming update (csrc/optim.hip): parameters are re-homed into a flat fp32 arena (the tensors torch sees become
    views, state_dict()/load_state_dict() keep working), the gradients are the FlatGradArena, the moments are flat too.
    `lr` is a device scalar tensor (share ScheduledOptim's capturable lr - `ScheduledOptim.lr_tensor` - so the Noam schedule keeps
    driving it); the step counter lives on the device, so `step()` replays inside a hipGraph.

    `state_dict()` / `load_state_dict()` speak torch.optim.Adam's format over the FULL parameter list the arena was built from
    (state: {index: {step, exp_avg, exp_avg_sq}}, param_groups[0]["params"] = 0 .. n_all-1, indices = positions in
    `model.parameters()`, frozen tensors included but without a state entry - exactly what torch writes for a parameter that never
    received a gradient), so the optimizer half of a reference checkpoint (`train.py:190-200`: {"model": ...,
    "optimizer": adam.state_dict()}) loads here and a checkpoint written here loads into `torch.optim.Adam(model.parameters())`."""

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
        from ._lib import ADAM_STATE_FLOATS
        self.state = torch.zeros(ADAM_STATE_FLOATS, dtype=torch.float32, device=dev)   # {sum g^2, step count, last total norm, partials}
        self.state[1] = float(current_step)

    def step(self):
        self._k.WEIGHTS_EPOCH[0] += 1          # the update below bypasses autograd's version counters: weight-derived caches are stale
        self._k.adam_clip_step(self.flat_param, self.arena.flat, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps,
                               self.weight_decay, self.max_norm, self.state)

    @property
    def total_norm(self):
        return self.state[2]

    # ---- checkpointing in torch.optim.Adam's format
    def state_dict(self):
        step = float(self.state[1])
        st = {}
        for pos, p, o in zip(self.arena.positions, self.arena.params, self.arena.offsets):
            sl = slice(o, o + p.numel())
            st[pos] = {"step": torch.tensor(step), "exp_avg": _strided_like(self.m[sl], p).detach().clone(),
                       "exp_avg_sq": _strided_like(self.v[sl], p).detach().clone()}
        group = {"lr": float(self.lr), "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "params": list(range(self.arena.n_all))}
        return {"state": st, "param_groups": [group]}

    def load_state_dict(self, sd, param_index=None):
        """Accepts a state dict of an Adam built over ALL `model.parameters()` (the reference's, ScheduledOptim's, this class's own):
        arena tensor i reads state[arena.positions[i]].  A dict whose param group lists only the trainable tensors (the round-2 format
        of this class) is recognised by its length.  `param_index` overrides the mapping (arena position -> checkpoint index)."""
        n = len(self.arena.params)
        n_ckpt = len(sd["param_groups"][0]["params"])
        if param_index is not None:
            idx = list(param_index)
        elif n_ckpt == self.arena.n_all:
            idx = list(self.arena.positions)
        elif n_ckpt == n:
            idx = list(range(n))
        else:
            raise ValueError(f"FlatAdam.load_state_dict: the checkpoint's optimizer covers {n_ckpt} parameters, this model has "
                             f"{self.arena.n_all} ({n} trainable)")
        if len(idx) != n:
            raise ValueError(f"FlatAdam.load_state_dict: need {n} parameter indices, got {len(idx)}")
        steps = set()
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(self.arena.params, self.arena.offsets)):
                s = sd["state"].get(idx[i])
                sl = slice(o, o + p.numel())
                if s is None:                                           # parameter never stepped in the checkpoint
                    self.m[sl].zero_(); self.v[sl].zero_()
                    continue
                if tuple(s["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"FlatAdam.load_state_dict: state {idx[i]} has shape {tuple(s['exp_avg'].shape)}, parameter "
                                     f"'{self.arena.names[i]}' has {tuple(p.shape)}")
                _strided_like(self.m[sl], p).copy_(s["exp_avg"])
                _strided_like(self.v[sl], p).copy_(s["exp_avg_sq"])
                steps.add(float(s["step"]))
            if len(steps) > 1:
                raise ValueError(f"FlatAdam.load_state_dict: per-parameter step counts differ ({sorted(steps)}); the fused update keeps one")
            if steps:
                self.state[1] = steps.pop()
        g = sd["param_groups"][0]
        self.betas, self.eps, self.weight_decay = tuple(g["betas"]), g["eps"], g["weight_decay"]
