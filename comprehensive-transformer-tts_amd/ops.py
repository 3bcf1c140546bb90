"""Autograd operators of the hot path.  Every forward AND backward runs hand-written gfx950
kernels from libctts_hip.so through `kernels.py`; torch is used for device memory, the
autograd graph and the stream only.

Layout is [B, T, C] (C contiguous) everywhere: Conv1d over time is an implicit GEMM whose A
operand is the activation itself with overlapping rows (row stride = C_in, K = k*C_in), so the
reference's transpose/contiguous copies (10.6 % of its CPU time)
disappear and bias / scale / activation / dropout / residual / pad-mask fuse into the GEMM epilogue.
"""
import math

import torch

from . import kernels as K
from .kernels import ACT_NONE, ACT_RELU, ACT_GELU, ACT_TANH, ACT_SWISH  # noqa: F401


_FUSE_GRAD_ACCUM = False


def set_grad_accumulation_fusion(flag):
    """When on, weight / bias / LayerNorm gradients are accumulated by the kernels straight into an existing,
    contiguous `param.grad` (e.g. the flat DP arena) and autograd gets `None` for them: no zero-fill, no temporary,
    no accumulate-add launch per parameter (~320 tiny launches per fs2 train step).  Off by default: plain autograd
    semantics (hooks on parameter gradients, double backward) need the unfused path."""
    global _FUSE_GRAD_ACCUM
    _FUSE_GRAD_ACCUM = bool(flag)


def grad_accumulation_fusion():
    return _FUSE_GRAD_ACCUM


def _grad_of(p):
    """The buffer a kernel may ACCUMULATE the gradient of `p` into when gradient-accumulation fusion is on, or None: `p.grad` of a leaf,
    or - for a dense reshaping view of a leaf, e.g. `conv.weight.view(Cout, Cin)` of a pointwise convolution - the same view of the
    leaf's gradient (autograd then gets None for the view, so nothing is added twice)."""
    if not _FUSE_GRAD_ACCUM or p is None or not p.requires_grad:
        return None
    if p.is_leaf:
        g = p.grad
        if g is None or g.dtype != torch.float32 or not (g.is_contiguous() or g.stride() == p.stride()):
            return None
        return g
    base = p._base
    if (base is None or not base.is_leaf or base.grad is None or base.grad.dtype != torch.float32 or p.numel() != base.numel()
            or not p.is_contiguous()):
        return None
    if base.is_contiguous() and base.grad.is_contiguous():
        return base.grad.view(p.shape)
    # a dense PERMUTED leaf (GEMM-major Conv2d weight: shape [Cout,Cin,3,3], memory [Cout][kh][kw][Cin]) seen through the view that
    # walks its memory in order: the same walk over the gradient, which the flat arena keeps with the leaf's own strides
    g = base.grad
    dense = sum((n - 1) * st for n, st in zip(base.shape, base.stride())) + 1 == base.numel()
    if dense and g.stride() == base.stride() and p.data_ptr() == base.data_ptr():
        return torch.as_strided(g, p.shape, p.stride(), g.storage_offset())
    return None


def _fusable(p):
    return _grad_of(p) is not None


# ---- staged backward (dp.py): named cut points in the forward ---------------------------------------------------------------
_CUTS = None


class CutRecorder:
    """While active, `stage_cut(x, name)` with `name` in `names` severs the autograd graph at x: the forward continues on a detached
    leaf copy, and (x, leaf) is recorded so that the backward pass can be run in stages - `loss.backward()` stops at the leaves,
    `x.backward(leaf.grad)` continues upstream.  Each stage can then be captured as its own hipGraph with the gradient all-reduce
    of the finished parameters launched between the replays (dp.BucketedReducer)."""

    def __init__(self, names):
        self.names = list(names)
        self.pairs = {}

    def __enter__(self):
        global _CUTS
        assert _CUTS is None, "CutRecorder is not re-entrant"
        _CUTS = self
        self.pairs = {}
        return self

    def __exit__(self, *exc):
        global _CUTS
        _CUTS = None
        return False

    def backward_stages(self, loss):
        """generator: runs backward stage 0 (from `loss`), yields 0, then stage s for every cut in backward order, yielding s."""
        missing = [n for n in self.names if n not in self.pairs]
        if missing:
            raise RuntimeError(f"staged backward: the forward never reached the cut point(s) {missing}")
        loss.backward()
        yield 0
        for s, name in enumerate(self.names, 1):
            xs, gs = [], []
            for x, leaf in self.pairs[name]:
                g, leaf.grad = leaf.grad, None
                if g is not None:                    # a severed tensor the loss does not use (e.g. before var_start_steps)
                    xs.append(x)
                    gs.append(g)
            if not xs:
                raise RuntimeError(f"staged backward: no gradient arrived at cut '{name}'")
            torch.autograd.backward(xs, gs)          # ONE traversal of the upstream graph for all tensors of the cut
            yield s
        self.pairs = {}


def stage_cut(x, name):
    """identity unless a CutRecorder asks for `name` (and x is part of an autograd graph)"""
    rec = _CUTS
    if rec is None or name not in rec.names or not x.requires_grad:
        return x
    leaf = x.detach().requires_grad_(True)
    rec.pairs.setdefault(name, []).append((x, leaf))
    return leaf


def stage_cut_tree(obj, name):
    """`stage_cut` for EVERY differentiable tensor inside a nested tuple / list / dict (a module boundary that several tensors
    cross: the variance adaptor hands the decoder input AND the predictions the loss reads).  All of them are severed under one
    name, so the region upstream is back-propagated exactly once, from all of them together.  Containers without a differentiable
    tensor keep their identity (the reference mutates and returns the caller's `p_targets` dict)."""
    rec = _CUTS
    if rec is None or name not in rec.names:
        return obj

    def walk(o):
        if torch.is_tensor(o):
            return stage_cut(o, name) if (o.requires_grad and o.is_floating_point()) else o
        if isinstance(o, dict):
            new = {k: walk(v) for k, v in o.items()}
            return new if any(new[k] is not o[k] for k in o) else o
        if isinstance(o, (tuple, list)):
            new = [walk(v) for v in o]
            if all(a is b for a, b in zip(new, o)):
                return o
            return type(o)(new)
        return o
    return walk(obj)


def _gemm_major(w):
    """[Cout, K*Cin] view of a Conv1d weight whose memory is already [Cout][K][Cin] (model._Conv), else None"""
    if w.dim() != 3:
        return None
    N, Cin, k = w.shape
    if w.stride() == (k * Cin, 1, Cin) or (k == 1 and w.is_contiguous()):
        return w.permute(0, 2, 1).view(N, k * Cin)
    return None


# ---- weight-gradient side stream -------------------------------------------------------------------------------
# dgrad and wgrad of a layer both consume dZ and are otherwise independent.  With a side stream set (and gradient
# accumulation fusion on, so the wgrad kernels write param.grad themselves and autograd never touches the result),
# every wgrad GEMM is issued on that stream: its workgroups fill the partially occupied last rounds ("tails") of the
# dgrad GEMMs running on the main stream, and vice versa.  The streams are re-joined by an autograd engine callback
# at the end of the backward pass, so callers (and hipGraph capture) see ordinary single-stream semantics.
_WGRAD = {"stream": None, "keep": [], "main": None, "max_rows": None}


# (With a side stream set the weight gradients are NOT deferred to the stage's PartialSink: their partials would be written on one stream
#  and summed on another.)
def set_wgrad_stream(stream, max_rows=None):
    """`stream`: a torch.cuda.Stream for weight-gradient GEMMs, or None for single-stream execution.  `max_rows`: only layers whose
    reduction (rows of the activation matrix) is at most this long use it - the under-filled launches of the phoneme-level layers
    (2,048 rows: 128 output tiles on 256 CUs), where the data- and the weight-gradient GEMM can share the chip; None = every layer.
    Measured under hipGraph replay (round 3, same box): fs2 23.87 ms single-stream, 24.49 ms with max_rows=4096, 25.12 ms for every
    layer; conformer 28.8 vs 29.85 ms - every fork / join of the captured graph costs more than the overlap returns, so the train
    step leaves it off."""
    _WGRAD["stream"] = stream
    _WGRAD["max_rows"] = max_rows


def _wgrad_join():
    side, main = _WGRAD["stream"], _WGRAD["main"]
    if side is not None and main is not None:
        main.wait_stream(side)
    _WGRAD["keep"].clear()          # operands read by the side stream may be recycled only after the join
    _WGRAD["main"] = None


class _OnWgradStream:
    def __init__(self, *operands):
        self.operands = operands

    def __enter__(self):
        side = _WGRAD["stream"]
        cur = torch.cuda.current_stream()
        if _WGRAD["main"] is None:
            _WGRAD["main"] = cur
            torch.autograd.Variable._execution_engine.queue_callback(_wgrad_join)
        side.wait_stream(cur)                                   # dZ (and everything before it) is ready
        _WGRAD["keep"].extend(self.operands)
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()

    def __exit__(self, *exc):
        return self.ctx.__exit__(*exc)


class _Inline:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


def _wgrad_scope(fused, *operands, rows=None):
    if not fused or _WGRAD["stream"] is None:
        return _Inline()
    if _WGRAD["max_rows"] is not None and rows is not None and rows > _WGRAD["max_rows"]:
        return _Inline()
    return _OnWgradStream(*operands)


import os as _os
_DGRAD_SPLIT_K = int(_os.environ.get("CTTS_DGRAD_SPLIT_K", "1"))
_FUSE_EPILOGUE_BWD = _os.environ.get("CTTS_FUSE_EPI_BWD", "1") != "0"      # EpiLink: producer's epilogue backward inside the consumer's dgrad GEMM     # tuning knob; 0 = never split the data-gradient reduction
_WGRAD_SPLIT_MULT = float(_os.environ.get("CTTS_WGRAD_SPLIT_MULT", "1"))
_WGRAD_SPLIT_CAP = int(_os.environ.get("CTTS_WGRAD_SPLIT_CAP", "512"))      # a split-K piece reduces over at least this many rows


def _split_k_for(Mo, No, Kred):
    """split-K factor for weight-gradient GEMMs (small output, long reduction)."""
    forced = int(_os.environ.get("CTTS_WGRAD_SPLIT", "0"))
    t128 = ((Mo + 127) // 128) * ((No + 127) // 128)
    if forced and t128 >= 64:
        return forced
    want = max(1, int(-(-512 // t128) * _WGRAD_SPLIT_MULT))
    return int(max(1, min(want, max(1, Kred // _WGRAD_SPLIT_CAP))))


class EpiLink:
    """Couples a layer whose forward epilogue is `drop(act(alpha (x (*) w + b)))` (the PRODUCER: FFN conv / first FF linear) with the ONE
    layer that consumes its output (the CONSUMER: ffn_2).  In the backward pass the consumer's data-gradient GEMM applies the producer's
    epilogue backward in its own epilogue (`ctts_gemm_desc.epi_bwd`): what it returns for its input is already the producer's dZ, and the
    producer skips its separate pass over the [M, N] gradient (epilogue_bwd: 2.9 % of the fs2 step).  Only valid when the producer's
    output has no other consumer - the model code that creates the link guarantees that."""

    def __init__(self):
        self.Z = self.seed = None
        self.act, self.p_drop, self.drop_offset = ACT_NONE, 0.0, 0
        self.conv_dims = None       # (cin, ksize) of a Conv1d producer: lets the consumer's backward decide whether the producer's dZ is wanted as planes
        self.armed = False          # producer forward has run with this link and no consumer has taken it yet
        self.gen = 0                # every producer forward is a new generation: a link object reused across forwards (or re-armed before
        self.done = set()           # backward) pairs each consumer backward with ITS producer; `done`: generations whose dZ was delivered

    def arm(self, Z, act, p_drop, seed, drop_offset, conv_dims=None):
        self.Z, self.act, self.p_drop, self.seed, self.drop_offset = Z, act, p_drop, seed, drop_offset
        self.conv_dims = conv_dims
        self.gen += 1
        self.armed = True
        return self.gen

    def take(self):
        """consumer forward: snapshot of the producer's epilogue (the consumer's backward must not read the mutable link: ADVICE r03) -
        one consumer per producer forward"""
        self.armed = False
        return (self.Z, self.act, self.p_drop, self.seed, self.drop_offset, self.gen, self.conv_dims)


class _LinearConv(torch.autograd.Function):
    """y = rowscale * (residual + drop(act(alpha * (x (*) w + b))))      (*) = matmul or Conv1d('same')

    Replaces nn.Linear / nn.Conv1d call sites of transformer_fs2.py:220-239,385-394,
    modules.py:140-148,1299-1356 and CompTransTTS.py:133 (fwd, dgrad, wgrad all on MFMA)."""

    @staticmethod
    def forward(ctx, x, w, b, residual, rowscale, act, alpha, p_drop, seed, drop_offset, ksize, row_lens, row_T, pr=None, link=None,
                link_role=0, pad_left=None):
        x = x.contiguous()
        Cin = x.shape[-1]
        M = x.numel() // Cin
        N = w.shape[0]
        out = torch.empty(*x.shape[:-1], N, dtype=torch.float32, device=x.device)
        Z = torch.empty_like(out) if act != ACT_NONE else None
        if ksize:
            T = x.shape[-2]
            wf = _gemm_major(w)
            if wf is None:
                wf = torch.empty(N, ksize * Cin, dtype=torch.float32, device=x.device)
                K.conv_weight_repack(w.contiguous(), wf, N, Cin, ksize, 0)
            # zeros in front of the sequence: (k-1)//2 = nn.Conv1d(padding=k//2) "SAME"; k-1 = ConstantPad1d((k-1, 0)) "LEFT" (causal)
            ctx.pad_left = (ksize - 1) // 2 if pad_left is None else int(pad_left)
            conv = (T, ctx.pad_left, Cin)
            Kdim = ksize * Cin
        else:
            wf, conv, Kdim = w.contiguous(), None, Cin
        if residual is not None:
            residual = residual.contiguous()
        gk = dict(conv=conv, alpha=alpha, bias=b, Z=Z, ldz=N, act=act, p_drop=p_drop, seed=seed, drop_offset=drop_offset, R=residual, ldr=N,
                  rowscale=rowscale, row_lens=row_lens, row_T=row_T, row_halo=0)
        # reference train.py:59,104 `with amp.autocast(args.use_amp)`: honoured by the launches the plane kernels take (the Conv1d layers
        # with many rows) - operands rounded to bf16, one MFMA term, fp32 accumulate and fp32 tensors everywhere; the backward of the
        # layer uses the arithmetic its forward ran with.  Everything else stays fp32-class.  Reduced precision: opt-in, own tolerance.
        ctx.amp = (K.amp_split() if (ksize and x.is_cuda and torch.is_autocast_enabled("cuda")) else None)
        if ctx.amp is not None:
            gk["bf16_split"] = ctx.amp
        planes = _operand_planes("fwd", w, x.view(M, Cin), wf, M, N, Kdim, Cin, Kdim, N, out, gk, a_given=K.planes_of(x)) if ksize else {}
        K.gemm(x, wf, out, M, N, Kdim, Cin, Kdim, N, True, True, tile_map=pr.tile_map(0, M) if (pr is not None and not planes) else None,
               **gk, **planes)
        ctx.x_planes = planes.get("a_planes")        # the weight-gradient launch reads the same plane set (gemm_plw.hip)
        ctx.save_for_backward(x, w, Z, rowscale, seed, row_lens, b)
        ctx.cfg = (act, alpha, p_drop, drop_offset, ksize, b is not None, residual is not None, row_T)
        ctx.pr = pr
        ctx.link, ctx.link_role, ctx.link_snap = None, 0, None
        if link is not None and _FUSE_EPILOGUE_BWD:
            if link_role == 1 and rowscale is None and residual is None and (act != ACT_NONE or p_drop > 0):
                ctx.link_gen = link.arm(Z, act, p_drop, seed, drop_offset, conv_dims=(Cin, ksize) if ksize else None)
                ctx.link, ctx.link_role = link, 1
            elif link_role == 2 and link.armed and not ksize:
                ctx.link, ctx.link_role, ctx.link_snap = link, 2, link.take()
        return out

    @staticmethod
    def backward(ctx, dY):
        x, w, Z, rowscale, seed, row_lens, b = ctx.saved_tensors
        act, alpha, p_drop, drop_offset, ksize, has_bias, has_res, row_T = ctx.cfg
        rl = dict(row_lens=row_lens, row_T=row_T) if row_lens is not None else {}
        if getattr(ctx, "amp", None) is not None:
            rl["bf16_split"] = ctx.amp           # travels with every GEMM of this backward (only the plane kernels act on it)
        pr = ctx.pr
        # weight gradients reduce over the (b,t) rows: the same device-built map, read as a schedule of the active 64-row K-blocks
        kmap = pr.tile_map(0, x.numel() // x.shape[-1]) if (pr is not None and row_lens is not None) else None
        dY = dY.contiguous()
        Cin = x.shape[-1]
        M = x.numel() // Cin
        N = w.shape[0]
        dX = dW = dB = None
        want_bias = has_bias and ctx.needs_input_grad[2]
        fuse_bias = want_bias and _fusable(b)
        if ctx.link_role == 1 and ctx.link_gen in ctx.link.done:
            # the consumer's data-gradient GEMM already applied this layer's epilogue backward: dY IS dZ (EpiLink)
            ctx.link.done.discard(ctx.link_gen)
            dZ, d_res = dY, None
            if want_bias:
                if fuse_bias:
                    K.colsum(dZ.view(M, N), scale=alpha, acc_into=_grad_of(b))
                else:
                    dB = K.colsum(dZ.view(M, N), scale=alpha)
        elif rowscale is None and act == ACT_NONE and p_drop == 0:
            dZ, d_res = dY, (dY if has_res else None)          # plain linear: nothing to undo
            if want_bias:
                if fuse_bias:
                    K.colsum(dZ.view(M, N), scale=alpha, acc_into=_grad_of(b))
                else:
                    dB = K.colsum(dZ.view(M, N), scale=alpha)
        else:
            # one pass: gm = dY * rowscale (gradient of the residual), dZ = gm * drop * act'(Z), bias gradient = alpha * colsum(dZ)
            dZ, gm, dBn = K.epilogue_bwd(dY, rowscale, Z, act, p_drop, seed, drop_offset, want_gm=has_res and rowscale is not None,
                                         want_bias=want_bias, bias_scale=alpha, bias_acc_into=_grad_of(b) if fuse_bias else None)
            d_res = (gm if rowscale is not None else dY) if has_res else None
            if want_bias and not fuse_bias:
                dB = dBn
        dz_planes = K.planes_of(dZ) if ksize else None          # written by dZ's producer (BatchNorm backward, the EpiLink GEMM epilogue)
        if ksize:
            T = x.shape[-2]
            pad = ctx.pad_left                   # forward / weight-gradient view
            pad_d = ksize - 1 - pad              # data gradient: correlation with the flipped taps (equal to `pad` for SAME with odd k)
            if ctx.needs_input_grad[0]:
                wd = _DGRAD_W.take(w, (Cin, ksize * N))          # prepared for the whole model at the start of the step (prepare_dgrad_weights)
                if wd is None:
                    wd = torch.empty(Cin, ksize * N, dtype=torch.float32, device=x.device)
                    wmaj = _gemm_major(w)
                    if wmaj is not None:
                        K.conv_weight_repack(wmaj, wd, N, Cin, ksize, 4)
                    else:
                        K.conv_weight_repack(w.contiguous(), wd, N, Cin, ksize, 1)
                # few output tiles but a long reduction (FFN conv dgrad: 1024 tiles, K = 9216): split K so that the launch fills
                # all 256 CUs x 8 resident workgroups (atomic accumulation into a zero-filled dX)
                tiles = -(-M // 64) * -(-Cin // 64)
                sk = min(4, max(2, -(-2304 // tiles))) if (_DGRAD_SPLIT_K and tiles < 1536 and ksize * N >= 4096) else 1
                dX = torch.empty_like(x)             # split: the ordered reduce launch writes every element (zeros in padded tiles)
                dg0 = dict(conv=(T, pad_d, N), alpha=alpha, row_halo=pad_d, **rl)
                planes = _operand_planes("dgrad", w, dZ.view(M, N), wd, M, Cin, ksize * N, N, ksize * N, Cin, dX, dict(split_overwrite=True, **dg0),
                                         a_given=dz_planes)
                if planes:          # pre-split operands on the persistent plane kernel: balances the reduction itself, writes every element
                    dz_planes = planes["a_planes"]
                    K.gemm(dZ, wd, dX, M, Cin, ksize * N, N, ksize * N, Cin, True, True, split_k=1, split_overwrite=True, **dg0, **planes)
                else:
                    dg = dict(tile_map=pr.tile_map(pad_d, M) if pr is not None else None, **dg0)
                    if sk > 1 and K.gemm_takes_persistent(dZ, wd, x, M, Cin, ksize * N, N, ksize * N, Cin, True, True, split_k=1, **dg):
                        sk = 1          # the persistent stream-K kernel balances the reduction itself: no split, no zero fill, no atomics
                    K.gemm(dZ, wd, dX, M, Cin, ksize * N, N, ksize * N, Cin, True, True, split_k=sk, split_overwrite=True, **dg)
            if ctx.needs_input_grad[1]:
                Kd = ksize * Cin
                fused = _fusable(w)
                gmaj = _gemm_major(_grad_of(w)) if fused else None
                wk = dict(conv=(T, pad, Cin), conv_on_b=True, split_k=max(2, _split_k_for(N, Kd, M)), alpha=alpha, **rl)
                if gmaj is not None:             # GEMM-major parameter: the split-K partials are added straight into param.grad
                    wp = _wgrad_planes(dZ.view(M, N), x.view(M, Cin), dz_planes, getattr(ctx, "x_planes", None), gmaj, N, Kd, M, Cin, wk)
                    with _wgrad_scope(True, dZ, x, *wp.values(), rows=M):        # the plane sets too: read on the side stream until the join
                        K.gemm(dZ, x, gmaj, N, Kd, M, N, Cin, Kd, False, False, tile_map=kmap, defer=_WGRAD["stream"] is None, **wk, **wp)
                else:
                    dwf = torch.empty(N, Kd, dtype=torch.float32, device=x.device)
                    wk["split_overwrite"] = True
                    wp = _wgrad_planes(dZ.view(M, N), x.view(M, Cin), dz_planes, getattr(ctx, "x_planes", None), dwf, N, Kd, M, Cin, wk)
                    with _wgrad_scope(fused, dZ, x, dwf, *wp.values(), rows=M):
                        K.gemm(dZ, x, dwf, N, Kd, M, N, Cin, Kd, False, False, tile_map=kmap, **wk, **wp)
                        if fused:
                            K.conv_weight_repack(dwf, _grad_of(w), N, Cin, ksize, 3)
                        elif _gemm_major(w) is not None:
                            dW = dwf.view(N, ksize, Cin).permute(0, 2, 1)      # a view with the parameter's own strides: no repack
                        else:
                            dW = torch.empty_like(w)
                            K.conv_weight_repack(dwf, dW, N, Cin, ksize, 2)
        else:
            if ctx.needs_input_grad[0]:
                dX = torch.empty_like(x)
                if ctx.link_role == 2:   # hand the producer its dZ: mask / (1-p) * act'(Z_producer) applied in this GEMM's epilogue
                    lZ, lact, lp, lseed, loff, lgen, lconv = ctx.link_snap     # the producer's epilogue as it was at THIS forward
                    gk = dict(alpha=alpha, epi_bwd=True, Z=lZ, ldz=Cin, act=lact, p_drop=lp, seed=lseed, drop_offset=loff,
                              tile_map=pr.tile_map(0, M) if pr is not None else None, **rl)
                    # dX IS the producer convolution's dZ: when that layer's data gradient runs on the plane kernel and this launch runs on
                    # the weight-stationary kernel (the only one whose epilogue writes planes), dZ arrives with its operand planes - no
                    # ctts_split_planes pass over the [M, 4C] gradient (round 6)
                    cpl = None
                    if (lconv is not None and K.PRODUCER_PLANES and dZ.is_cuda and Cin % 32 == 0
                            and K.plane_shape_ok(M, lconv[0], lconv[1] * Cin, Cin)):
                        cand = K.new_planes(M, Cin, dZ.device)
                        if K.gemm_takes_weight_stationary(dZ, w, dX, M, Cin, N, N, Cin, Cin, True, False, c_planes=cand, **gk):
                            cpl = cand
                    K.gemm(dZ, w, dX, M, Cin, N, N, Cin, Cin, True, False, c_planes=cpl, **gk)
                    if cpl is not None:
                        K.attach_planes(dX, cpl)
                    ctx.link.done.add(lgen)
                else:
                    K.gemm(dZ, w, dX, M, Cin, N, N, Cin, Cin, True, False, alpha=alpha,
                           tile_map=pr.tile_map(0, M) if pr is not None else None, **rl)
            if ctx.needs_input_grad[1] and N == 1 and row_lens is None:
                # one-output head: dW[0,:] = alpha * sum_r dZ[r] x[r,:] - a weighted column sum, not a 1 x C GEMM
                fused = _fusable(w)
                got = K.weighted_colsum(x.view(M, Cin), dZ.reshape(M), scale=alpha, acc_into=_grad_of(w).view(-1) if fused else None)
                dW = None if fused else got.view(1, Cin)
            elif ctx.needs_input_grad[1]:
                fused = _fusable(w)
                # fused: the ordered split-K sum is ADDED to param.grad (deferred to the stage's PartialSink); else it is WRITTEN into a fresh
                # tensor (split_overwrite: no zero fill)
                dW = _grad_of(w) if fused else torch.empty_like(w)
                with _wgrad_scope(fused, dZ, x, rows=M):
                    K.gemm(dZ, x, dW, N, Cin, M, N, Cin, Cin, False, False, split_k=max(2, _split_k_for(N, Cin, M)), alpha=alpha,
                           tile_map=kmap, defer=fused and _WGRAD["stream"] is None, split_overwrite=not fused, **rl)
                if fused:
                    dW = None
        return dX, dW, dB, d_res, None, None, None, None, None, None, None, None, None, None, None, None, None


# data-gradient operands of the Conv1d weights, keyed by the weight's device address; filled by prepare_dgrad_weights at the start of a
# train step (ONE launch for all layers instead of one repack inside every layer's backward), consumed by _LinearConv.backward.
# Entries carry the weight's autograd version: an in-place update through torch after the preparation invalidates them.


def _wstamp(w):
    """what a weight-derived cache entry is valid for: the tensor's autograd version AND the epoch of raw-pointer updates
    (dp.FlatAdam.step never bumps `_version`)"""
    return (w._version, K.WEIGHTS_EPOCH[0])


class _DgradCache(dict):
    """data_ptr -> (wd, version of the weight it was made from); pop() hands out only entries that still match the weight"""

    def take(self, w, shape):
        ent = dict.pop(self, w.data_ptr(), None)
        if ent is None or ent[1] != _wstamp(w) or tuple(ent[0].shape) != tuple(shape):
            return None
        return ent[0]


_DGRAD_W = _DgradCache()

# Pre-split bf16 planes (include/ctts.h ctts_split_planes) of the weight-side GEMM operands: the forward matrix [Cout, k*Cin] ("fwd") and
# the data-gradient matrix [Cin, k*Cout] ("dgrad") of the Conv1d layers whose launches the persistent plane kernel takes.  A layer
# announces itself the first time its launch qualifies (`want`); from then on prepare_dgrad_weights splits all announced weights in ONE
# launch at the start of the step.  Without a prepared entry (first step, plain forward outside trainer.TrainStep) the weight is split
# on the spot - a 5 - 8 us launch per layer.
_PLANES = {"fwd": {}, "dgrad": {}, "want_fwd": set(), "want_dgrad": set()}


def _operand_planes(kind, w, a_mat, b_mat, M, N, Kdim, lda, ldb, ldc, out, gk, a_given=None):
    """{} or dict(a_planes=, b_planes=) for K.gemm: the three-way bf16 split of the activation-side matrix `a_mat` [M, lda] (made here:
    one streaming launch) and of the weight-side matrix `b_mat` [N, Kdim] (from the step's cache) - when the library would run this
    launch on the plane kernel (asked with placeholder planes: no device work)."""
    conv = gk.get("conv")
    if not K.plane_shape_ok(M, N, Kdim, conv[2] if conv is not None else None):
        return {}
    if not (a_mat.is_contiguous() and b_mat.is_contiguous() and a_mat.shape[1] % 32 == 0 and b_mat.shape[1] % 32 == 0):
        return {}
    if not K.gemm_takes_planes(a_mat, b_mat, out, M, N, Kdim, lda, ldb, ldc, True, True, a_planes=_FakePlanes(a_mat), b_planes=_FakePlanes(b_mat), **gk):
        return {}
    ent = _PLANES[kind].get(w.data_ptr())
    if ent is not None and ent[1] == _wstamp(w) and ent[0].numel() == 3 * b_mat.numel():
        bp = ent[0]
    else:
        _PLANES["want_" + kind].add(w.data_ptr())
        bp = K.split_planes([b_mat])[0]
    # `a_given`: the set the activation's PRODUCER wrote next to the fp32 tensor (kernels.planes_of) - no split launch for it
    return dict(a_planes=a_given if a_given is not None else K.split_planes([a_mat])[0], b_planes=bp)


def _wgrad_planes(dz_mat, x_mat, dz_pl, x_pl, out, Mo, No, Kred, cin, kw):
    """{} or dict(a_planes=, b_planes=) for the weight-gradient launch `out [Mo, No] (+)= dz_mat^T (*) x_mat`: the ROW-MAJOR plane sets of
    dZ (made for the data-gradient launch) and of x (made for the forward launch) are what csrc/gemm_plw.hip reads - it transposes in
    the LDS; a set that is not at hand (first layer: no data gradient; forward on another kernel) is split here."""
    if not K.plane_wgrad_shape_ok(Mo, No, Kred, cin) or not (dz_mat.is_contiguous() and x_mat.is_contiguous()):
        return {}
    a = dz_pl if dz_pl is not None else _FakePlanes(dz_mat)
    b = x_pl if x_pl is not None else _FakePlanes(x_mat)
    if not K.gemm_takes_planes(dz_mat, x_mat, out, Mo, No, Kred, Mo, cin, No, False, False, a_planes=a, b_planes=b, **kw):
        return {}
    made = iter(K.split_planes([m for m, pl in ((dz_mat, dz_pl), (x_mat, x_pl)) if pl is None]))
    return dict(a_planes=dz_pl if dz_pl is not None else next(made), b_planes=x_pl if x_pl is not None else next(made))


class _FakePlanes:
    """stand-in with the addressing of a plane set (for the library's eligibility query only: nothing is read)"""

    def __init__(self, mat):
        self._m = mat

    def data_ptr(self):
        return self._m.data_ptr()


def prepare_dgrad_weights(params):
    """`params`: Conv1d weights in the GEMM-major layout of model._Conv (others are ignored).  Valid until the weights change - call it
    once per step, after the optimizer update and before the backward pass (trainer.TrainStep does, in front of the forward).  Builds
    the data-gradient matrices of all layers (one launch) and the bf16 planes of the weights the plane kernel consumes (one launch)."""
    clear_dgrad_weights()
    params = list(params)
    live = {w.data_ptr() for w in params}          # addresses announced by tensors that no longer exist (another model) are dropped
    _PLANES["want_fwd"] &= live
    _PLANES["want_dgrad"] &= live
    todo = []
    for w in params:
        wm = _gemm_major(w) if w.dim() == 3 else None
        if wm is not None and w.requires_grad:
            N, Cin, k = w.shape
            todo.append((w, (wm.detach(), N, Cin, k)))
    mats, slots = [], []
    for (w, (wm, _, _, _)), wd in zip(todo, K.conv_dgrad_weights([t for _, t in todo])):
        _DGRAD_W[w.data_ptr()] = (wd, _wstamp(w))
        if w.data_ptr() in _PLANES["want_fwd"]:
            mats.append(wm.contiguous()); slots.append(("fwd", w))
        if w.data_ptr() in _PLANES["want_dgrad"]:
            mats.append(wd); slots.append(("dgrad", w))
    for (kind, w), pl in zip(slots, K.split_planes(mats)):
        _PLANES[kind][w.data_ptr()] = (pl, _wstamp(w))


def clear_dgrad_weights():
    _DGRAD_W.clear()
    _PLANES["fwd"].clear()
    _PLANES["dgrad"].clear()


class PadRows:
    """Padding description of activations whose rows are (b, t) pairs: `lens` int32 [B] valid lengths, `T` padded length.
    Besides the per-tile skipping predicate the GEMM gets a device-built m-tile SCHEDULE (kernels.row_tile_map: active 64-row
    tiles first) per row halo - built once per forward/backward and shared by every layer of the stack (the lengths live on the
    device, so the host cannot order the tiles itself without a sync)."""

    def __init__(self, lens, T):
        self.lens, self.T = lens, int(T)
        self._maps = {}

    def tile_map(self, halo, M):
        key = (int(halo), int(M))
        if key not in self._maps:
            self._maps[key] = K.row_tile_map(self.lens, self.T, halo, M)
        return self._maps[key]


def _pad_rows(pad_rows):
    if pad_rows is None:
        return None, 0, None
    if isinstance(pad_rows, PadRows):
        return pad_rows.lens, pad_rows.T, pad_rows
    return pad_rows[0], int(pad_rows[1]), PadRows(pad_rows[0], pad_rows[1])


def linear(x, w, b=None, act=ACT_NONE, alpha=1.0, residual=None, rowscale=None, p_drop=0.0, drop=None, pad_rows=None, link=None,
           link_role=0):
    """pad_rows=(lens int32 [B], T) or an ops.PadRows: rows (b,t) with t >= lens[b] are padding - their outputs are don't-care
    (written as zero) and the incoming gradient there is zero, so whole padded tiles / K-blocks are skipped."""
    seed, off = (drop.seed, drop.next_offset()) if (drop is not None and p_drop > 0) else (None, 0)
    rl, rT, pr = _pad_rows(pad_rows)
    return _LinearConv.apply(x, w, b, residual, rowscale, act, alpha, p_drop if seed is not None else 0.0, seed, off, 0, rl, rT, pr,
                             link, link_role)


def conv1d(x, w, b=None, act=ACT_NONE, alpha=1.0, residual=None, rowscale=None, p_drop=0.0, drop=None, pad_rows=None, link=None,
           link_role=0, padding="SAME"):
    """x [B,T,Cin], w [Cout,Cin,k] (nn.Conv1d layout), stride 1, output length T.  padding "SAME": k//2 zeros on both sides
    (nn.Conv1d(padding=k//2), odd k); "LEFT": k-1 zeros in front, none behind (ConstantPad1d((k-1, 0)): the reference's causal
    ffn_padding, transformer_fs2.py:209-218, modules.py:1328-1331)."""
    seed, off = (drop.seed, drop.next_offset()) if (drop is not None and p_drop > 0) else (None, 0)
    rl, rT, pr = _pad_rows(pad_rows)
    if padding not in ("SAME", "LEFT"):
        raise ValueError(f"conv1d: padding '{padding}' (SAME or LEFT)")
    k = w.shape[2]
    return _LinearConv.apply(x, w, b, residual, rowscale, act, alpha, p_drop if seed is not None else 0.0, seed, off,
                             k, rl, rT, pr, link, link_role, (k - 1) if padding == "LEFT" else (k - 1) // 2)


def _adjacent(ts):
    """the tensors are dense, share one storage and follow each other without gaps (consecutive parameters of a flat arena)"""
    if any((not t.is_contiguous()) or t.dtype != torch.float32 for t in ts):
        return False
    try:
        st = ts[0].untyped_storage().data_ptr()
        if any(t.untyped_storage().data_ptr() != st for t in ts):
            return False
    except Exception:          # noqa: BLE001
        return False
    return all(b.data_ptr() == a.data_ptr() + a.numel() * 4 for a, b in zip(ts, ts[1:]))


class _PackedLinear(torch.autograd.Function):
    """y = x [w_0; w_1; ...]^T for weights that share the input (conformer q / k / v projections, conformer.py:264-295) as ONE GEMM.
    When the weights are consecutive tensors of one storage (trainer.TrainStep re-homes parameters into dp.FlatAdam's flat arena, in
    model.parameters() order) the stacked matrix is a VIEW: no concatenation in the forward, and in the backward the weight gradient
    is accumulated straight into the equally consecutive gradients of the flat gradient arena.  Otherwise: cat + split."""

    @staticmethod
    def forward(ctx, x, *ws):
        x = x.contiguous()
        Kd = x.shape[-1]
        M = x.numel() // Kd
        sizes = [w.shape[0] for w in ws]
        N = sum(sizes)
        wd = [w.detach() for w in ws]
        W = torch.as_strided(wd[0], (N, Kd), (Kd, 1)) if _adjacent(wd) else torch.cat(wd, 0)
        out = torch.empty(*x.shape[:-1], N, dtype=torch.float32, device=x.device)
        K.gemm(x, W, out, M, N, Kd, Kd, Kd, N, True, True)
        ctx.save_for_backward(x, *ws)
        ctx.sizes = sizes
        return out

    @staticmethod
    def backward(ctx, dY):
        x, ws = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        dY = dY.contiguous()
        Kd = x.shape[-1]
        M = x.numel() // Kd
        N = sum(ctx.sizes)
        wd = [w.detach() for w in ws]
        W = torch.as_strided(wd[0], (N, Kd), (Kd, 1)) if _adjacent(wd) else torch.cat(wd, 0)
        dX = None
        if ctx.needs_input_grad[0]:
            dX = torch.empty_like(x)
            K.gemm(dY, W, dX, M, Kd, N, N, Kd, Kd, True, False)
        gs = [_grad_of(w) for w in ws]
        sk = max(2, _split_k_for(N, Kd, M))
        if all(g is not None for g in gs) and _adjacent(gs):
            G = torch.as_strided(gs[0], (N, Kd), (Kd, 1))
            K.gemm(dY, x, G, N, Kd, M, N, Kd, Kd, False, False, split_k=sk, defer=_WGRAD["stream"] is None)
            return (dX,) + (None,) * len(ws)
        dW = torch.empty(N, Kd, dtype=torch.float32, device=x.device)
        K.gemm(dY, x, dW, N, Kd, M, N, Kd, Kd, False, False, split_k=sk, split_overwrite=True)
        return (dX,) + tuple(dW.split(ctx.sizes, 0))


def linear_packed(x, weights):
    """x @ cat(weights, 0)^T without the concatenation when the weights are neighbours in memory (see _PackedLinear)"""
    return _PackedLinear.apply(x, *weights)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, rowscale, p_drop, seed, drop_offset):
        x = x.contiguous()
        y, mean, rstd = K.layernorm_fwd(x, gamma, beta, eps, p_drop, seed, drop_offset, rowscale)
        ctx.save_for_backward(x, gamma, mean, rstd, rowscale, seed, beta)
        ctx.cfg = (p_drop, drop_offset)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd, rowscale, seed, beta = ctx.saved_tensors
        p_drop, drop_offset = ctx.cfg
        if _fusable(gamma) and _fusable(beta):
            dx, _, _ = K.layernorm_bwd(dy.contiguous(), x, gamma, mean, rstd, p_drop, seed, drop_offset, rowscale,
                                       acc_into=(gamma.grad, beta.grad))
            return dx, None, None, None, None, None, None, None
        dx, dg, db = K.layernorm_bwd(dy.contiguous(), x, gamma, mean, rstd, p_drop, seed, drop_offset, rowscale)
        return dx, dg, db, None, None, None, None, None


class _LayerNormRes(torch.autograd.Function):
    """(LN(x), x): the second output is x itself, handed to the caller as the residual operand of the sub-layer's last GEMM.  Both
    gradient paths then arrive HERE, and the LN backward kernel adds the residual one while it writes dx - no separate autograd add."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, want_planes=False):
        x = x.contiguous()
        y, mean, rstd = K.layernorm_fwd(x, gamma, beta, eps, 0.0, None, 0, None, want_planes=want_planes)
        ctx.save_for_backward(x, gamma, mean, rstd, beta)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres):
        x, gamma, mean, rstd, beta = ctx.saved_tensors
        if dy is None:
            return dres, None, None, None, None
        dres = dres.contiguous() if dres is not None else None
        if _fusable(gamma) and _fusable(beta):
            dx, _, _ = K.layernorm_bwd(dy.contiguous(), x, gamma, mean, rstd, acc_into=(gamma.grad, beta.grad), dres=dres)
            return dx, None, None, None, None
        dx, dg, db = K.layernorm_bwd(dy.contiguous(), x, gamma, mean, rstd, dres=dres)
        return dx, dg, db, None, None


def consumer_takes_planes(x, cout, ksize):
    """would the Conv1d (cout, k = ksize) that consumes the activation x [B, T, C] run on the plane kernel?  Then x's producer writes the
    bf16 plane set next to x (kernels.attach_planes) instead of leaving a ctts_split_planes pass over x to the consumer."""
    Cc = x.shape[-1]
    return bool(x.is_cuda and K.PRODUCER_PLANES and K.plane_shape_ok(x.numel() // Cc, int(cout), int(ksize) * Cc, Cc))


def layer_norm_res(x, gamma, beta, eps, planes_for=None):
    """-> (LayerNorm(x), x_res): use x_res as the `residual=` of the GEMM that closes the pre-LN sub-layer (transformer_fs2.py:186-199).
    planes_for = (cout, ksize) of the Conv1d that consumes the normalised tensor: when that launch runs on the plane kernel the LN launch
    writes the operand's bf16 plane set as well."""
    want = planes_for is not None and consumer_takes_planes(x, *planes_for)
    return _LayerNormRes.apply(x, gamma, beta, eps, want)


def layer_norm(x, gamma, beta, eps, rowscale=None, p_drop=0.0, drop=None):
    """y = rowscale * drop(LayerNorm(x)) over the last dim."""
    seed, off = (drop.seed, drop.next_offset()) if (drop is not None and p_drop > 0) else (None, 0)
    return _LayerNorm.apply(x, gamma, beta, eps, rowscale, p_drop if seed is not None else 0.0, seed, off)


# tuning knob; 1 = off (default since round 4: with the ordered reduce launch a 2-way split of the [T, T] x [T, dh] products costs more
# than the under-filled launch it avoids - fs2 step 23.73 -> 23.50 ms, same box)
_ATTN_SPLIT_K = int(_os.environ.get("CTTS_ATTN_SPLIT_K", "1"))


def _attn_split_k(nbatch, T, dh):
    """The [T, d_h] outputs of attention (P V, dV, dQ, dK) are only ceil(T/64) x ceil(d_h/64) tiles per (batch, head) - 1,024 workgroups
    for 2,048 slots (256 CUs x 8 resident workgroups) at the canonical batch - but reduce over T keys / queries: split that reduction (atomic accumulation into the
    zero-initialised outputs) until the launch fills the machine."""
    tiles = -(-T // 64) * -(-dh // 64) * nbatch
    return _ATTN_SPLIT_K if (_ATTN_SPLIT_K > 1 and tiles < 1536 and T >= 512) else 1


# Which attention pipeline runs (CTTS_FUSED_ATTN = auto | 1 | 0, or set_fused_attention):
#   auto  relative-position attention (conformer, d_head 32): fused - it removes 5 of the 7 [B,H,T,T] maps (46.4 -> 36.4 ms per train step);
#         fs2 attention (2 heads x 128): fused for inference / no-grad forwards (165 vs 187 us per decoder layer, no 134 MB score tensor),
#         UNFUSED when gradients are needed: at d_head 128 the fp32-MFMA time dominates and the recomputing backward costs 450 us per
#         decoder layer against 305 us for the GEMM pipeline that keeps P (B*H = 32 gives one wave per SIMD and a 32-tile critical path)
#   1 / 0 force the fused / unfused kernels everywhere (tests, A/B measurements)
_FUSED_ATTN = {"1": True, "0": False}.get(_os.environ.get("CTTS_FUSED_ATTN", "auto"), None)
_ATTN_Q_SPLIT = int(_os.environ.get("CTTS_ATTN_Q_SPLIT", "0"))     # 0 = auto; n: split a key tile's query loop over n waves


def set_fused_attention(flag):
    """True / False force a pipeline, None = auto (see above)"""
    global _FUSED_ATTN
    _FUSED_ATTN = flag if flag is None else bool(flag)


class _FusedSelfAttention(torch.autograd.Function):
    """The same operator as _SelfAttention on csrc/attn.hip: no [B,H,T,T] tensor in the forward pass (the unfused pipeline writes
    S, rewrites it in the softmax and re-reads it in P V: 134 MB per decoder layer each time), backward by recomputation with dK / dV
    accumulated in registers; only dS is materialised once for the dQ GEMM."""

    @staticmethod
    def forward(ctx, qkv, lens, n_heads):
        qkv = qkv.contiguous()
        dh = qkv.shape[-1] // 3 // n_heads
        out, lse = K.mha_fwd(qkv, lens, n_heads, dh ** -0.5)
        ctx.save_for_backward(qkv, lens, out, lse)
        ctx.n_heads = n_heads
        return out

    @staticmethod
    def backward(ctx, dO):
        qkv, lens, out, lse = ctx.saved_tensors
        H = ctx.n_heads
        B, T, C3 = qkv.shape
        dh = C3 // 3 // H
        # one wave per 32 keys walks ALL query tiles: with B*H*T/32 <= 2 waves per SIMD (fs2: 1,024 waves for 1,024 SIMDs) the launch
        # lasts as long as its longest wave, so the query loop is split over 2 waves (partial dK / dV, summed in a fixed order)
        qs = _ATTN_Q_SPLIT if _ATTN_Q_SPLIT > 0 else (2 if (B * H * ((T + 31) // 32) <= 2048 and T >= 256) else 1)
        return K.mha_bwd(qkv, lens, out, dO.contiguous(), lse, H, dh ** -0.5, qs), None, None


class _SelfAttention(torch.autograd.Function):
    """Multi-head self-attention core on the packed projection qkv [B,T,3C] with a key-padding
    mask given as valid lengths (F.multi_head_attention_forward semantics,
    transformer_fs2.py:385-394: q scaled by d_h^-0.5, padded keys get -inf, no biases, no
    attention dropout).  Query rows >= len are skipped (their output is zero; the caller
    multiplies by the non-pad mask anyway, transformer_fs2.py:190)."""

    @staticmethod
    def forward(ctx, qkv, lens, n_heads):
        qkv = qkv.contiguous()
        B, T, C3 = qkv.shape
        C = C3 // 3
        dh = C // n_heads
        scale = dh ** -0.5
        S = torch.empty(B, n_heads, T, T, dtype=torch.float32, device=qkv.device)
        K.gemm(qkv, qkv, S, T, T, dh, C3, C3, T, True, True, a_off=0, b_off=C, nb0=B, nb1=n_heads,
               sA=(T * C3, dh), sB=(T * C3, dh), sC=(n_heads * T * T, T * T), lens=lens, lim=(1, 1, 0), alpha=scale)
        K.softmax_fwd(S, lens, B, n_heads, T)
        sk = _attn_split_k(B * n_heads, T, dh)
        # "write everything" (split_overwrite): the query rows >= len are written as zeros by the launch itself - no fill launch
        out = torch.empty(B, T, C, dtype=torch.float32, device=qkv.device)
        K.gemm(S, qkv, out, T, dh, T, T, C3, C, True, False, b_off=2 * C, nb0=B, nb1=n_heads,
               sA=(n_heads * T * T, T * T), sB=(T * C3, dh), sC=(T * C, dh), lens=lens, lim=(1, 0, 1), split_k=sk, split_overwrite=True)
        ctx.save_for_backward(qkv, S, lens, out)
        ctx.n_heads = n_heads
        return out

    @staticmethod
    def backward(ctx, dO):
        qkv, P, lens, O = ctx.saved_tensors
        H = ctx.n_heads
        dO = dO.contiguous()
        B, T, C3 = qkv.shape
        C = C3 // 3
        dh = C // H
        scale = dh ** -0.5
        sP = (H * T * T, T * T)
        sk = _attn_split_k(B * H, T, dh)
        dqkv = torch.empty_like(qkv)            # the three launches below write all of it ("write everything": zeros beyond each utterance's length)
        # dV[key,d] = sum_q P[q,key] dO[q,d]
        K.gemm(P, dO, dqkv, T, dh, T, T, C, C3, False, False, c_off=2 * C, nb0=B, nb1=H, sA=sP, sB=(T * C, dh),
               sC=(T * C3, dh), lens=lens, lim=(1, 0, 1), split_k=sk, split_overwrite=True)
        # dS[q,key] = P * (dO V^T - D),  D[q] = sum_key dP P = sum_d dO[q,d] O[q,d]: the softmax backward rides in the epilogue of the
        # dP GEMM (one read of P) instead of a separate pass that re-reads P and dP and rewrites dS
        Dv = K.rowdot_heads(dO, O, H)
        dP = torch.empty_like(P)
        K.gemm(dO, qkv, dP, T, T, dh, C, C3, T, True, True, b_off=2 * C, nb0=B, nb1=H, sA=(T * C, dh), sB=(T * C3, dh),
               sC=sP, lens=lens, lim=(1, 1, 0), E=P, rowsub=Dv)
        # dQ[q,d] = scale * sum_key dS[q,key] K[key,d]
        K.gemm(dP, qkv, dqkv, T, dh, T, T, C3, C3, True, False, b_off=C, c_off=0, nb0=B, nb1=H, sA=sP, sB=(T * C3, dh),
               sC=(T * C3, dh), lens=lens, lim=(1, 0, 1), alpha=scale, split_k=sk, split_overwrite=True)
        # dK[key,d] = scale * sum_q dS[q,key] Q[q,d]
        K.gemm(dP, qkv, dqkv, T, dh, T, T, C3, C3, False, False, b_off=0, c_off=C, nb0=B, nb1=H, sA=sP, sB=(T * C3, dh),
               sC=(T * C3, dh), lens=lens, lim=(1, 0, 1), alpha=scale, split_k=sk, split_overwrite=True)
        return dqkv, None, None


def self_attention(qkv, lens_i32, n_heads):
    want = _FUSED_ATTN if _FUSED_ATTN is not None else not (torch.is_grad_enabled() and qkv.requires_grad)
    if want and K.mha_supported(qkv.shape[-1] // 3, n_heads):
        return _FusedSelfAttention.apply(qkv, lens_i32, n_heads)
    return _SelfAttention.apply(qkv, lens_i32, n_heads)


class _LRGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mel2ph, cum):
        x = x.contiguous()
        ctx.save_for_backward(cum)
        ctx.Ts = x.shape[1]
        return K.lr_gather_fwd(x, mel2ph)

    @staticmethod
    def backward(ctx, dy):
        (cum,) = ctx.saved_tensors
        return K.lr_gather_bwd(dy.contiguous(), cum, ctx.Ts), None, None


def length_regulate(x, dur, max_len=None):
    """LengthRegulator (modules.py:1216-1249): expand row i of x `int(dur[i])` times, pad/crop to
    max_len.  Returns (out [B,Tm,C], mel_len int64 [B] (un-cropped), mel2ph int32 [B,Tm]).
    One prefix-scan kernel replaces the reference's B*Ts `.item()` host syncs."""
    if max_len is None:
        _, mel_len, _ = K.lr_index(dur, 0, want_mel2ph=False)
        max_len = max(int(mel_len.max().item()), 1)   # inference only: output width is data dependent
    mel2ph, mel_len, cum = K.lr_index(dur, int(max_len))
    return _LRGather.apply(x, mel2ph, cum), mel_len, mel2ph


def dur_to_mel2ph(dur, dur_padding=None):
    """utils/tools.py:598-628 -> int64 [B, max total]."""
    _, total, _ = K.lr_index(dur, 0, pad=dur_padding, round_mode=1, want_mel2ph=False)
    Tm = max(int(total.max().item()), 0)
    mel2ph, _, _ = K.lr_index(dur, Tm, pad=dur_padding, round_mode=1)
    return mel2ph.long()


class _BatchNormAct(torch.autograd.Function):
    """x [..., C] -> drop(act(BN(x))) with the flattening to [rows, C] INSIDE the Function, so that the tensors that cross its boundary
    are the ones the neighbouring convolutions see: a plane set attached to the output (forward) / the input gradient (backward) by the
    kernels reaches the consuming GEMM with the tensor (kernels.attach_planes)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mean, rstd, act, p_drop, seed, drop_offset, batch_stats, want_planes, want_dx_planes):
        x2d = x.contiguous().view(-1, x.shape[-1])
        if want_planes:
            y2d, pl = K.bn_apply(x2d, mean, rstd, gamma, beta, act, p_drop, seed, drop_offset, want_planes=True)
        else:
            y2d, pl = K.bn_apply(x2d, mean, rstd, gamma, beta, act, p_drop, seed, drop_offset), None
        ctx.save_for_backward(x2d, gamma, beta, mean, rstd, seed)
        ctx.cfg = (act, p_drop, drop_offset, batch_stats, bool(want_dx_planes), x.shape)
        y = y2d.view(x.shape)
        if pl is not None:
            K.attach_planes(y, pl)
        return y

    @staticmethod
    def backward(ctx, dy):
        x2d, gamma, beta, mean, rstd, seed = ctx.saved_tensors
        act, p_drop, drop_offset, batch_stats, want_dx_planes, shape = ctx.cfg
        gg, gb = _grad_of(gamma), _grad_of(beta)
        acc = (gg, gb) if (gg is not None and gb is not None and gg.is_contiguous() and gb.is_contiguous()) else None
        dx2d, dg, db = K.bn_bwd(dy.contiguous().view(-1, x2d.shape[-1]), x2d, mean, rstd, gamma, beta, act, p_drop, seed, drop_offset, batch_stats,
                                acc_into=acc, want_planes=want_dx_planes)
        dx = dx2d.view(shape)
        pl = K.planes_of(dx2d)
        if pl is not None:
            K.attach_planes(dx, pl)
        return dx, dg, db, None, None, None, None, None, None, None, None, None


def batch_norm_act(x, gamma, beta, running_mean, running_var, num_batches_tracked, training, act=ACT_NONE, p_drop=0.0,
                   drop=None, eps=1e-5, momentum=0.1, planes=False, dx_planes=False):
    """drop(act(BatchNorm1d(x))) on channel-last x [B,T,C]; statistics over all B*T rows,
    pads included, exactly as nn.BatchNorm1d sees them in modules.py:140-148.
    planes / dx_planes: the launch that writes the output / the input gradient also writes its bf16 plane set (the neighbouring
    convolution consumes it on the plane kernel: ops.consumer_takes_planes)."""
    C = x.shape[-1]
    x2d = x.contiguous().view(-1, C)
    if training:
        with torch.no_grad():
            mean, rstd = K.bn_batch_stats(x2d, eps, momentum, running_mean, running_var, num_batches_tracked)
    else:
        mean = running_mean
        rstd = torch.rsqrt(running_var + eps)
    seed, off = (drop.seed, drop.next_offset()) if (drop is not None and p_drop > 0) else (None, 0)
    return _BatchNormAct.apply(x, gamma, beta, mean, rstd, act, p_drop if seed is not None else 0.0, seed, off, bool(training),
                               bool(planes), bool(dx_planes))


def sinusoid_table(n_pos, dim, device):
    """fs2 SinusoidalPositionalEmbedding.get_embedding (blocks.py:66-83): [sin|cos], row 0 = 0."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    freq = torch.exp(torch.arange(half, dtype=torch.float) * -e)
    ang = torch.arange(n_pos, dtype=torch.float)[:, None] * freq[None, :]
    tab = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)
    tab[0] = 0
    return tab.to(device)


def mel_spectrogram(y, dft_basis, mel_basis_padded, n_fft, hop, n_mel, nbins, clip=1e-5):
    """TacotronSTFT.mel_spectrogram (audio/stft.py:166-185) -> (mel [B,n_mel,F], energy [B,F]).
    Frames are an overlapping-row view of the reflect-padded waveform (row stride = hop), so the
    windowed DFT is one MFMA GEMM [B*F,1024] x [1026,1024]^T with no framing copy."""
    B, N = y.shape
    pad = n_fft // 2
    F = 1 + N // hop
    ypad = K.reflect_pad(y.contiguous(), pad)
    W = ypad.shape[1]
    nre = dft_basis.shape[0]
    reim = torch.empty(B * F, nre, dtype=torch.float32, device=y.device)
    K.gemm(ypad, dft_basis, reim, F, nre, n_fft, hop, n_fft, nre, True, True, nb0=B, nb1=1, sA=(W, 0), sB=(0, 0),
           sC=(F * nre, 0))
    ld_mag = mel_basis_padded.shape[1]
    mag, energy = K.stft_magnitude(reim, B * F, nbins, ld_mag)
    mel_fm = torch.empty(B * F, n_mel, dtype=torch.float32, device=y.device)
    K.gemm(mag, mel_basis_padded, mel_fm, B * F, n_mel, ld_mag, ld_mag, ld_mag, n_mel, True, True)
    mel = K.log_clamp_transpose(mel_fm, B, F, n_mel, clip)
    return mel, energy.view(B, F), mag


class _RowscaleDropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rowscale, p_drop, seed, drop_offset):
        ctx.save_for_backward(rowscale, seed)
        ctx.cfg = (p_drop, drop_offset)
        return K.rowscale_dropout(x.contiguous(), rowscale, p_drop, seed, drop_offset)

    @staticmethod
    def backward(ctx, dy):
        rowscale, seed = ctx.saved_tensors
        p_drop, drop_offset = ctx.cfg
        return K.rowscale_dropout(dy.contiguous(), rowscale, p_drop, seed, drop_offset), None, None, None, None


class _PosEmbedAdd(torch.autograd.Function):
    """y = rowscale * dropout(x + alpha * table[pos])  (csrc/elementwise.hip posembed_*): one launch forward, one backward"""

    @staticmethod
    def forward(ctx, x, alpha, pos, table, rowscale, p_drop, seed, drop_offset):
        ctx.save_for_backward(pos, table, rowscale, seed, alpha)
        ctx.cfg = (p_drop, drop_offset)
        return K.posembed_fwd(x.contiguous(), pos, table, alpha.detach() if alpha is not None else None, rowscale, p_drop, seed, drop_offset)

    @staticmethod
    def backward(ctx, dy):
        pos, table, rowscale, seed, alpha = ctx.saved_tensors
        p_drop, drop_offset = ctx.cfg
        want_alpha = alpha is not None and ctx.needs_input_grad[1]
        acc = _grad_of(alpha) if want_alpha else None
        dx, dalpha = K.posembed_bwd(dy.contiguous(), pos, table, rowscale, p_drop, seed, drop_offset, want_alpha=want_alpha, alpha_acc_into=acc)
        return dx, (None if (acc is not None or not want_alpha) else dalpha.view_as(alpha)), None, None, None, None, None, None


def posembed_add(x, pos, table, alpha=None, rowscale=None, p_drop=0.0, drop=None):
    """rowscale * dropout(x + alpha * table[pos]): the fs2 `x + pos_embed_alpha * embed_positions(x)` (+ F.dropout + non-pad mask) fused;
    pos int32 [B, T] (kernels.positions), table [n_pos, C] (ops.sinusoid_table), alpha: the [1] parameter or None (= 1)."""
    seed, off = (drop.seed, drop.next_offset()) if (drop is not None and p_drop > 0) else (None, 0)
    return _PosEmbedAdd.apply(x, alpha, pos, table, rowscale, p_drop if seed is not None else 0.0, seed, off)


def rowscale_dropout(x, rowscale=None, p_drop=0.0, drop=None):
    """y = rowscale[row] * dropout(x)  (F.dropout followed by the non-pad mask multiply)."""
    seed, off = (drop.seed, drop.next_offset()) if (drop is not None and p_drop > 0) else (None, 0)
    if rowscale is None and seed is None:
        return x
    return _RowscaleDropout.apply(x, rowscale, p_drop if seed is not None else 0.0, seed, off)


class _GradScale(torch.autograd.Function):
    """x.detach() + g * (x - x.detach())  (modules.py:1025-1027): identity forward, gradient * g."""

    @staticmethod
    def forward(ctx, x, g):
        ctx.g = g
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        return dy * ctx.g, None


def grad_scale(x, g):
    return _GradScale.apply(x, g)


# ------------------------------------------------------------------------- conformer operators
class _GLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):
        a = a.contiguous()
        ctx.save_for_backward(a)
        return K.glu_fwd(a)

    @staticmethod
    def backward(ctx, dout):
        (a,) = ctx.saved_tensors
        return K.glu_bwd(a, dout.contiguous())


def glu(a):
    """GLU over the last dim: a[..., :C] * sigmoid(a[..., C:])  (blocks.GLU with channel-last layout)."""
    return _GLU.apply(a)


class _DepthwiseConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        x = x.contiguous()
        C, _, Kk = w.shape
        wT = w.view(C, Kk).t().contiguous()          # [K, C] tap-major copy (31 x 256 floats)
        ctx.save_for_backward(x, wT, w)
        ctx.wshape = w.shape
        return K.dwconv_fwd(x, wT, False)

    @staticmethod
    def backward(ctx, dy):
        x, wT, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = K.dwconv_fwd(dy, wT, True) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            g = _grad_of(w)
            if g is not None and g.is_contiguous():
                K.dwconv_wgrad(dy, x, wT.shape[0], acc_into=g)          # straight into param.grad ([C,1,K] is [C,K] in memory)
            else:
                dw = K.dwconv_wgrad(dy, x, wT.shape[0]).view(ctx.wshape)
        return dx, dw


def depthwise_conv1d(x, w):
    """x [B,T,C], w [C,1,k] (nn.Conv1d groups=C layout), 'same' padding, no bias (conformer.py:522-560)."""
    return _DepthwiseConv.apply(x, w)


class _RelPosAttention(torch.autograd.Function):
    """Core of RelativeMultiHeadAttention (conformer.py:396-421) on channel-last projections.

    qu = q + u_bias, qv = q + v_bias [B,T,C]; kv [B,T,2C] (k | v); pos [T,C] (projected sinusoid table, shared by
    the batch).  score = (qu k^T + shift(qv pos^T)) / sqrt(d_model), softmax over ALL keys (the reference never passes
    the mask, conformer.py:243), dropout on the probabilities, context = P v."""

    @staticmethod
    def forward(ctx, qu, qv, kv, pos, n_heads, scale, p_drop, seed, drop_offset):
        qu, qv, kv, pos = qu.contiguous(), qv.contiguous(), kv.contiguous(), pos.contiguous()
        B, T, C = qu.shape
        dh = C // n_heads
        H = n_heads
        sS = (H * T * T, T * T)
        S = torch.empty(B, H, T, T, dtype=torch.float32, device=qu.device)
        PS = torch.empty(B, H, T, T, dtype=torch.float32, device=qu.device)
        K.gemm(qu, kv, S, T, T, dh, C, 2 * C, T, True, True, nb0=B, nb1=H, sA=(T * C, dh), sB=(T * 2 * C, dh), sC=sS)
        K.gemm(qv, pos, PS, T, T, dh, C, C, T, True, True, nb0=B, nb1=H, sA=(T * C, dh), sB=(0, dh), sC=sS)
        Pd = K.relpos_softmax_fwd(S, PS, T, scale, p_drop, seed, drop_offset, want_dropped=True)   # S <- P
        del PS
        out = torch.empty(B, T, C, dtype=torch.float32, device=qu.device)
        K.gemm(Pd, kv, out, T, dh, T, T, 2 * C, C, True, False, b_off=C, nb0=B, nb1=H, sA=sS, sB=(T * 2 * C, dh), sC=(T * C, dh))
        ctx.save_for_backward(qu, qv, kv, pos, S, seed)
        ctx.cfg = (n_heads, scale, p_drop, drop_offset)
        return out

    @staticmethod
    def backward(ctx, dO):
        qu, qv, kv, pos, P, seed = ctx.saved_tensors
        H, scale, p_drop, drop_offset = ctx.cfg
        dO = dO.contiguous()
        B, T, C = qu.shape
        dh = C // H
        sS = (H * T * T, T * T)
        dkv = torch.empty_like(kv)
        # dV = Pd^T dO   (Pd regenerated from P and the counter-based mask)
        Pd = K.rowscale_dropout(P, None, p_drop, seed, drop_offset) if p_drop > 0 else P
        K.gemm(Pd, dO, dkv, T, dh, T, T, C, 2 * C, False, False, c_off=C, nb0=B, nb1=H, sA=sS, sB=(T * C, dh), sC=(T * 2 * C, dh))
        del Pd
        # dPd = dO V^T ; dS = P * (dP - sum dP P) * scale
        dS = torch.empty_like(P)
        K.gemm(dO, kv, dS, T, T, dh, C, 2 * C, T, True, True, b_off=C, nb0=B, nb1=H, sA=(T * C, dh), sB=(T * 2 * C, dh), sC=sS)
        K.relpos_softmax_bwd(P, dS, T, scale, p_drop, seed, drop_offset)
        # content path: dQU = dS K ; dK = dS^T QU
        dqu = torch.empty_like(qu)
        K.gemm(dS, kv, dqu, T, dh, T, T, 2 * C, C, True, False, nb0=B, nb1=H, sA=sS, sB=(T * 2 * C, dh), sC=(T * C, dh))
        K.gemm(dS, qu, dkv, T, dh, T, T, C, 2 * C, False, False, c_off=0, nb0=B, nb1=H, sA=sS, sB=(T * C, dh), sC=(T * 2 * C, dh))
        # position path: dPS = unshift(dS) ; dQV = dPS pos ; dpos = sum_b dPS^T QV
        dPS = K.relshift_bwd(dS, T)
        del dS
        dqv = torch.empty_like(qv)
        K.gemm(dPS, pos, dqv, T, dh, T, T, C, C, True, False, nb0=B, nb1=H, sA=sS, sB=(0, dh), sC=(T * C, dh))
        dpos_b = torch.empty(B, T, C, dtype=torch.float32, device=qu.device)
        K.gemm(dPS, qv, dpos_b, T, dh, T, T, C, C, False, False, nb0=B, nb1=H, sA=sS, sB=(T * C, dh), sC=(T * C, dh))
        dpos = K.colsum(dpos_b.view(dpos_b.shape[0], -1)).view(dpos_b.shape[1:])          # ordered sum over the batch (no torch reduction)
        return dqu, dqv, dkv, dpos, None, None, None, None, None


class _FusedRelPosAttention(torch.autograd.Function):
    """_RelPosAttention on csrc/attn.hip: the position scores are computed inside the kernels (the reference's pad-and-reshape shift,
    conformer.py:423-431, in closed form); the only [B,H,T,T]-sized tensor left is dS in the backward, written once in the layout
    the three remaining GEMMs read it in.  The unfused pipeline moves 7 such maps, each 512 MB per decoder layer at B=16, T=1000.
    Same counter-RNG element indices as the unfused path, so both draw identical dropout masks."""

    @staticmethod
    def forward(ctx, qu, qv, kv, pos, n_heads, scale, p_drop, seed, drop_offset):
        qu, qv, kv, pos = qu.contiguous(), qv.contiguous(), kv.contiguous(), pos.contiguous()
        out, lse = K.relmha_fwd(qu, qv, kv, pos, n_heads, scale, p_drop, seed, drop_offset)
        ctx.save_for_backward(qu, qv, kv, pos, out, lse, seed)
        ctx.cfg = (n_heads, scale, p_drop, drop_offset)
        return out

    @staticmethod
    def backward(ctx, dO):
        qu, qv, kv, pos, out, lse, seed = ctx.saved_tensors
        H, scale, p_drop, drop_offset = ctx.cfg
        dqu, dqv, dkv, dpos_b = K.relmha_bwd(qu, qv, kv, pos, out, dO.contiguous(), lse, H, scale, p_drop, seed, drop_offset)
        return dqu, dqv, dkv, K.colsum(dpos_b.view(dpos_b.shape[0], -1)).view(dpos_b.shape[1:]), None, None, None, None, None


class _RelAttnSplit(torch.autograd.Function):
    """(qkv [B,T,3C], u_bias [C], v_bias [C]) -> (q + u_bias, q + v_bias, k | v): the operand preparation of
    RelativeMultiHeadAttention.forward (conformer.py:396-407) in one pass; backward one pass + two column sums."""

    @staticmethod
    def forward(ctx, qkv, u_bias, v_bias):
        ctx.fuse = (_fusable(u_bias), _fusable(v_bias))
        ctx.biases = (u_bias, v_bias)
        return K.relattn_split_fwd(qkv.contiguous(), u_bias.contiguous(), v_bias.contiguous())

    @staticmethod
    def backward(ctx, dqu, dqv, dkv):
        dqu, dqv, dkv = dqu.contiguous(), dqv.contiguous(), dkv.contiguous()
        Cc = dqu.shape[-1]
        u_bias, v_bias = ctx.biases
        du = dv = None
        if ctx.needs_input_grad[1]:
            if ctx.fuse[0]:
                K.colsum(dqu.view(-1, Cc), acc_into=u_bias.grad.view(-1))
            else:
                du = K.colsum(dqu.view(-1, Cc)).view_as(u_bias)
        if ctx.needs_input_grad[2]:
            if ctx.fuse[1]:
                K.colsum(dqv.view(-1, Cc), acc_into=v_bias.grad.view(-1))
            else:
                dv = K.colsum(dqv.view(-1, Cc)).view_as(v_bias)
        return K.relattn_split_bwd(dqu, dqv, dkv), du, dv


def relattn_split(qkv, u_bias, v_bias):
    return _RelAttnSplit.apply(qkv, u_bias, v_bias)


def relpos_attention(qu, qv, kv, pos, n_heads, scale, p_drop=0.0, drop=None):
    seed, off = (drop.seed, drop.next_offset()) if (drop is not None and p_drop > 0) else (None, 0)
    fn = _FusedRelPosAttention if (_FUSED_ATTN is not False and K.mha_supported(qu.shape[-1], n_heads)) else _RelPosAttention
    return fn.apply(qu, qv, kv, pos, n_heads, scale, p_drop if seed is not None else 0.0, seed, off)


# ------------------------------------------------------------------------- unsupervised alignment operators
class _BmmNN(torch.autograd.Function):
    """out[b] = A[b] @ X[b]  (A [B,M,K], X [B,K,N]) on ctts_gemm; used for the soft-attention upsampling
    `torch.bmm(A_soft, x)` of modules.py:1048-1049."""

    @staticmethod
    def forward(ctx, A, X):
        A, X = A.contiguous(), X.contiguous()
        B, M, Kd = A.shape
        N = X.shape[2]
        out = torch.empty(B, M, N, dtype=torch.float32, device=A.device)
        K.gemm(A, X, out, M, N, Kd, Kd, N, N, True, False, nb0=B, nb1=1, sA=(M * Kd, 0), sB=(Kd * N, 0), sC=(M * N, 0))
        ctx.save_for_backward(A, X)
        return out

    @staticmethod
    def backward(ctx, dO):
        A, X = ctx.saved_tensors
        dO = dO.contiguous()
        B, M, Kd = A.shape
        N = X.shape[2]
        dA = dX = None
        if ctx.needs_input_grad[0]:
            dA = torch.empty_like(A)          # dA[m,k] = sum_n dO[m,n] X[k,n]
            K.gemm(dO, X, dA, M, Kd, N, N, N, Kd, True, True, nb0=B, nb1=1, sA=(M * N, 0), sB=(Kd * N, 0), sC=(M * Kd, 0))
        if ctx.needs_input_grad[1]:
            dX = torch.empty_like(X)          # dX[k,n] = sum_m A[m,k] dO[m,n]
            K.gemm(A, dO, dX, Kd, N, M, Kd, N, N, False, False, nb0=B, nb1=1, sA=(M * Kd, 0), sB=(M * N, 0), sC=(Kd * N, 0))
        return dA, dX


def bmm_nn(A, X):
    return _BmmNN.apply(A, X)


class _NegSqDist(torch.autograd.Function):
    """attn[b,t,s] = -temp * sum_c (q[b,t,c] - k[b,s,c])^2   (AlignmentEncoder, modules.py:1199-1200)"""

    @staticmethod
    def forward(ctx, q, k, temp):
        q, k = q.contiguous(), k.contiguous()
        ctx.save_for_backward(q, k)
        ctx.temp = temp
        return K.neg_sqdist(q, k, temp)

    @staticmethod
    def backward(ctx, g):
        q, k = ctx.saved_tensors
        g = g.contiguous()
        t2 = -2.0 * ctx.temp
        # d/dq = -2 temp (q * rowsum(g) - g k);   d/dk = -2 temp (k * colsum(g) - g^T q).  The two sums of g ride in the GEMMs that read g
        # anyway: a ones column appended to k / q (padded to a multiple of four columns) makes the last used output column rowsum(g) /
        # colsum(g) - no stock-torch reduction over the [B, Tq, Tk] gradient on the captured path
        B, Tq, Tk = g.shape
        Cc = q.shape[2]
        C1 = (Cc + 1 + 3) // 4 * 4

        def with_ones(x):
            out = x.new_zeros(x.shape[0], x.shape[1], C1)
            out[:, :, :Cc] = x
            out[:, :, Cc] = 1.0
            return out
        gk = bmm_nn(g, with_ones(k))                                  # [B, Tq, C1]: g k | rowsum(g)
        dq = t2 * (q * gk[:, :, Cc:Cc + 1] - gk[:, :, :Cc])
        gtq = torch.empty(B, Tk, C1, dtype=torch.float32, device=g.device)
        K.gemm(g, with_ones(q), gtq, Tk, C1, Tq, Tk, C1, C1, False, False, nb0=B, nb1=1, sA=(Tq * Tk, 0), sB=(Tq * C1, 0), sC=(Tk * C1, 0))
        dk = t2 * (k * gtq[:, :, Cc:Cc + 1] - gtq[:, :, :Cc])
        return dq, dk, None


def neg_sqdist(q, k, temp):
    return _NegSqDist.apply(q, k, temp)


def mas_binarize(attn_soft, in_lens, out_lens):
    """binarize_attention_parallel (modules.py:863-872) on the device: attn_soft [B,1,Tm,Ts] -> (hard [B,1,Tm,Ts], dur [B,Ts])."""
    with torch.no_grad():
        hard, dur = K.mas(attn_soft[:, 0].contiguous(), in_lens, out_lens)
    return hard.unsqueeze(1), dur


# ---- liu2021 prosody modelling -----------------------------------------------------------------------------
class _Im2Col3x3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.shape = tuple(x.shape)
        return K.im2col_3x3s2(x)

    @staticmethod
    def backward(ctx, dcol):
        return K.col2im_3x3s2(dcol.contiguous(), *ctx.shape)


def conv2d_3x3s2(x, w, b):
    """nn.Conv2d(kernel 3x3, stride (1,2), padding (1,1)) of ReferenceEncoder (modules.py:351-361) on channel-last x [B,T,W,Cin];
    w [Cout,Cin,3,3] (reference layout), b [Cout] -> [B,T,Wo,Cout].  Patch matrix (csrc/prosody.hip) + GEMM on MFMA: forward,
    dgrad (then col2im) and wgrad all go through ctts_gemm."""
    B, T, W, Cin = x.shape
    Cout = w.shape[0]
    col = _Im2Col3x3s2.apply(x) if x.requires_grad else K.im2col_3x3s2(x.contiguous())
    wf = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin)              # [Cout][kh][kw][Cin]: tiny, autograd un-permutes the gradient
    y = linear(col, wf, b)
    return y.view(B, T, (W - 1) // 2 + 1, Cout)


class _GRU(torch.autograd.Function):
    """Recurrent part of a one-layer nn.GRU (batch_first, h0 = 0, ndir 1|2) on csrc/prosody.hip; gi = W_ih x + b_ih comes from
    `linear` (so W_ih / b_ih / input gradients are ordinary GEMMs)."""

    @staticmethod
    def forward(ctx, gi, whh, bhh, H, ndir, rev_mask=None):
        gi, whh, bhh = gi.contiguous(), whh.contiguous(), bhh.contiguous()
        need = gi.requires_grad or whh.requires_grad or bhh.requires_grad
        out, gates = K.gru_fwd(gi, whh, bhh, H, ndir, save_gates=need, rev_mask=rev_mask)
        ctx.save_for_backward(out, gates, whh)
        ctx.cfg = (H, ndir, rev_mask)
        return out

    @staticmethod
    def backward(ctx, dout):
        out, gates, whh = ctx.saved_tensors
        H, ndir, rev_mask = ctx.cfg
        B, T = out.shape[0], out.shape[1]
        dgi, dgh, hprev = K.gru_bwd(dout.contiguous(), out, gates, whh, H, ndir, rev_mask)
        rows = B * T
        dwhh = torch.empty_like(whh)
        for d in range(ndir):              # dW_hh[d] = dgh_d^T h_prev_d : [3H, H], reduction over all B*T steps (split-K on MFMA)
            K.gemm(dgh, hprev, dwhh, 3 * H, H, rows, ndir * 3 * H, ndir * H, H, False, False, a_off=d * 3 * H, b_off=d * H,
                   c_off=d * 3 * H * H, split_k=max(2, _split_k_for(3 * H, H, rows)), split_overwrite=True)
        dbhh = K.colsum(dgh.view(rows, ndir * 3 * H)).view(ndir, 3 * H)
        return dgi, dwhh, dbhh, None, None, None


def gru_group(gis, w_hhs, b_hhs):
    """Several independent forward-in-time GRUs of equal hidden size and length in ONE launch (the recurrence is latency bound and
    one GRU fills 16 of 256 CUs, so a second one rides along for free): gis[i] [B,T,3H] -> list of outputs [B,T,H]."""
    n, H = len(gis), w_hhs[0].shape[1]
    out = _GRU.apply(torch.stack(gis, 2).flatten(2), torch.stack(w_hhs, 0), torch.stack(b_hhs, 0), H, n, 0)
    return [out[..., i * H:(i + 1) * H] for i in range(n)]


def gru(x, w_ih, w_hh, b_ih, b_hh, w_ih_r=None, w_hh_r=None, b_ih_r=None, b_hh_r=None):
    """nn.GRU(batch_first=True[, bidirectional=True]) forward over ALL T steps (the reference never packs padded batches,
    modules.py:390-391,636-637).  x [B,T,In] -> out [B,T,ndir*H] (forward | backward halves, as torch lays them out)."""
    H = w_hh.shape[1]
    if w_ih_r is None:
        gi = linear(x, w_ih, b_ih)
        return _GRU.apply(gi, w_hh.unsqueeze(0), b_hh.unsqueeze(0), H, 1)
    gi = linear(x, torch.cat([w_ih, w_ih_r], 0), torch.cat([b_ih, b_ih_r], 0))      # one GEMM for both directions
    return _GRU.apply(gi, torch.stack([w_hh, w_hh_r], 0), torch.stack([b_hh, b_hh_r], 0), H, 2)


class _BmmNT(torch.autograd.Function):
    """out[b] = alpha * A[b] @ Bm[b]^T   (A [B,M,K], Bm [B,N,K]) - attention scores of PhonemeLevelProsodyEncoder (modules.py:443)."""

    @staticmethod
    def forward(ctx, A, Bm, alpha):
        A, Bm = A.contiguous(), Bm.contiguous()
        B, M, Kd = A.shape
        N = Bm.shape[1]
        out = torch.empty(B, M, N, dtype=torch.float32, device=A.device)
        K.gemm(A, Bm, out, M, N, Kd, Kd, Kd, N, True, True, nb0=B, nb1=1, sA=(M * Kd, 0), sB=(N * Kd, 0), sC=(M * N, 0), alpha=alpha)
        ctx.save_for_backward(A, Bm)
        ctx.alpha = alpha
        return out

    @staticmethod
    def backward(ctx, dO):
        A, Bm = ctx.saved_tensors
        dO = dO.contiguous()
        B, M, Kd = A.shape
        N = Bm.shape[1]
        dA = dB = None
        if ctx.needs_input_grad[0]:        # dA[m,k] = alpha * sum_n dO[m,n] Bm[n,k]
            dA = torch.empty_like(A)
            K.gemm(dO, Bm, dA, M, Kd, N, N, Kd, Kd, True, False, nb0=B, nb1=1, sA=(M * N, 0), sB=(N * Kd, 0), sC=(M * Kd, 0),
                   alpha=ctx.alpha)
        if ctx.needs_input_grad[1]:        # dB[n,k] = alpha * sum_m dO[m,n] A[m,k]
            dB = torch.empty_like(Bm)
            K.gemm(dO, A, dB, N, Kd, M, N, Kd, Kd, False, False, nb0=B, nb1=1, sA=(M * N, 0), sB=(M * Kd, 0), sC=(N * Kd, 0),
                   alpha=ctx.alpha)
        return dA, dB, None


def bmm_nt(A, Bm, alpha=1.0):
    return _BmmNT.apply(A, Bm, alpha)


class _SoftmaxRect(torch.autograd.Function):
    @staticmethod
    def forward(ctx, S, klens, qlens):
        P = K.softmax_rect_fwd(S.contiguous().clone(), klens, qlens)
        ctx.save_for_backward(P, klens, qlens)
        return P

    @staticmethod
    def backward(ctx, dP):
        P, klens, qlens = ctx.saved_tensors
        return K.softmax_rect_bwd(P, dP.contiguous().clone(), klens, qlens), None, None


def masked_softmax(S, key_lens=None, query_lens=None):
    """softmax over the last dim of S [nb,Tq,Tk]; keys >= key_lens[b] get probability 0 (masked_fill(-inf) before the softmax),
    query rows >= query_lens[b] are zero rows (masked_fill(0) after it) - modules.py:444-446.  lens: int32 device tensors."""
    return _SoftmaxRect.apply(S, key_lens, query_lens)


class _ForwardSum(torch.autograd.Function):
    """per-utterance CTC negative log-likelihood of ForwardSumLoss (loss.py:350-377) - csrc/align.hip, one launch per batch."""

    @staticmethod
    def forward(ctx, attn_logprob, in_lens, out_lens, blank):
        a = attn_logprob.contiguous()
        in32, out32 = in_lens.to(torch.int32).contiguous(), out_lens.to(torch.int32).contiguous()
        nll, lse, alpha = K.forward_sum_fwd(a, in32, out32, blank)
        ctx.save_for_backward(a, in32, out32, lse, alpha, nll)
        ctx.blank = blank
        return nll

    @staticmethod
    def backward(ctx, g):
        a, in32, out32, lse, alpha, nll = ctx.saved_tensors
        return K.forward_sum_bwd(a, in32, out32, ctx.blank, lse, alpha, nll, g.contiguous()), None, None, None


class _ForwardSumLossSide(torch.autograd.Function):
    """mean over utterances of nll_b / K_b (infinite terms zeroed: ForwardSumLoss, loss.py:350-377) with BOTH recursions - the alpha pass
    and the beta pass that yields d loss / d attn_logprob - run inside the FORWARD on the side stream `side`, forked (ops.mark_ready)
    where the aligner produced the log-probabilities: two latency chains of 16 workgroups (0.55 + 0.39 ms) beside the decoder instead
    of in front of the loss and at the head of the backward.  The backward is one multiply on the caller's stream (the loss is linear
    in the upstream gradient), so autograd never crosses streams."""

    @staticmethod
    def forward(ctx, attn_logprob, in_lens, out_lens, blank, side):
        cur = torch.cuda.current_stream()
        with torch.cuda.stream(side):
            a = attn_logprob.contiguous()
            B = a.shape[0]
            in32, out32 = in_lens.to(torch.int32).contiguous(), out_lens.to(torch.int32).contiguous()
            nll, lse, alpha = K.forward_sum_fwd(a, in32, out32, blank)
            finite = ~torch.isinf(nll)                                                  # zero_infinity=True
            kf = in_lens.clamp(min=1).to(torch.float32)
            w = finite.to(torch.float32) / (kf * B)
            loss = K.colsum((torch.where(finite, nll, torch.zeros_like(nll)) / kf).view(-1, 1), scale=1.0 / B)    # ordered sum -> [1]
            grad_unit = K.forward_sum_bwd(a, in32, out32, blank, lse, alpha, nll, w.contiguous())     # d loss / d attn_logprob
        cur.wait_stream(side)
        for t in (loss, grad_unit, a):
            t.record_stream(cur)
        ctx.save_for_backward(grad_unit)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad_unit,) = ctx.saved_tensors
        return grad_unit * g.reshape(()), None, None, None, None


def forward_sum_loss_beside(attn_logprob, in_lens, out_lens, blank_logprob, ready):
    return _ForwardSumLossSide.apply(attn_logprob, in_lens, out_lens, float(blank_logprob), ready)


def forward_sum_nll(attn_logprob, in_lens, out_lens, blank_logprob=-1.0):
    """attn_logprob [B,Tm,Ts] (device) -> nll [B]: -log p(1..K_b | frames) with a blank of log-prob `blank_logprob` prepended
    to every frame's logits and the log-softmax taken over [blank, first K_b tokens]."""
    return _ForwardSum.apply(attn_logprob, in_lens, out_lens, float(blank_logprob))


class _Embedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, weight, padding_idx):
        ids = ids.contiguous()
        ctx.save_for_backward(ids, weight)
        ctx.padding_idx = padding_idx
        return K.embedding_fwd(ids, weight.contiguous())

    @staticmethod
    def backward(ctx, dy):
        ids, weight = ctx.saved_tensors
        if _fusable(weight):
            K.embedding_bwd(ids, dy.contiguous(), weight.shape[0], ctx.padding_idx, acc_into=weight.grad)
            return None, None, None
        return None, K.embedding_bwd(ids, dy.contiguous(), weight.shape[0], ctx.padding_idx), None


def embedding(ids, weight, padding_idx=-1):
    """nn.Embedding(padding_idx=...) lookup (blocks.py:10-15; pitch / energy embeddings modules.py:947,958): gather forward, sort-free
    backward (torch sorts the ids and runs a 110 us kernel for the 16 k pitch ids)."""
    return _Embedding.apply(ids, weight, -1 if padding_idx is None else int(padding_idx))


class _MelL1(torch.autograd.Function):
    """both masked mel L1 terms of CompTransTTSLoss (loss.py:130-138,303-304) in one pass - csrc/optim.hip"""

    @staticmethod
    def forward(ctx, p1, p2, tgt, pad_mask):
        p1, p2, tgt = p1.contiguous(), p2.contiguous(), tgt.contiguous()
        sums, roww = K.mel_l1_fwd(p1, p2, tgt, pad_mask.contiguous().view(torch.uint8))
        ctx.save_for_backward(p1, p2, tgt, roww, sums)
        return sums[:2] / (tgt.shape[-1] * sums[2])

    @staticmethod
    def backward(ctx, g):
        p1, p2, tgt, roww, sums = ctx.saved_tensors
        d1, d2 = K.mel_l1_bwd(p1, p2, tgt, roww, sums, g.contiguous())
        return d1, d2, None, None


def mel_l1_pair(mel_pred, postnet_pred, target, pad_mask):
    """-> tensor [2] = (mel_loss, postnet_mel_loss)"""
    return _MelL1.apply(mel_pred, postnet_pred, target, pad_mask)


class _VarLoss(torch.autograd.Function):
    """duration (phone / word / sentence), cwt, uv, f0-statistics and energy terms of CompTransTTSLoss in one kernel pair - csrc/loss.hip"""

    @staticmethod
    def forward(ctx, log_d, cwt, f0m_p, f0s_p, e_pred, dur, texts, src_pad, cwt_spec, uv, mel_pad, f0m_t, f0s_t, e_tgt, lambdas_t,
                cwt_l2, sil_t):
        c = lambda t: t.contiguous()                                                                    # noqa: E731
        tensors = (c(log_d), c(dur), c(texts), c(src_pad).view(torch.uint8), c(cwt), c(cwt_spec), c(uv), c(mel_pad).view(torch.uint8),
                   c(f0m_p), c(f0m_t), c(f0s_p), c(f0s_t), c(e_pred), c(e_tgt))
        terms, partials, wsum, denoms = K.var_loss_fwd(tensors, lambdas_t, cwt_l2, sil_t)
        ctx.save_for_backward(*tensors, partials, wsum, denoms)
        ctx.cfg = (lambdas_t, cwt_l2, sil_t)
        return terms

    @staticmethod
    def backward(ctx, g):
        *tensors, partials, wsum, denoms = ctx.saved_tensors
        lambdas_t, cwt_l2, sil_t = ctx.cfg
        d_log_d, d_cwt, d_f0m, d_f0s, d_e = K.var_loss_bwd(tuple(tensors), lambdas_t, cwt_l2, sil_t, partials, wsum, denoms, g.contiguous())
        return (d_log_d, d_cwt, d_f0m, d_f0s, d_e) + (None,) * 12


def variance_losses(log_d, cwt, f0_mean, f0_std, e_pred, dur, texts, src_pad, cwt_spec, uv, mel_pad, f0_mean_t, f0_std_t, e_tgt,
                    lambdas_t, cwt_l2, sil_t):
    """-> tensor [8] = (pdur, wdur, sdur, C, uv, f0_mean, f0_std, energy), lambda-weighted (model/loss.py:123-243)"""
    return _VarLoss.apply(log_d, cwt, f0_mean, f0_std, e_pred, dur, texts, src_pad, cwt_spec, uv, mel_pad, f0_mean_t, f0_std_t, e_tgt,
                          lambdas_t, cwt_l2, sil_t)


# ---- low-occupancy work beside the main stream -----------------------------------------------------------------------------------------
# A few kernels of the unsupervised-duration configuration are latency chains on 16 workgroups (one per utterance: the CTC forward-sum
# recursions, 0.55 ms forward + 0.39 ms backward on 256 CUs' worth of chip) whose result nothing needs until the total loss is formed.
# `mark_ready(t)` FORKS a side stream at the point where `t` was produced (record + wait back to back: an event recorded earlier and
# waited for later lost its dependency under hipGraph capture on this stack - the replayed CTC term read half-written log-probabilities),
# the consumer (`take_ready`) launches its chain on that stream - beside the decoder, as a graph branch under capture, by the host running
# ahead in eager mode - and joins before the total loss is formed.  Only while a caller that is known to consume the marker has switched it
# on (`side_loss_scope`, trainer.TrainStep): a fork nobody joins would invalidate a capture.  CTTS_SIDE_LOSS=0 switches it off (A/B).
_READY = {}
_SIDE = {}
SIDE_LOSS = _os.environ.get("CTTS_SIDE_LOSS", "1") != "0"
_SIDE_ACTIVE = [False]


class side_loss_scope:
    """with side_loss_scope(): model forward + loss.  Leaves no unjoined fork behind."""

    def __enter__(self):
        self.prev, _SIDE_ACTIVE[0] = _SIDE_ACTIVE[0], SIDE_LOSS
        return self

    def __exit__(self, *exc):
        _SIDE_ACTIVE[0] = self.prev
        for side in list(_READY.values()) + list(_FORKED):      # a fork nobody joined (e.g. the loss skipped the term): join it here
            torch.cuda.current_stream().wait_stream(side)
        _READY.clear()
        del _FORKED[:]
        return False


def side_stream(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _SIDE.get(idx)
    if st is None:
        st = _SIDE[idx] = torch.cuda.Stream(device=device)
    return st


SIDE_PROSODY = _os.environ.get("CTTS_SIDE_PROSODY", "1") != "0"
_FORKED = []


def fork_side(t):
    """the side stream, forked from the current stream HERE, or None (not inside a side_loss_scope / switched off / host tensors).  The
    caller runs an input-only branch on it (the liu2021 reference encoders read nothing but the target mel: their Conv2d stacks and
    1,000-step GRUs run beside the text encoder, and autograd runs their backward beside the encoder's) and joins with join_side()."""
    if not (_SIDE_ACTIVE[0] and SIDE_PROSODY and t.is_cuda):
        return None
    side = side_stream(t.device)
    side.wait_stream(torch.cuda.current_stream())
    _FORKED.append(side)
    return side


def join_side(side=None):
    """the current stream waits for `side` (default: every side stream of this module - trainer.TrainStep calls it in front of the deferred
    sums of a backward stage, whose partials autograd may have had written on a side stream)"""
    streams = [side] if side is not None else list(_SIDE.values())
    if not streams:
        return
    cur = torch.cuda.current_stream()
    for st in streams:
        cur.wait_stream(st)
        while st in _FORKED:
            _FORKED.remove(st)


def mark_ready(t):
    if _SIDE_ACTIVE[0] and t.is_cuda and not _READY:
        side = side_stream(t.device)
        side.wait_stream(torch.cuda.current_stream())
        _READY[t.data_ptr()] = side


def take_ready(t):
    """the side stream forked where `t` was produced, or None"""
    return _READY.pop(t.data_ptr(), None) if t.is_cuda else None


class _AddOverTime(torch.autograd.Function):
    """x [B,T,C] + v [B,1,C] broadcast over T (speaker / utterance-prosody embeddings added to every position: modules.py:1008,1027,
    CompTransTTS.py:101).  Forward is the plain broadcast add; the backward's sum over T - a stock-torch reduction when autograd does
    it - is one batched [1,T] x [T,C] product on ctts_gemm (fixed summation order)."""

    @staticmethod
    def forward(ctx, x, v):
        ctx.T = x.shape[1]
        return x + v

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        B, T, Cc = g.shape
        dv = None
        if ctx.needs_input_grad[1]:
            ones = torch.ones(B, 1, T, dtype=torch.float32, device=g.device)
            dv = torch.empty(B, 1, Cc, dtype=torch.float32, device=g.device)
            K.gemm(ones, g, dv, 1, Cc, T, T, Cc, Cc, True, False, nb0=B, nb1=1, sA=(T, 0), sB=(T * Cc, 0), sC=(Cc, 0))
        return (g if ctx.needs_input_grad[0] else None), dv


def add_over_time(x, v):
    """x [B,T,C] + v [B,1,C] (v may also be [B,C])"""
    if v.dim() == 2:
        v = v.unsqueeze(1)
    return _AddOverTime.apply(x, v)


class _SumAll(torch.autograd.Function):
    """scale * sum of all elements -> shape [1], through the ordered column-sum kernel (csrc/elementwise.hip): the handful of scalar
    reductions of the loss glue (sum of the eight variance terms, mean over utterances of the CTC term, the `ph` variant's all-padding
    flag) without torch's reduce_kernel - none of which may sit on a captured path"""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.shape, ctx.scale = x.shape, float(scale)
        return K.colsum(x.contiguous().view(-1, 1), scale=float(scale))

    @staticmethod
    def backward(ctx, g):
        return (g.reshape(1) * ctx.scale).expand(ctx.shape), None


def sum_all(x, scale=1.0):
    return _SumAll.apply(x, scale)


class _MaskedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, weight, kind):
        pred, target, weight = pred.contiguous(), target.contiguous().float(), weight.contiguous().float()
        out2 = K.masked_loss_fwd(pred, target, weight, kind)
        ctx.save_for_backward(pred, target, weight, out2)
        ctx.kind = kind
        return out2[0]

    @staticmethod
    def backward(ctx, g):
        pred, target, weight, out2 = ctx.saved_tensors
        return K.masked_loss_bwd(pred, target, weight, out2, g.reshape(1).contiguous(), ctx.kind), None, None, None


def masked_loss(pred, target, weight, kind="l1"):
    """sum(w * l(pred, target)) / sum(w) with l = l1 / l2 / bce (with logits): the f0 and uv terms of pitch_type "frame" / "ph"
    (loss.py:173-178,206-219) and the frame-level energy term (loss.py:238-242); ordered two-stage reduction (csrc/loss.hip)"""
    return _MaskedLoss.apply(pred, target, weight, {"l1": K.MASKED_L1, "l2": K.MASKED_L2, "bce": K.MASKED_BCE}[kind])


class _BinLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, soft, hard):
        soft, hard = soft.contiguous(), hard.contiguous()
        out2 = K.bin_loss_fwd(soft, hard)
        ctx.save_for_backward(soft, hard, out2)
        return out2[0]

    @staticmethod
    def backward(ctx, g):
        soft, hard, out2 = ctx.saved_tensors
        return K.bin_loss_bwd(soft, hard, out2, g.reshape(1).contiguous()), None


def bin_loss(hard, soft):
    """BinLoss (loss.py:380-386): -sum(log(clamp(soft, 1e-12)) * hard) / sum(hard), deterministic two-stage reduction"""
    return _BinLoss.apply(soft, hard)
