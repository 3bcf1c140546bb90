"""Thin tensor-level wrappers over the C ABI (one Python function per exported kernel).

Tensors are only used for device memory + the current stream; every call goes through
libctts_hip.so.  Inputs must live on a HIP device ("cuda" in PyTorch-ROCm) - a CPU tensor
raises: there is no CPU path in the product.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import GemmDesc

ACT_NONE, ACT_RELU, ACT_GELU, ACT_TANH, ACT_SWISH = 0, 1, 2, 3, 4


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t, offset=0):
    """device pointer of tensor `t` advanced by `offset` elements (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.CttsError("ctts kernels need device (HIP) tensors; got a CPU tensor - no CPU fallback exists")
    return t.data_ptr() + offset * t.element_size()


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise _lib.CttsError(f"{name}: expected contiguous float32 tensor, got {t.dtype} contiguous={t.is_contiguous()}")
    return t


def _dp_rank():
    import os
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank()
    return int(os.environ.get("RANK", "0"))


class DropCtx:
    """Dropout bookkeeping: device-resident 64-bit seed + per-call-site offsets.

    The seed tensor is bumped once per step *on the device* (`advance`), so a captured
    hipGraph draws fresh masks on every replay; offsets are plain host constants assigned
    in program order, identical between eager and captured runs."""

    def __init__(self, device, seed=None):
        if seed is None:
            # follows torch.manual_seed (like nn.Dropout would) and differs per data-parallel rank: ranks that were seeded identically
            # for identical initial weights must still draw different masks
            seed = (torch.initial_seed() + 0x51ED27 * _dp_rank()) & 0x3FFFFFFFFFFF
        self.seed = torch.tensor([(int(seed) * 0x9E3779B1 + 0x7F4A7C15) & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device=device)
        self.counter = 0

    def next_offset(self):
        self.counter = (self.counter + 1) & 0x7FFFFFFF
        return self.counter

    def begin_step(self):
        self.counter = 0

    def advance(self):
        self.seed.add_(0x632BE5AB)


import os as _os

SK_ENABLED = _os.environ.get("CTTS_SK", "1") != "0"      # persistent stream-K GEMM (csrc/gemm_sk.hip) for large unbatched launches
BF16_SPLIT = 0 if _os.environ.get("CTTS_X6", "1") == "0" else 1          # default ctts_gemm_desc.bf16_split of this module's descriptors
PLW_ENABLED = _os.environ.get("CTTS_PLW", "1") != "0"        # weight gradients on the plane kernel (csrc/gemm_plw.hip)
PLANES_ENABLED = _os.environ.get("CTTS_PL", "1") != "0"                  # pre-split operand planes + the persistent plane kernel (csrc/gemm_pl.hip)
PLANES_MIN_UNITS = int(_os.environ.get("CTTS_PL_MIN_UNITS", "4096"))
PLANES_MIN_TILES = int(_os.environ.get("CTTS_PL_MIN_TILES", "64"))
_SK_WS = {}          # (device index, stream handle) -> zero-filled workspace (include/ctts.h ctts_workspace_bytes)


def gemm_workspace(device):
    """The per-stream workspace of the library (include/ctts.h: ctts_gemm_desc.sk_ws and every `ws` argument): stream-K flags and slabs,
    split-K tickets and partial tiles, tickets and partials of the ordered column reductions.  One per (device, stream), because launches
    that share it must be stream-ordered.  Zero-filled once; every kernel leaves its flag / ticket words at zero when it finishes."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    ws = _SK_WS.get(key)
    if ws is None:
        if SK_ENABLED and not torch.cuda.is_current_stream_capturing():
            check_xcd_dispatch(device)
        ws = torch.zeros(_lib.load().ctts_workspace_bytes(), dtype=torch.uint8, device=device)
        _SK_WS[key] = ws
    return ws


_XCD_OK = {}


def check_xcd_dispatch(device):
    """Once per device: workgroup b of a launch must run on XCD b % 8 (SPX, the default compute partition of an MI355X) - the schedule
    and the slab hand-off of the persistent stream-K GEMM are built on it (nothing asserted the mode).  Any other mapping
    switches that kernel OFF for the process (the tile kernels take its launches) and says so once."""
    global SK_ENABLED
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx in _XCD_OK:
        return _XCD_OK[idx]
    out = torch.full((64,), -1, dtype=torch.int32, device=device)
    _lib.check(_lib.load().ctts_xcd_probe(_p(out), 64, _stream()), "ctts_xcd_probe")
    got = out.cpu().tolist()
    ok = got == [b % 8 for b in range(64)]
    _XCD_OK[idx] = ok
    if not ok:
        import warnings
        SK_ENABLED = False
        warnings.warn(f"ctts_amd: workgroups are not dispatched round-robin over 8 XCDs on cuda:{idx} (XCC_ID of workgroups 0..15: {got[:16]}) - "
                      "not the SPX partition mode; the persistent stream-K GEMM is disabled, the tile kernels take its launches")
    return ok


workspace = gemm_workspace


def _ws(t):
    """device pointer of the current stream's workspace on t's device"""
    return gemm_workspace(t.device).data_ptr()


class WorkspaceErrorProbe:
    """Host-side watch on the stream-K hand-off error word of every workspace in use (VERDICT r03 #2 / ADVICE r03: the owner of a cut
    tile that gives up waiting writes that word and the launch's result is garbage - nobody but the tests used to read it).
    `poll()` starts an asynchronous copy of the words into pinned memory (no sync: callable every step); `check()` waits for the last
    copy and raises CttsError if any word is set (and re-zeroes that workspace, so a later launch does not trip over a stale flag)."""

    def __init__(self):
        self._pending = []          # (workspace, pinned host word, event)

    def poll(self):
        lib = _lib.load()
        self._pending = []
        for ws in list(_SK_WS.values()):
            off = lib.ctts_workspace_error_word(C.c_void_p(ws.data_ptr())) - ws.data_ptr()
            host = torch.empty(4, dtype=torch.uint8, pin_memory=True)
            with torch.cuda.device(ws.device):
                host.copy_(ws[off:off + 4], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            self._pending.append((ws, host, ev))

    def check(self):
        bad = []
        for ws, host, ev in self._pending:
            ev.synchronize()
            word = int(host.view(torch.int32)[0])
            if word != 0:
                bad.append((ws, word))
        self._pending = []
        for ws, word in bad:
            torch.cuda.synchronize(ws.device)
            ws.zero_()
        if bad:
            raise _lib.CttsError("stream-K GEMM hand-off failed: an owner workgroup gave up waiting for workgroup "
                                 f"{bad[0][1] - 1}'s partial tile - the results of that launch are invalid (workspace re-zeroed)")

    def poll_and_check(self):
        self.poll()
        self.check()


class PartialSink:
    """Deferred ordered reductions of a backward stage (include/ctts.h ctts_partial_sums).  While a sink is installed
    (`set_partial_sink`; trainer.TrainStep does it around every backward stage), the wrappers below that accumulate a cross-workgroup sum
    into `param.grad` - split-K weight gradients (`gemm(..., defer=True)`), bias / LayerNorm column sums - only have their kernels WRITE
    the per-workgroup partials into a scratch tensor and register (partials, count, destination) here; `flush()` then finishes all of
    them with one launch per 24 tasks, each sum in index order (bit-reproducible).  Replaces one reduce launch or one ticket tail per
    layer (fs2: ~160 per step).  The destination must not be READ before flush() - TrainStep flushes at the end of every stage, before
    the stage's gradient bucket is all-reduced."""

    # Partials are consumed while they are still in the 256 MB memory-side cache: once this many bytes are pending the sink flushes by
    # itself (a flush is legal at any point of the stage - only the END-of-stage flush is mandatory).  0 = only at the end of the stage.
    # Measured (fs2, same box): 0 / 64 / 128 / 256 MB -> 23.69 / 23.79 / 23.76 / 23.68 ms: no effect on time, the threshold only bounds the scratch.
    FLUSH_BYTES = int(float(_os.environ.get("CTTS_DEFER_FLUSH_MB", "256")) * (1 << 20))

    def __init__(self):
        self.tasks = []
        self.keep = []
        self.pending = 0
        self.streams = []          # streams the pending partials were produced on (a flush must be ordered behind ALL of them)

    def add(self, src, count, stride, n, dst, alpha, src_off=0, dst_off=0):
        d0 = dst.data_ptr() + 4 * int(dst_off)
        # a launch runs its tasks side by side as plain read-modify-write: two pending sums into overlapping destinations (a parameter
        # used by two call sites of one backward stage, tied weights) must not share a launch - finish the pending ones first
        if any(d0 < t[1] + 4 * t[2] and t[1] < d0 + 4 * int(n) for t in self.tasks):
            self.flush()
        self.tasks.append((src.data_ptr() + 4 * int(src_off), d0, int(n), int(stride), int(count), float(alpha)))
        self.keep.append((src, dst))
        if src.is_cuda:
            st = torch.cuda.current_stream(src.device)
            if st not in self.streams:
                self.streams.append(st)
        self.pending += 4 * int(count) * int(n)
        if self.FLUSH_BYTES and self.pending >= self.FLUSH_BYTES:
            self.flush()

    def flush(self):
        if not self.tasks:
            return
        arr = (_lib.PsumTask * len(self.tasks))()
        for t, (src, dst, n, stride, count, alpha) in zip(arr, self.tasks):
            t.src, t.dst, t.n, t.stride, t.count, t.alpha = src, dst, n, stride, count, alpha
        self.tasks = []
        self.pending = 0
        producers, self.streams = self.streams, []
        try:
            # The partials of a stage are written by whichever stream autograd ran the producing node on (main stream, the side stream
            # of an input-only branch).  A flush - the mandatory one at the end of the stage, or one triggered inside a node by the byte
            # threshold / an overlapping destination - reads ALL pending partials: the flushing stream waits for every other producer
            # stream first (a graph edge under capture), whichever stream it happens to be.
            cur = torch.cuda.current_stream()
            for st in producers:
                if st != cur:
                    cur.wait_stream(st)
            _lib.check(_lib.load().ctts_partial_sums(arr, len(arr), _stream()), "ctts_partial_sums")
            for src, _ in self.keep:                 # partials a side stream allocated are read by THIS stream's launch: tell the allocator
                if src.is_cuda:
                    src.record_stream(cur)
        finally:
            self.keep = []


# Bumped by every update of the parameters that autograd's version counters cannot see (dp.FlatAdam writes through raw pointers):
# per-step caches made from the weights (ops._DGRAD_W, ops._PLANES) carry the epoch they were made in.
WEIGHTS_EPOCH = [0]

_SINK = None
DEFER_ENABLED = _os.environ.get("CTTS_DEFER_SUMS", "1") != "0"       # A/B switch: 0 = every reduction finishes inside its own call


def set_partial_sink(sink):
    """install (or remove: None) the PartialSink of the current backward stage; returns the previous one"""
    global _SINK
    prev, _SINK = _SINK, (sink if DEFER_ENABLED else None)
    return prev


def _sink_for(dst):
    """the active sink if `dst` (a tensor the caller wants a sum ADDED to) is dense in memory, else None"""
    if _SINK is None or dst is None or not dst.is_contiguous():
        return None
    return _SINK


def _gemm_desc(A, B, Cout, M, N, K, lda, ldb, ldc, a_kc=True, b_kc=True, a_off=0, b_off=0, c_off=0, nb0=1, nb1=1,
               sA=(0, 0), sB=(0, 0), sC=(0, 0), lens=None, lim=(0, 0, 0), conv=None, conv_on_b=False, split_k=1, alpha=1.0,
               bias=None, Z=None, ldz=0, act=ACT_NONE, p_drop=0.0, seed=None, drop_offset=0, R=None, ldr=0, rowscale=None,
               row_lens=None, row_T=0, row_halo=0, tile_map=None, E=None, rowsub=None, use_sk=None, epi_bwd=False, split_overwrite=False,
               bf16_split=None, a_planes=None, b_planes=None, c_planes=None):
    d = GemmDesc()
    if c_planes is not None:
        d.C_planes = c_planes.data_ptr()
    d.bf16_split = int(BF16_SPLIT if bf16_split is None else bf16_split)
    if a_planes is not None and b_planes is not None:
        d.A_planes, d.B_planes = a_planes.data_ptr(), b_planes.data_ptr()
    d.A, d.B, d.C = _p(A, a_off), _p(B, b_off), _p(Cout, c_off)
    d.M, d.N, d.K = int(M), int(N), int(K)
    d.lda, d.ldb, d.ldc = int(lda), int(ldb), int(ldc)
    d.a_kc, d.b_kc = int(a_kc), int(b_kc)
    d.nb0, d.nb1 = int(nb0), int(nb1)
    d.sA0, d.sA1 = int(sA[0]), int(sA[1])
    d.sB0, d.sB1 = int(sB[0]), int(sB[1])
    d.sC0, d.sC1 = int(sC[0]), int(sC[1])
    d.lens = _p(lens)
    d.lim_m, d.lim_n, d.lim_k = int(lim[0]), int(lim[1]), int(lim[2])
    if conv is not None:
        d.conv_T, d.conv_pad, d.conv_cin = int(conv[0]), int(conv[1]), int(conv[2])
        d.conv_on_b = int(conv_on_b)
    d.split_k = int(split_k)
    d.alpha = float(alpha)
    d.bias = _p(bias)
    d.Z, d.ldz = _p(Z), int(ldz)
    d.act = int(act)
    d.p_drop = float(p_drop)
    d.seed = _p(seed)
    d.drop_offset = int(drop_offset)
    d.R, d.ldr = _p(R), int(ldr)
    d.rowscale = _p(rowscale)
    if row_lens is not None:
        d.row_lens, d.row_T, d.row_halo = _p(row_lens), int(row_T), int(row_halo)
        d.tile_map = _p(tile_map)
    d.E, d.rowsub = _p(E), _p(rowsub)
    d.epi_bwd = int(bool(epi_bwd))
    d.split_overwrite = int(bool(split_overwrite))
    if int(split_k) > 1 or a_planes is not None or ((SK_ENABLED if use_sk is None else use_sk) and nb0 * nb1 == 1 and M * N * K >= (1 << 24)):
        # split-K sums its pieces in a fixed order through the workspace (required); large unbatched GEMMs may run on the persistent
        # stream-K kernel (the library decides: ctts_gemm_sk_try)
        ws = gemm_workspace(A.device)
        d.sk_ws, d.sk_ws_bytes = ws.data_ptr(), ws.numel()
    return d


def gemm(A, B, Cout, M, N, K, lda, ldb, ldc, a_kc=True, b_kc=True, defer=False, **kw):
    """C = epi(alpha * (opA @ opB + bias)); see include/ctts.h ctts_gemm_desc (keyword arguments: _gemm_desc).
    defer=True (split_k > 1, C a dense [M, N] accumulation target such as param.grad): with a PartialSink installed the split-K partial
    matrices stay in a scratch tensor and are added to C at the sink's flush instead of by a reduce launch of their own."""
    d = _gemm_desc(A, B, Cout, M, N, K, lda, ldb, ldc, a_kc, b_kc, **kw)
    lib = _lib.load()
    if defer and _SINK is not None and d.split_k > 1 and d.nb0 * d.nb1 == 1 and ldc == N and N % 4 == 0:
        cnt, stride = C.c_int32(0), C.c_int64(0)
        if lib.ctts_gemm_split_plan(C.byref(d), C.byref(cnt), C.byref(stride)) == 1:
            P = torch.empty(d.split_k * stride.value, dtype=torch.float32, device=A.device)
            d.split_out, d.split_out_floats = P.data_ptr(), P.numel()
            _lib.check(lib.ctts_gemm(C.byref(d), _stream()), "ctts_gemm")
            _SINK.add(P, cnt.value, stride.value, M * N, Cout, d.alpha, dst_off=kw.get("c_off", 0))
            return Cout
    _lib.check(lib.ctts_gemm(C.byref(d), _stream()), "ctts_gemm")
    return Cout


def gemm_takes_persistent(A, B, Cout, M, N, K, lda, ldb, ldc, a_kc=True, b_kc=True, **kw):
    """True when ctts_gemm would run these arguments on the persistent stream-K kernel (no launch)."""
    if not SK_ENABLED:
        return False
    d = _gemm_desc(A, B, Cout, M, N, K, lda, ldb, ldc, a_kc, b_kc, **kw)
    return bool(_lib.load().ctts_gemm_takes_persistent(C.byref(d)))


def gemm_takes_bf16_split(A, B, Cout, M, N, K, lda, ldb, ldc, a_kc=True, b_kc=True, **kw):
    """True when ctts_gemm would run these arguments on the fp32-on-bf16-pipe kernel (six-term operand split; no launch)."""
    d = _gemm_desc(A, B, Cout, M, N, K, lda, ldb, ldc, a_kc, b_kc, **kw)
    return bool(_lib.load().ctts_gemm_takes_bf16_split(C.byref(d)))


def gemm_bf16_split_enable(on):
    """Default arithmetic of the descriptors this module builds (ctts_gemm_desc.bf16_split; env CTTS_X6=0 starts with it off): False =
    fp32 MFMA only, True = the large launches may run on the bf16 matrix pipe with the exact six-term operand split, 2 = also launches
    below the kernels' size thresholds (tests); "amp" (= 3; 4 without the thresholds) = the plane kernels round their operands to bf16
    and use ONE MFMA term (reference train.py:59,104 `amp.autocast` - reduced precision, never the default: see amp_split).  Returns
    the previous setting (pass it back to restore).  The library itself holds no such state any more: the choice travels in every
    descriptor, so a captured graph keeps the arithmetic it was captured with and two models can differ (`bf16_split=` on a single call
    overrides the default)."""
    global BF16_SPLIT
    prev = BF16_SPLIT
    BF16_SPLIT = 3 if on == "amp" else (int(on) if on in (2, 3, 4) else int(bool(on)))
    return prev


def amp_split():
    """`bf16_split` for a launch made under torch.amp.autocast (ops._LinearConv asks torch.is_autocast_enabled): the one-term arithmetic
    with the current setting's thresholds, or None when the bf16 kernels are switched off altogether (CTTS_X6=0 / enable(False))."""
    if BF16_SPLIT < 1:
        return None
    return 4 if BF16_SPLIT in (2, 4) else 3


def gemm_takes_planes(A, B, Cout, M, N, K, lda, ldb, ldc, a_kc=True, b_kc=True, **kw):
    """True when ctts_gemm would run these arguments (a_planes / b_planes given) on the persistent plane kernel (no launch)."""
    d = _gemm_desc(A, B, Cout, M, N, K, lda, ldb, ldc, a_kc, b_kc, **kw)
    return bool(_lib.load().ctts_gemm_takes_planes(C.byref(d)))


def plane_shape_ok(M, N, K, conv_cin=None):
    """Host-side pre-filter of the plane kernel's shape rules (csrc/gemm_pl.hip pl_try) - callers use it to decide whether splitting
    an operand is worth a launch; the library's answer (gemm_takes_planes) stays authoritative."""
    if not PLANES_ENABLED or BF16_SPLIT < 1 or not SK_ENABLED:
        return False
    if K % 32 or K < 64 or N % 128 or N < 256 or M < 128:
        return False
    if conv_cin is not None and (conv_cin % 32 or K % conv_cin):
        return False
    tiles = -(-M // 128) * -(-N // 256)
    return BF16_SPLIT in (2, 4) or (tiles * (K // 32) >= PLANES_MIN_UNITS and tiles >= PLANES_MIN_TILES)


def plane_wgrad_shape_ok(Mo, No, Kred, cin=None):
    """The same pre-filter for the weight-gradient plane kernel (csrc/gemm_plw.hip plw_try): output [Mo, No] = [cout, k * cin], reduction
    over Kred (b, t) rows."""
    if not PLANES_ENABLED or not PLW_ENABLED or BF16_SPLIT < 1 or not SK_ENABLED:
        return False
    if Mo % 128 or No % 256 or Kred < 64:
        return False
    if cin is not None and (cin % 256 or No % cin):
        return False
    tiles = (Mo // 128) * (No // 256)
    return BF16_SPLIT in (2, 4) or (tiles >= 32 and tiles * (-(-Kred // 32)) >= PLANES_MIN_UNITS)


def split_planes(mats):
    """Exact three-way bf16 split of fp32 matrices in one launch per 32 (include/ctts.h ctts_split_planes): `mats` = list of dense 2-D
    float32 tensors [rows, cols] (cols % 32 == 0) -> list of bf16 tensors [rows, cols / 32, 3, 32]: pieces hi | mid | lo of every 32-deep
    K-block of a row side by side (the layout the plane kernel's DMA reads; `planes_piece` gives a piece back as [rows, cols])."""
    if not mats:
        return []
    outs = []
    arr = (_lib.SplitTask * len(mats))()
    for t, m in zip(arr, mats):
        _f32c(m, "split_planes operand")
        rows, cols = m.shape
        if cols % 32:
            raise _lib.CttsError(f"split_planes: cols = {cols} is not a multiple of 32")
        o = torch.empty(rows, cols // 32, 3, 32, dtype=torch.bfloat16, device=m.device)
        outs.append(o)
        t.src, t.dst, t.rows, t.cols, t.ld = _p(m), o.data_ptr(), int(rows), int(cols), int(cols)
    _lib.check(_lib.load().ctts_split_planes(arr, len(arr), _stream()), "ctts_split_planes")
    return outs


# ---- plane sets written by PRODUCERS (round 6) ------------------------------------------------------------------------------------------
# A kernel that holds an activation in registers (LayerNorm forward, BatchNorm apply / backward, the producer-epilogue backward of the
# K = 256 GEMM) can write the activation's bf16 plane set in the same launch; the set travels WITH the fp32 tensor as a Python attribute
# (autograd keeps the Python object of a tensor alive across Function boundaries, forward and backward), validated against the tensor's
# address, size and version when a GEMM asks for it.  CTTS_PRODUCER_PLANES=0: every consumer splits its own operand again (A/B switch).
PRODUCER_PLANES = _os.environ.get("CTTS_PRODUCER_PLANES", "1") != "0"


def _al16(*ts):
    return all(t.data_ptr() % 16 == 0 for t in ts)


def new_planes(rows, cols, device):
    return torch.empty(rows, cols // 32, 3, 32, dtype=torch.bfloat16, device=device)


def attach_planes(t, pl):
    t._ctts_planes = (pl, t.data_ptr(), t._version, t.numel())
    return t


def planes_of(t):
    """the plane set a producer attached to `t`, or None (never attached / `t` was written since / another tensor)"""
    ent = getattr(t, "_ctts_planes", None)
    if ent is None:
        return None
    pl, ptr, ver, n = ent
    if ptr != t.data_ptr() or ver != t._version or n != t.numel() or pl.numel() != 3 * n:
        return None
    return pl


def planes_piece(pl, q):
    """piece q (0 hi, 1 mid, 2 lo) of a plane set as a [rows, cols] bf16 tensor"""
    return pl[:, :, q, :].reshape(pl.shape[0], -1)


def gemm_takes_weight_stationary(A, B, Cout, M, N, K, lda, ldb, ldc, a_kc=True, b_kc=True, **kw):
    """True when ctts_gemm would run these arguments on the weight-stationary K = 256 kernel (no launch)."""
    d = _gemm_desc(A, B, Cout, M, N, K, lda, ldb, ldc, a_kc, b_kc, **kw)
    return bool(_lib.load().ctts_gemm_takes_weight_stationary(C.byref(d)))


def gemm_ws_enable(on):
    """Switch of the weight-stationary K = 256 kernel (csrc/gemm_ws.hip); returns the previous setting.  Tests / A-B timing only."""
    return bool(_lib.load().ctts_gemm_ws_enable(1 if on else 0))


def row_tile_map(row_lens, row_T, row_halo, M):
    """m-tile schedule (active 64-row tiles first) for GEMMs over padded (b,t) rows; int32 [1 + ceil(M/64)]"""
    tm = torch.empty(1 + (M + 63) // 64, dtype=torch.int32, device=row_lens.device)
    lib = _lib.load()
    _lib.check(lib.ctts_row_tile_map(_p(row_lens), int(row_T), int(row_halo), int(M), _p(tm), _stream()), "ctts_row_tile_map")
    return tm


def conv_dgrad_weights(weights):
    """Data-gradient operands [Cin, K * Cout] of many Conv1d weights (GEMM-major forward layout, viewed [Cout, K * Cin]) in one launch
    per 32 layers: `weights` = list of (wmaj [Cout, K*Cin], cout, cin, k) -> list of wd tensors (include/ctts.h ctts_conv_dgrad_weights)."""
    if not weights:
        return []
    outs = [torch.empty(cin, k * cout, dtype=torch.float32, device=w.device) for w, cout, cin, k in weights]
    arr = (_lib.RepackTask * len(weights))()
    for t, (w, cout, cin, k), o in zip(arr, weights, outs):
        t.src, t.dst, t.cout, t.cin, t.k = _p(_f32c(w, "conv weight (GEMM-major)")), _p(o), int(cout), int(cin), int(k)
    _lib.check(_lib.load().ctts_conv_dgrad_weights(arr, len(arr), _stream()), "ctts_conv_dgrad_weights")
    return outs


def conv_weight_repack(src, dst, cout, cin, k, mode):
    lib = _lib.load()
    _lib.check(lib.ctts_conv_weight_repack(_p(src), _p(dst), cout, cin, k, mode, _stream()), "ctts_conv_weight_repack")
    return dst


def lr_index(dur, Tm, pad=None, round_mode=0, want_mel2ph=True):
    """-> (mel2ph int32 [B,Tm] or None, mel_len int64 [B], cum int32 [B,Ts])"""
    B, Ts = dur.shape
    is_float = dur.dtype == torch.float32
    if not is_float and dur.dtype != torch.int64:
        raise _lib.CttsError(f"lr_index: durations must be int64 or float32, got {dur.dtype}")
    dur = dur.contiguous()
    mel2ph = torch.empty(B, Tm, dtype=torch.int32, device=dur.device) if want_mel2ph else None
    mel_len = torch.empty(B, dtype=torch.int64, device=dur.device)
    cum = torch.empty(B, Ts, dtype=torch.int32, device=dur.device)
    padp = None
    if pad is not None:
        pad = pad.to(torch.uint8).contiguous()
        padp = _p(pad)
    lib = _lib.load()
    _lib.check(lib.ctts_lr_index(_p(dur), int(is_float), int(round_mode), padp, B, Ts, int(Tm), _p(mel2ph), _p(mel_len),
                                 _p(cum), _stream()), "ctts_lr_index")
    return mel2ph, mel_len, cum


def lr_gather_fwd(x, mel2ph):
    B, Ts, Cc = x.shape
    Tm = mel2ph.shape[1]
    out = torch.empty(B, Tm, Cc, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    _lib.check(lib.ctts_lr_gather_fwd(_p(_f32c(x, "x")), _p(mel2ph), _p(out), B, Ts, Tm, Cc, _stream()), "ctts_lr_gather_fwd")
    return out


def lr_gather_bwd(dy, cum, Ts):
    B, Tm, Cc = dy.shape
    dx = torch.empty(B, Ts, Cc, dtype=torch.float32, device=dy.device)
    lib = _lib.load()
    _lib.check(lib.ctts_lr_gather_bwd(_p(_f32c(dy, "dy")), _p(cum), _p(dx), B, Ts, Tm, Cc, _stream()), "ctts_lr_gather_bwd")
    return dx


def positions(src, stride=1):
    """src: int64 tokens [B,T] or float32 [B,T,C] (channel 0 is tested, stride=C)."""
    B, T = src.shape[0], src.shape[1]
    pos = torch.empty(B, T, dtype=torch.int32, device=src.device)
    is_float = src.dtype == torch.float32
    lib = _lib.load()
    _lib.check(lib.ctts_positions(_p(src), int(is_float), int(stride), B, T, _p(pos), _stream()), "ctts_positions")
    return pos


def cwt_pitch(spec, f0_mean, f0_std, std_scale, eps, mel_min, mel_max, f0_bin, uv=None, uv_chan=-1, nscale=10, width=None):
    """include/ctts.h ctts_cwt_pitch: spec [B,T,C>=nscale] -> (f0 [B,width] float (log2 domain), f0_denorm [B,width] float, ids [B,width] int64)"""
    spec = _f32c(spec, "cwt_pitch spec")
    B, T, ld = spec.shape
    width = T if width is None else int(width)
    f0 = torch.empty(B, width, dtype=torch.float32, device=spec.device)
    den = torch.empty_like(f0)
    ids = torch.empty(B, width, dtype=torch.int64, device=spec.device)
    if uv is not None:
        uv = _f32c(uv if uv.dtype == torch.float32 else uv.float(), "cwt_pitch uv")
        if tuple(uv.shape) != (B, width):
            raise _lib.CttsError(f"cwt_pitch: uv has shape {tuple(uv.shape)}, expected {(B, width)}")
    f0_mean, f0_std = _f32c(f0_mean.contiguous(), "f0_mean"), _f32c(f0_std.contiguous(), "f0_std")      # named: the copies must outlive the launch
    _lib.check(_lib.load().ctts_cwt_pitch(_p(spec), ld, int(nscale), _p(f0_mean), _p(f0_std), float(std_scale),
                                          _p(uv), int(uv_chan), float(eps), float(mel_min), float(mel_max), int(f0_bin), _p(f0), _p(den), _p(ids),
                                          B, T, width, _stream()), "ctts_cwt_pitch")
    return f0, den, ids


def layernorm_fwd(x, gamma, beta, eps, p_drop=0.0, seed=None, drop_offset=0, rowscale=None, want_planes=False):
    """want_planes: also write the bf16 plane set of y (same launch) and attach it to y (`planes_of(y)`)"""
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    pl = new_planes(rows, Cc, x.device) if (want_planes and PRODUCER_PLANES and Cc % 32 == 0 and x.is_cuda) else None
    lib = _lib.load()
    _lib.check(lib.ctts_layernorm_fwd(_p(_f32c(x, "x")), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, Cc, eps,
                                      p_drop, _p(seed), drop_offset, _p(rowscale), _p(pl), _stream()), "ctts_layernorm_fwd")
    if pl is not None:
        attach_planes(y, pl)
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, p_drop=0.0, seed=None, drop_offset=0, rowscale=None, acc_into=None, dres=None):
    """acc_into=(dgamma_buf, dbeta_buf): accumulate the parameter gradients into existing buffers (param.grad)."""
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    dx = torch.empty_like(x)
    if acc_into is None:
        dgamma = torch.empty(Cc, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(Cc, dtype=torch.float32, device=x.device)
    else:
        dgamma, dbeta = acc_into
    lib = _lib.load()
    sink = _sink_for(dgamma) if (acc_into is not None and dbeta.is_contiguous()) else None
    nparts = lib.ctts_reduce_parts(3, rows, Cc) if sink is not None else 0
    if nparts > 1:              # deferred: the kernel only writes its per-workgroup partial rows [dgamma | dbeta]
        parts = torch.empty(nparts, 2 * Cc, dtype=torch.float32, device=x.device)
        _lib.check(lib.ctts_layernorm_bwd(_p(_f32c(dy, "dy")), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx), None, None, rows, Cc, p_drop,
                                          _p(seed), drop_offset, _p(rowscale), 1, _p(dres), None, _p(parts), _stream()), "ctts_layernorm_bwd")
        sink.add(parts, nparts, 2 * Cc, Cc, dgamma, 1.0)
        sink.add(parts, nparts, 2 * Cc, Cc, dbeta, 1.0, src_off=Cc)
        return dx, dgamma, dbeta
    _lib.check(lib.ctts_layernorm_bwd(_p(_f32c(dy, "dy")), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dgamma),
                                      _p(dbeta), rows, Cc, p_drop, _p(seed), drop_offset, _p(rowscale), int(acc_into is not None),
                                      _p(dres), _ws(dy), None, _stream()), "ctts_layernorm_bwd")
    return dx, dgamma, dbeta


def colstats(x2d):
    rows, Cc = x2d.shape
    sums = torch.empty(2 * Cc, dtype=torch.float64, device=x2d.device)
    lib = _lib.load()
    _lib.check(lib.ctts_colstats(_p(_f32c(x2d, "x")), _p(sums), rows, Cc, _ws(x2d), _stream()), "ctts_colstats")
    return sums


def bn_batch_stats(x2d, eps, momentum, running_mean, running_var, num_batches):
    """train-mode BatchNorm statistics of x2d [rows,C] (+ running-stat update) -> (mean, rstd) float32 [C]; two launches"""
    rows, Cc = x2d.shape
    sums = colstats(x2d)
    mean = torch.empty(Cc, dtype=torch.float32, device=x2d.device)
    rstd = torch.empty_like(mean)
    lib = _lib.load()
    _lib.check(lib.ctts_bn_finalize(_p(sums), rows, Cc, float(eps), float(momentum), _p(mean), _p(rstd), _p(running_mean),
                                    _p(running_var), _p(num_batches), _stream()), "ctts_bn_finalize")
    return mean, rstd


def bn_apply(x2d, mean, rstd, gamma, beta, act, p_drop=0.0, seed=None, drop_offset=0, want_planes=False):
    """-> y, or (y, planes of y) with want_planes (None when the shape has no plane layout)"""
    rows, Cc = x2d.shape
    y = torch.empty_like(x2d)
    pl = (new_planes(rows, Cc, x2d.device) if (want_planes and PRODUCER_PLANES and Cc % 32 == 0 and x2d.is_cuda and rows > 0
                                               and _al16(x2d, mean, rstd, gamma, beta)) else None)
    lib = _lib.load()
    _lib.check(lib.ctts_bn_apply(_p(_f32c(x2d, "x")), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(y), rows, Cc, act, p_drop,
                                 _p(seed), drop_offset, _p(pl), _stream()), "ctts_bn_apply")
    return (y, pl) if want_planes else y


def bn_bwd(dy, x2d, mean, rstd, gamma, beta, act, p_drop, seed, drop_offset, batch_stats, acc_into=None, want_planes=False):
    """acc_into = (dgamma, dbeta): existing float32 [C] buffers the parameter gradients are ADDED to (param.grad); returns (dx, None, None).
    want_planes: the bf16 plane set of dx is written by the apply launch and attached to dx (`planes_of(dx)`)."""
    rows, Cc = x2d.shape
    pl = (new_planes(rows, Cc, x2d.device) if (want_planes and PRODUCER_PLANES and Cc % 32 == 0 and x2d.is_cuda and rows > 0
                                               and _al16(dy, x2d, mean, rstd, gamma, beta)) else None)
    sums = torch.empty(2 * Cc, dtype=torch.float64, device=x2d.device)
    dx = torch.empty_like(x2d)
    if acc_into is not None:
        dgamma, dbeta = _f32c(acc_into[0], "dgamma"), _f32c(acc_into[1], "dbeta")
        batch_stats = int(batch_stats) | 2
    else:
        dgamma = torch.empty(Cc, dtype=torch.float32, device=x2d.device)
        dbeta = torch.empty(Cc, dtype=torch.float32, device=x2d.device)
    lib = _lib.load()
    _lib.check(lib.ctts_bn_bwd_reduce(_p(_f32c(dy, "dy")), _p(x2d), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(sums), rows,
                                      Cc, act, p_drop, _p(seed), drop_offset, _ws(dy), _stream()), "ctts_bn_bwd_reduce")
    _lib.check(lib.ctts_bn_bwd_apply(_p(dy), _p(x2d), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(sums), _p(dx), _p(dgamma),
                                     _p(dbeta), rows, Cc, act, p_drop, _p(seed), drop_offset, int(batch_stats), _p(pl), _stream()),
               "ctts_bn_bwd_apply")
    if pl is not None:
        attach_planes(dx, pl)
    return (dx, None, None) if acc_into is not None else (dx, dgamma, dbeta)


def softmax_fwd(S, lens, nb0, nb1, T):
    lib = _lib.load()
    _lib.check(lib.ctts_softmax_fwd(_p(S), _p(lens), nb0, nb1, T, T, _stream()), "ctts_softmax_fwd")
    return S


def softmax_bwd(P, dP, lens, nb0, nb1, T):
    lib = _lib.load()
    _lib.check(lib.ctts_softmax_bwd(_p(P), _p(dP), _p(lens), nb0, nb1, T, T, _stream()), "ctts_softmax_bwd")
    return dP


def act_dropout_bwd(dg, z, act, p_drop=0.0, seed=None, drop_offset=0):
    Cc = z.shape[-1]
    rows = z.numel() // Cc
    dz = torch.empty_like(z)
    lib = _lib.load()
    _lib.check(lib.ctts_act_dropout_bwd(_p(_f32c(dg, "dg")), _p(z), _p(dz), rows, Cc, act, 1.0, p_drop, _p(seed),
                                        drop_offset, _stream()), "ctts_act_dropout_bwd")
    return dz


def rowscale_dropout(x, rowscale=None, p_drop=0.0, seed=None, drop_offset=0):
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    y = torch.empty_like(x)
    lib = _lib.load()
    _lib.check(lib.ctts_rowscale_dropout(_p(_f32c(x, "x")), _p(y), rows, Cc, _p(rowscale), p_drop, _p(seed), drop_offset,
                                         _stream()), "ctts_rowscale_dropout")
    return y


def posembed_fwd(x, pos, table, alpha=None, rowscale=None, p_drop=0.0, seed=None, drop_offset=0):
    """y = rowscale * dropout(x + alpha * table[pos]); x [.., C] float32, pos int32 [rows], table [n_pos, C], alpha device scalar or None (= 1)"""
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    y = torch.empty_like(x)
    _lib.check(_lib.load().ctts_posembed_fwd(_p(_f32c(x, "x")), _p(pos), _p(_f32c(table, "table")), _p(alpha), _p(rowscale), _p(y), rows, Cc,
                                             float(p_drop), _p(seed), int(drop_offset), _stream()), "ctts_posembed_fwd")
    return y


def posembed_bwd(dy, pos, table, rowscale=None, p_drop=0.0, seed=None, drop_offset=0, want_alpha=True, alpha_acc_into=None):
    """-> (dx, dalpha [1] or None); alpha_acc_into: ADD the alpha gradient into that [1] buffer (param.grad)"""
    Cc = dy.shape[-1]
    rows = dy.numel() // Cc
    dx = torch.empty_like(dy)
    dalpha = None
    if want_alpha:
        dalpha = alpha_acc_into if alpha_acc_into is not None else torch.empty(1, dtype=torch.float32, device=dy.device)
    _lib.check(_lib.load().ctts_posembed_bwd(_p(_f32c(dy, "dy")), _p(pos), _p(table), _p(rowscale), _p(dx), _p(dalpha), rows, Cc, float(p_drop),
                                             _p(seed), int(drop_offset), int(alpha_acc_into is not None), _ws(dy) if want_alpha else None,
                                             _stream()), "ctts_posembed_bwd")
    return dx, dalpha


def colsum(x2d, ld=None, scale=1.0, acc_into=None):
    """out[c] = scale * sum_r x[r,c]; acc_into: add into an existing buffer (param.grad) instead."""
    rows, Cc = x2d.shape
    out = torch.empty(Cc, dtype=torch.float32, device=x2d.device) if acc_into is None else acc_into
    lib = _lib.load()
    sink = _sink_for(acc_into)
    nparts = lib.ctts_reduce_parts(0, rows, Cc) if sink is not None else 0
    if nparts > 1:
        parts = torch.empty(nparts, Cc, dtype=torch.float32, device=x2d.device)
        _lib.check(lib.ctts_colsum(_p(x2d), None, rows, Cc, ld if ld is not None else Cc, 1.0, 1, None, _p(parts), _stream()), "ctts_colsum")
        sink.add(parts, nparts, Cc, Cc, out, scale)
        return out
    ldv = ld if ld is not None else Cc
    CH = 65536                                   # the kernel's column-block limit (1024 blocks of 64): wider matrices go in column chunks
    for c0 in range(0, Cc, CH):
        _lib.check(lib.ctts_colsum(_p(x2d, c0), _p(out, c0), rows, min(CH, Cc - c0), ldv, float(scale), int(acc_into is not None),
                                   _ws(x2d), None, _stream()), "ctts_colsum")
    return out


def reflect_pad(y, pad):
    """-> [B, ld] with ld = N + 2*pad rounded up to a multiple of 4 (16-B aligned rows), tail zero."""
    B, N = y.shape
    ld = (N + 2 * pad + 3) // 4 * 4
    out = torch.empty(B, ld, dtype=torch.float32, device=y.device)
    lib = _lib.load()
    _lib.check(lib.ctts_reflect_pad(_p(_f32c(y, "y")), _p(out), B, N, pad, ld, _stream()), "ctts_reflect_pad")
    return out


def stft_magnitude(reim, frames, nbins, ld_mag):
    mag = torch.empty(frames, ld_mag, dtype=torch.float32, device=reim.device)
    energy = torch.empty(frames, dtype=torch.float32, device=reim.device)
    lib = _lib.load()
    _lib.check(lib.ctts_stft_magnitude(_p(reim), reim.shape[-1], _p(mag), ld_mag, _p(energy), frames, nbins, _stream()),
               "ctts_stft_magnitude")
    return mag, energy


def log_clamp_transpose(mel_fm, B, F, n_mel, clip):
    out = torch.empty(B, n_mel, F, dtype=torch.float32, device=mel_fm.device)
    lib = _lib.load()
    _lib.check(lib.ctts_log_clamp_transpose(_p(mel_fm), _p(out), B, F, n_mel, clip, _stream()), "ctts_log_clamp_transpose")
    return out


# ------------------------------------------------------------------------- conformer kernels
def glu_fwd(a):
    C2 = a.shape[-1]
    rows = a.numel() // C2
    out = torch.empty(*a.shape[:-1], C2 // 2, dtype=torch.float32, device=a.device)
    lib = _lib.load()
    _lib.check(lib.ctts_glu_fwd(_p(_f32c(a, "a")), _p(out), rows, C2 // 2, _stream()), "ctts_glu_fwd")
    return out


def glu_bwd(a, dout):
    C2 = a.shape[-1]
    rows = a.numel() // C2
    da = torch.empty_like(a)
    lib = _lib.load()
    _lib.check(lib.ctts_glu_bwd(_p(a), _p(_f32c(dout, "dout")), _p(da), rows, C2 // 2, _stream()), "ctts_glu_bwd")
    return da


def dwconv_fwd(x, wT, flip=False):
    """x [B,T,C], wT [K,C] -> y [B,T,C] (depthwise 'same' conv; flip=True is the data gradient)."""
    B, T, Cc = x.shape
    Kk = wT.shape[0]
    y = torch.empty_like(x)
    lib = _lib.load()
    _lib.check(lib.ctts_dwconv_fwd(_p(_f32c(x, "x")), _p(_f32c(wT, "wT")), _p(y), B, T, Cc, Kk, int(flip), _stream()),
               "ctts_dwconv_fwd")
    return y


def dwconv_wgrad(dy, x, Kk, acc_into=None):
    """acc_into: existing contiguous float32 buffer of C * K elements ([C,1,K] = nn.Conv1d groups=C layout) the gradient is ADDED to"""
    B, T, Cc = x.shape
    dw = _f32c(acc_into, "dw") if acc_into is not None else torch.empty(Cc, Kk, dtype=torch.float32, device=x.device)
    partials = torch.empty(128 * Cc * 32, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    _lib.check(lib.ctts_dwconv_wgrad(_p(_f32c(dy, "dy")), _p(x), _p(dw), _p(partials), B, T, Cc, Kk, int(acc_into is not None), _stream()),
               "ctts_dwconv_wgrad")
    return None if acc_into is not None else dw


def relpos_softmax_fwd(S, PS, T, scale, p_drop=0.0, seed=None, drop_offset=0, want_dropped=True):
    nb = S.numel() // (T * T)
    Pd = torch.empty_like(S) if want_dropped else None
    lib = _lib.load()
    _lib.check(lib.ctts_relpos_softmax_fwd(_p(S), _p(PS), _p(Pd), nb, T, scale, p_drop, _p(seed), drop_offset, _stream()),
               "ctts_relpos_softmax_fwd")
    return Pd


def relpos_softmax_bwd(P, dPd, T, scale, p_drop=0.0, seed=None, drop_offset=0):
    nb = P.numel() // (T * T)
    lib = _lib.load()
    _lib.check(lib.ctts_relpos_softmax_bwd(_p(P), _p(dPd), nb, T, scale, p_drop, _p(seed), drop_offset, _stream()),
               "ctts_relpos_softmax_bwd")
    return dPd


def relshift_bwd(dS, T):
    nb = dS.numel() // (T * T)
    dPS = torch.empty_like(dS)
    lib = _lib.load()
    _lib.check(lib.ctts_relshift_bwd(_p(dS), _p(dPS), nb, T, _stream()), "ctts_relshift_bwd")
    return dPS


# ------------------------------------------------------------------------- unsupervised alignment kernels
def neg_sqdist(q, k, temp):
    """q [B,Tq,C], k [B,Tk,C] -> [B,Tq,Tk] = -temp * ||q_t - k_s||^2"""
    B, Tq, Cc = q.shape
    Tk = k.shape[1]
    out = torch.empty(B, Tq, Tk, dtype=torch.float32, device=q.device)
    lib = _lib.load()
    _lib.check(lib.ctts_neg_sqdist(_p(_f32c(q, "q")), _p(_f32c(k, "k")), _p(out), B, Tq, Tk, Cc, float(temp), _stream()),
               "ctts_neg_sqdist")
    return out


def mas(attn, in_lens, out_lens):
    """attn [B,Tq,Tk] soft attention -> (hard [B,Tq,Tk] 0/1 float, dur [B,Tk] float)"""
    B, Tq, Tk = attn.shape
    opt = torch.empty_like(attn)
    dur = torch.empty(B, Tk, dtype=torch.float32, device=attn.device)
    back = torch.empty(B, Tq, Tk, dtype=torch.uint8, device=attn.device)
    lib = _lib.load()
    # keep both converted tensors alive across the launch: a temporary freed between the two conversions would hand the same
    # allocator block to the second one and the two pointers would alias
    in32 = in_lens.to(torch.int32).contiguous()
    out32 = out_lens.to(torch.int32).contiguous()
    _lib.check(lib.ctts_mas(_p(_f32c(attn, "attn")), _p(in32), _p(out32), _p(opt), _p(dur), _p(back), B, Tq, Tk, _stream()), "ctts_mas")
    return opt, dur


# ---- liu2021 prosody modelling (csrc/prosody.hip) -----------------------------------------------------------------------
def im2col_3x3s2(x):
    """x [B,T,W,C] channel-last -> col [B*T*Wo, 9*C] of Conv2d(3x3, stride (1,2), padding (1,1)); Wo = (W-1)//2+1"""
    B, T, W, Cc = x.shape
    Wo = (W - 1) // 2 + 1
    col = torch.empty(B * T * Wo, 9 * Cc, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    _lib.check(lib.ctts_im2col_3x3s2(_p(_f32c(x, "x")), _p(col), B, T, W, Cc, _stream()), "ctts_im2col_3x3s2")
    return col


def col2im_3x3s2(dcol, B, T, W, Cc):
    dx = torch.empty(B, T, W, Cc, dtype=torch.float32, device=dcol.device)
    lib = _lib.load()
    _lib.check(lib.ctts_col2im_3x3s2(_p(_f32c(dcol, "dcol")), _p(dx), B, T, W, Cc, _stream()), "ctts_col2im_3x3s2")
    return dx


def gru_fwd(gi, whh, bhh, H, ndir, save_gates=True, rev_mask=None):
    """gi [B,T,ndir*3H], whh [ndir,3H,H], bhh [ndir,3H] -> (out [B,T,ndir*H], gates [B,T,ndir*4H] or None)"""
    B, T = gi.shape[0], gi.shape[1]
    out = torch.empty(B, T, ndir * H, dtype=torch.float32, device=gi.device)
    gates = torch.empty(B, T, ndir * 4 * H, dtype=torch.float32, device=gi.device) if save_gates else None
    rev_mask = (2 if ndir == 2 else 0) if rev_mask is None else int(rev_mask)       # default: nn.GRU(bidirectional) = fwd | bwd
    lib = _lib.load()
    _lib.check(lib.ctts_gru_fwd(_p(_f32c(gi, "gi")), _p(_f32c(whh, "whh")), _p(_f32c(bhh, "bhh")), _p(out), _p(gates), B, T, H, ndir,
                                rev_mask, _stream()), "ctts_gru_fwd")
    return out, gates


def gru_bwd(dout, out, gates, whh, H, ndir, rev_mask=None):
    """-> (dgi, dgh [B,T,ndir*3H], hprev [B,T,ndir*H])"""
    B, T = out.shape[0], out.shape[1]
    dgi = torch.empty(B, T, ndir * 3 * H, dtype=torch.float32, device=out.device)
    dgh = torch.empty_like(dgi)
    hprev = torch.empty_like(out)
    rev_mask = (2 if ndir == 2 else 0) if rev_mask is None else int(rev_mask)
    lib = _lib.load()
    _lib.check(lib.ctts_gru_bwd(_p(_f32c(dout, "dout")), _p(out), _p(gates), _p(_f32c(whh, "whh")), _p(dgi), _p(dgh), _p(hprev), B, T, H,
                                ndir, rev_mask, _stream()), "ctts_gru_bwd")
    return dgi, dgh, hprev


def softmax_rect_fwd(S, klens, qlens):
    """in-place masked softmax over the last dim of S [nb,Tq,Tk]; klens/qlens int32 [nb] or None"""
    nb, Tq, Tk = S.shape
    lib = _lib.load()
    _lib.check(lib.ctts_softmax_rect_fwd(_p(_f32c(S, "S")), _p(klens), _p(qlens), nb, Tq, Tk, _stream()), "ctts_softmax_rect_fwd")
    return S


def softmax_rect_bwd(P, dP, klens, qlens):
    nb, Tq, Tk = P.shape
    lib = _lib.load()
    _lib.check(lib.ctts_softmax_rect_bwd(_p(P), _p(_f32c(dP, "dP")), _p(klens), _p(qlens), nb, Tq, Tk, _stream()),
               "ctts_softmax_rect_bwd")
    return dP


def forward_sum_fwd(attn_logprob, in32, out32, blank):
    """attn_logprob [B,Tq,Tk] -> (nll [B], lse [B,Tq], alpha [B,Tq,2Tk+1])"""
    B, Tq, Tk = attn_logprob.shape
    dev = attn_logprob.device
    lse = torch.empty(B, Tq, dtype=torch.float32, device=dev)
    alpha = torch.empty(B, Tq, 2 * Tk + 1, dtype=torch.float32, device=dev)
    nll = torch.empty(B, dtype=torch.float32, device=dev)
    lib = _lib.load()
    _lib.check(lib.ctts_forward_sum_fwd(_p(_f32c(attn_logprob, "attn_logprob")), _p(in32), _p(out32), float(blank), _p(lse), _p(alpha),
                                        _p(nll), B, Tq, Tk, _stream()), "ctts_forward_sum_fwd")
    return nll, lse, alpha


def forward_sum_bwd(attn_logprob, in32, out32, blank, lse, alpha, nll, gscale):
    B, Tq, Tk = attn_logprob.shape
    grad = torch.empty_like(attn_logprob)
    lib = _lib.load()
    _lib.check(lib.ctts_forward_sum_bwd(_p(attn_logprob), _p(in32), _p(out32), float(blank), _p(lse), _p(alpha), _p(nll),
                                        _p(_f32c(gscale, "gscale")), _p(grad), B, Tq, Tk, _stream()), "ctts_forward_sum_bwd")
    return grad


def adam_clip_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, max_norm, state):
    """fused clip_grad_norm_ + Adam on flat fp32 arenas (csrc/optim.hip); lr / state are device tensors (graph-replayable)"""
    if state.numel() < _lib.ADAM_STATE_FLOATS or state.dtype != torch.float32:
        raise _lib.CttsError(f"adam_clip_step: state must hold CTTS_ADAM_STATE_FLOATS = {_lib.ADAM_STATE_FLOATS} float32 "
                             f"(got {state.numel()} {state.dtype}): the kernel writes its per-block norm partials behind the 3 scalars")
    if not (p.numel() == g.numel() == m.numel() == v.numel()):
        raise _lib.CttsError("adam_clip_step: p, g, m, v must have the same number of elements")
    lib = _lib.load()
    _lib.check(lib.ctts_adam_clip_step(_p(_f32c(p, "p")), _p(_f32c(g, "g")), _p(_f32c(m, "m")), _p(_f32c(v, "v")), p.numel(), _p(lr),
                                       float(beta1), float(beta2), float(eps), float(weight_decay), float(max_norm), _p(state),
                                       _stream()), "ctts_adam_clip_step")


def embedding_fwd(ids, weight):
    """ids int64 [...], weight [V,C] -> [..., C]"""
    V, Cc = weight.shape
    ids = ids.contiguous()
    out = torch.empty(*ids.shape, Cc, dtype=torch.float32, device=weight.device)
    lib = _lib.load()
    _lib.check(lib.ctts_embedding_fwd(_p(ids), _p(_f32c(weight, "weight")), _p(out), ids.numel(), Cc, V, _stream()), "ctts_embedding_fwd")
    return out


def embedding_bwd(ids, dy, V, padding_idx=-1, acc_into=None):
    Cc = dy.shape[-1]
    dw = torch.empty(V, Cc, dtype=torch.float32, device=dy.device) if acc_into is None else acc_into
    lib = _lib.load()
    _lib.check(lib.ctts_embedding_bwd(_p(ids), _p(_f32c(dy, "dy")), _p(dw), ids.numel(), Cc, V, int(padding_idx),
                                      int(acc_into is not None), _ws(dy), _stream()), "ctts_embedding_bwd")
    return dw


def epilogue_bwd(dy, rowscale=None, z=None, act=0, p_drop=0.0, seed=None, drop_offset=0, want_gm=False, want_bias=False,
                 bias_scale=1.0, bias_acc_into=None):
    """one-pass backward of the GEMM epilogue -> (dZ, gm or None, dbias or None); see include/ctts.h ctts_epilogue_bwd"""
    Cc = dy.shape[-1]
    rows = dy.numel() // Cc
    dz = torch.empty_like(dy)
    gm = torch.empty_like(dy) if want_gm else None
    dbias = None
    if want_bias:
        dbias = bias_acc_into if bias_acc_into is not None else torch.empty(Cc, dtype=torch.float32, device=dy.device)
    lib = _lib.load()
    sink = _sink_for(bias_acc_into) if want_bias else None
    nparts = lib.ctts_reduce_parts(2, rows, Cc) if sink is not None else 0
    if nparts > 1:              # deferred bias gradient: per-stripe partial rows, added at the sink's flush
        parts = torch.empty(nparts, Cc, dtype=torch.float32, device=dy.device)
        _lib.check(lib.ctts_epilogue_bwd(_p(_f32c(dy, "dy")), _p(rowscale), _p(z if act else None), _p(dz), _p(gm), None, rows, Cc,
                                         int(act), float(p_drop), _p(seed), int(drop_offset), 1.0, 1, None, _p(parts), _stream()),
                   "ctts_epilogue_bwd")
        sink.add(parts, nparts, Cc, Cc, dbias, bias_scale)
        return dz, gm, dbias
    _lib.check(lib.ctts_epilogue_bwd(_p(_f32c(dy, "dy")), _p(rowscale), _p(z if act else None), _p(dz), _p(gm), _p(dbias), rows, Cc,
                                     int(act), float(p_drop), _p(seed), int(drop_offset), float(bias_scale),
                                     int(bias_acc_into is not None), _ws(dy) if want_bias else None, None, _stream()), "ctts_epilogue_bwd")
    return dz, gm, dbias


def rowdot_heads(a, b, n_heads):
    """a, b [B,T,C] -> [B,H,T]: per-head dot products along the channel slices (D = rowsum(dO * O) of the fused softmax backward)"""
    B, T, Cc = a.shape
    out = torch.empty(B, n_heads, T, dtype=torch.float32, device=a.device)
    lib = _lib.load()
    _lib.check(lib.ctts_rowdot_heads(_p(_f32c(a, "a")), _p(_f32c(b, "b")), _p(out), B, T, n_heads, Cc // n_heads, _stream()),
               "ctts_rowdot_heads")
    return out


# ---- fused attention (csrc/attn.hip) -------------------------------------------------------------------------------------
def mha_supported(C, H):
    return bool(_lib.load().ctts_mha_supported(int(C), int(H)))


def mha_fwd(qkv, lens, n_heads, scale):
    """qkv [B,T,3C] packed projections, lens int32 [B] or None -> (out [B,T,C], lse [B,H,T])"""
    B, T, C3 = qkv.shape
    Cc = C3 // 3
    out = torch.empty(B, T, Cc, dtype=torch.float32, device=qkv.device)
    lse = torch.empty(B, n_heads, T, dtype=torch.float32, device=qkv.device)
    lib = _lib.load()
    _lib.check(lib.ctts_mha_fwd(_p(_f32c(qkv, "qkv")), _p(lens), _p(out), _p(lse), B, T, n_heads, Cc, float(scale), _stream()), "ctts_mha_fwd")
    return out, lse


def mha_bwd(qkv, lens, out, dout, lse, n_heads, scale, q_split=1):
    """-> dqkv [B,T,3C]"""
    B, T, C3 = qkv.shape
    Cc = C3 // 3
    dev = qkv.device
    dqkv = torch.empty_like(qkv)
    Dws = torch.empty(B, n_heads, T, dtype=torch.float32, device=dev)
    dS = torch.empty(B, n_heads, T, T, dtype=torch.float32, device=dev)
    part = torch.empty(q_split, B, T, 2 * Cc, dtype=torch.float32, device=dev) if q_split > 1 else None
    lib = _lib.load()
    _lib.check(lib.ctts_mha_bwd(_p(qkv), _p(lens), _p(_f32c(out, "out")), _p(_f32c(dout, "dout")), _p(lse), _p(Dws), _p(dS), _p(part), _p(dqkv),
                                B, T, n_heads, Cc, float(scale), int(q_split), _ws(qkv), _stream()), "ctts_mha_bwd")
    return dqkv


def relattn_split_fwd(qkv, u_bias, v_bias):
    """packed q | k | v projection [..., 3C] -> (qu = q + u_bias, qv = q + v_bias [..., C], kv [..., 2C])"""
    Cc = qkv.shape[-1] // 3
    rows = qkv.numel() // (3 * Cc)
    qu = torch.empty(*qkv.shape[:-1], Cc, dtype=torch.float32, device=qkv.device)
    qv = torch.empty_like(qu)
    kv = torch.empty(*qkv.shape[:-1], 2 * Cc, dtype=torch.float32, device=qkv.device)
    _lib.check(_lib.load().ctts_relattn_split_fwd(_p(_f32c(qkv, "qkv")), _p(_f32c(u_bias, "u_bias")), _p(_f32c(v_bias, "v_bias")), _p(qu), _p(qv),
                                                  _p(kv), rows, Cc, _stream()), "ctts_relattn_split_fwd")
    return qu, qv, kv


def relattn_split_bwd(dqu, dqv, dkv):
    """-> dqkv [..., 3C] = (dqu + dqv) | dkv"""
    Cc = dqu.shape[-1]
    rows = dqu.numel() // Cc
    dqkv = torch.empty(*dqu.shape[:-1], 3 * Cc, dtype=torch.float32, device=dqu.device)
    _lib.check(_lib.load().ctts_relattn_split_bwd(_p(_f32c(dqu, "dqu")), _p(_f32c(dqv, "dqv")), _p(_f32c(dkv, "dkv")), _p(dqkv), rows, Cc,
                                                  _stream()), "ctts_relattn_split_bwd")
    return dqkv


def relmha_fwd(qu, qv, kv, pos, n_heads, scale, p_drop=0.0, seed=None, drop_offset=0):
    """-> (out [B,T,C], lse [B,H,T])"""
    B, T, Cc = qu.shape
    dev = qu.device
    lib = _lib.load()
    out = torch.empty(B, T, Cc, dtype=torch.float32, device=dev)
    lse = torch.empty(B, n_heads, T, dtype=torch.float32, device=dev)
    _lib.check(lib.ctts_relmha_fwd(_p(_f32c(qu, "qu")), _p(_f32c(qv, "qv")), _p(_f32c(kv, "kv")), _p(_f32c(pos, "pos")), _p(out),
                                   _p(lse), B, T, n_heads, Cc, float(scale), float(p_drop), _p(seed), int(drop_offset), _stream()),
               "ctts_relmha_fwd")
    return out, lse


def relmha_bwd(qu, qv, kv, pos, out, dout, lse, n_heads, scale, p_drop=0.0, seed=None, drop_offset=0):
    """-> (dqu, dqv [B,T,C], dkv [B,T,2C], dpos_b [B,T,C])"""
    B, T, Cc = qu.shape
    dev = qu.device
    lib = _lib.load()
    Dws = torch.empty(B, n_heads, T, dtype=torch.float32, device=dev)
    dS = torch.empty(lib.ctts_relmha_workspace_floats(B, T, n_heads), dtype=torch.float32, device=dev)
    dqu, dqv, dkv = torch.empty_like(qu), torch.empty_like(qv), torch.empty_like(kv)
    dpos_b = torch.empty(B, T, Cc, dtype=torch.float32, device=dev)
    _lib.check(lib.ctts_relmha_bwd(_p(qu), _p(qv), _p(kv), _p(pos), _p(_f32c(out, "out")), _p(_f32c(dout, "dout")), _p(lse), _p(Dws),
                                   _p(dS), _p(dqu), _p(dqv), _p(dkv), _p(dpos_b), B, T, n_heads, Cc, float(scale), float(p_drop),
                                   _p(seed), int(drop_offset), _stream()), "ctts_relmha_bwd")
    return dqu, dqv, dkv, dpos_b


def weighted_colsum(x2d, w, scale=1.0, acc_into=None):
    """out[c] = scale * sum_r w[r] * x[r,c]"""
    rows, Cc = x2d.shape
    out = torch.empty(Cc, dtype=torch.float32, device=x2d.device) if acc_into is None else acc_into
    lib = _lib.load()
    sink = _sink_for(acc_into)
    nparts = lib.ctts_reduce_parts(1, rows, Cc) if sink is not None else 0
    if nparts > 1:
        parts = torch.empty(nparts, Cc, dtype=torch.float32, device=x2d.device)
        _lib.check(lib.ctts_weighted_colsum(_p(_f32c(x2d, "x")), _p(_f32c(w, "w")), None, rows, Cc, 1.0, 1, None, _p(parts), _stream()),
                   "ctts_weighted_colsum")
        sink.add(parts, nparts, Cc, Cc, out, scale)
        return out
    _lib.check(lib.ctts_weighted_colsum(_p(_f32c(x2d, "x")), _p(_f32c(w, "w")), _p(out), rows, Cc, float(scale),
                                        int(acc_into is not None), _ws(x2d), None, _stream()), "ctts_weighted_colsum")
    return out


def mel_l1_fwd(p1, p2, tgt, pad_u8):
    """-> (sums [3] = {sum w|p1-t|, sum w|p2-t|, sum w}, roww [rows])"""
    Cc = tgt.shape[-1]
    rows = tgt.numel() // Cc
    sums = torch.empty(3, dtype=torch.float32, device=tgt.device)        # written by the workgroup that finishes the ordered sum
    roww = torch.empty(rows, dtype=torch.float32, device=tgt.device)
    lib = _lib.load()
    _lib.check(lib.ctts_mel_l1_fwd(_p(_f32c(p1, "p1")), _p(_f32c(p2, "p2")), _p(_f32c(tgt, "tgt")), _p(pad_u8), _p(sums), _p(roww), rows,
                                   Cc, _ws(tgt), _stream()), "ctts_mel_l1_fwd")
    return sums, roww


def mel_l1_bwd(p1, p2, tgt, roww, sums, g):
    Cc = tgt.shape[-1]
    rows = tgt.numel() // Cc
    d1, d2 = torch.empty_like(p1), torch.empty_like(p2)
    lib = _lib.load()
    _lib.check(lib.ctts_mel_l1_bwd(_p(p1), _p(p2), _p(tgt), _p(roww), _p(sums), _p(_f32c(g, "g")), _p(d1), _p(d2), rows, Cc, _stream()),
               "ctts_mel_l1_bwd")
    return d1, d2


# ---- fused variance / duration loss terms (csrc/loss.hip) ----------------------------------------------------------------
def _var_loss_args(log_d, dur, texts, src_pad, cwt, cwt_spec, uv, mel_pad, f0m_p, f0m_t, f0s_p, f0s_t, e_pred, e_tgt):
    B, Ts = log_d.shape
    Tm = cwt.shape[1]
    if texts.dtype != torch.int64:
        raise _lib.CttsError(f"var_loss: texts must be int64 token ids, got {texts.dtype}")
    if dur.dtype not in (torch.int64, torch.float32):
        raise _lib.CttsError(f"var_loss: durations must be int64 or float32, got {dur.dtype}")
    if cwt.shape[-1] != 11 or cwt_spec.shape[-1] != 10:
        raise _lib.CttsError("var_loss: expected cwt [B,Tm,11] (10 bins + uv logit) and cwt_spec [B,Tm,10]")
    ptrs = [_p(_f32c(log_d, "log_d")), _p(dur), int(dur.dtype == torch.float32), _p(texts), _p(src_pad), _p(_f32c(cwt, "cwt")),
            _p(_f32c(cwt_spec, "cwt_spec")), _p(_f32c(uv, "uv")), _p(mel_pad), _p(_f32c(f0m_p, "f0_mean")), _p(_f32c(f0m_t, "f0_mean tgt")),
            _p(_f32c(f0s_p, "f0_std")), _p(_f32c(f0s_t, "f0_std tgt")), _p(_f32c(e_pred, "e_pred")), _p(_f32c(e_tgt, "e_tgt")), B, Ts, Tm]
    return ptrs, B, Ts, Tm


def _check_var_loss_host_args(lambdas_t, sil_t):
    """the C ABI reads lambdas5[0..4] and sil_ids3[0..2] through HOST pointers at launch"""
    if lambdas_t.is_cuda or sil_t.is_cuda:
        raise _lib.CttsError("var_loss: lambdas / silence ids must be HOST tensors (the library reads them at launch); a .to(device) on "
                             "the loss module must not move them")
    if lambdas_t.dtype != torch.float32 or lambdas_t.numel() != 5 or not lambdas_t.is_contiguous():
        raise _lib.CttsError(f"var_loss: lambdas must be 5 contiguous float32, got {lambdas_t.numel()} {lambdas_t.dtype}")
    if sil_t.dtype != torch.int64 or sil_t.numel() != 3 or not sil_t.is_contiguous():
        raise _lib.CttsError(f"var_loss: silence ids must be 3 contiguous int64 (loss.py:149), got {sil_t.numel()} {sil_t.dtype}")


def var_loss_fwd(tensors, lambdas_t, cwt_l2, sil_t):
    """tensors = (log_d, dur, texts, src_pad u8, cwt, cwt_spec, uv, mel_pad u8, f0m_p, f0m_t, f0s_p, f0s_t, e_pred, e_tgt);
    lambdas_t float32 [5] and sil_t int64 [3] are HOST tensors (read at launch).  -> (terms [8], partials, wsum, denoms)"""
    _check_var_loss_host_args(lambdas_t, sil_t)
    ptrs, B, Ts, Tm = _var_loss_args(*tensors)
    dev = tensors[0].device
    partials = torch.empty(B, 16, dtype=torch.float32, device=dev)
    wsum = torch.empty(B, 2, Ts + 1, dtype=torch.float32, device=dev)
    terms = torch.empty(8, dtype=torch.float32, device=dev)
    denoms = torch.empty(4, dtype=torch.float32, device=dev)
    lib = _lib.load()
    _lib.check(lib.ctts_var_loss_fwd(*ptrs, lambdas_t.data_ptr(), int(cwt_l2), sil_t.data_ptr(), _p(partials), _p(wsum), _p(terms),
                                     _p(denoms), _stream()), "ctts_var_loss_fwd")
    return terms, partials, wsum, denoms


def var_loss_bwd(tensors, lambdas_t, cwt_l2, sil_t, partials, wsum, denoms, g8):
    _check_var_loss_host_args(lambdas_t, sil_t)
    ptrs, B, Ts, Tm = _var_loss_args(*tensors)
    dev = tensors[0].device
    d_log_d = torch.empty(B, Ts, dtype=torch.float32, device=dev)
    d_e = torch.empty(B, Ts, dtype=torch.float32, device=dev)
    d_cwt = torch.empty(B, Tm, 11, dtype=torch.float32, device=dev)
    d_f0m = torch.empty(B, dtype=torch.float32, device=dev)
    d_f0s = torch.empty(B, dtype=torch.float32, device=dev)
    lib = _lib.load()
    _lib.check(lib.ctts_var_loss_bwd(*ptrs, lambdas_t.data_ptr(), int(cwt_l2), sil_t.data_ptr(), _p(partials), _p(wsum), _p(denoms),
                                     _p(_f32c(g8, "g8")), _p(d_log_d), _p(d_cwt), _p(d_f0m), _p(d_f0s), _p(d_e), _stream()),
               "ctts_var_loss_bwd")
    return d_log_d, d_cwt, d_f0m, d_f0s, d_e


def bin_loss_fwd(soft, hard):
    n = soft.numel()
    partials = torch.empty(1024, dtype=torch.float32, device=soft.device)
    out2 = torch.empty(2, dtype=torch.float32, device=soft.device)
    lib = _lib.load()
    _lib.check(lib.ctts_bin_loss_fwd(_p(_f32c(soft, "soft")), _p(_f32c(hard, "hard")), n, _p(partials), _p(out2), _stream()), "ctts_bin_loss_fwd")
    return out2


def bin_loss_bwd(soft, hard, out2, g):
    dsoft = torch.empty_like(soft)
    lib = _lib.load()
    _lib.check(lib.ctts_bin_loss_bwd(_p(soft), _p(hard), _p(out2), _p(_f32c(g, "g")), _p(dsoft), soft.numel(), _stream()), "ctts_bin_loss_bwd")
    return dsoft


MASKED_L1, MASKED_L2, MASKED_BCE = 0, 1, 2


def masked_loss_fwd(pred, target, weight, kind):
    n = pred.numel()
    partials = torch.empty(1024, dtype=torch.float32, device=pred.device)
    out2 = torch.empty(2, dtype=torch.float32, device=pred.device)
    lib = _lib.load()
    _lib.check(lib.ctts_masked_loss_fwd(_p(_f32c(pred, "pred")), _p(_f32c(target, "target")), _p(_f32c(weight, "weight")), n, int(kind),
                                        _p(partials), _p(out2), _stream()), "ctts_masked_loss_fwd")
    return out2


def masked_loss_bwd(pred, target, weight, out2, g, kind):
    dpred = torch.empty_like(pred)
    lib = _lib.load()
    _lib.check(lib.ctts_masked_loss_bwd(_p(pred), _p(target), _p(weight), _p(out2), _p(_f32c(g, "g")), _p(dpred), pred.numel(), int(kind),
                                        _stream()), "ctts_masked_loss_bwd")
    return dpred


# ---- mel front end as a real FFT (csrc/mel.hip) ------------------------------------------------------------------------------
def mel_prepare(mel_basis, n_fft):
    """-> workspace tensor for `mel_spectrogram_fft` (twiddles, transposed filterbank, per-tile bin ranges)"""
    n_mel = mel_basis.shape[0]
    lib = _lib.load()
    ws = torch.empty(lib.ctts_mel_spectrogram_workspace_bytes(int(n_fft), int(n_mel)) // 4, dtype=torch.float32, device=mel_basis.device)
    _lib.check(lib.ctts_mel_prepare(_p(_f32c(mel_basis, "mel_basis")), int(n_fft), int(n_mel), _p(ws), _stream()), "ctts_mel_prepare")
    return ws


def mel_spectrogram_fft(y, window, ws, n_fft, hop, n_mel, clip=1e-5, want_mag=False, kmax=0, lens=None, range_flag=None):
    """y [B,N] -> (mel [B,n_mel,F], energy [B,F], mag [B*F,516] or None) in one launch; range_flag: optional device int32 [1] the kernel
    raises when a sample leaves [-1, 1] (float bits of the largest |sample|)"""
    B, N = y.shape
    F = 1 + N // hop
    dev = y.device
    mel = torch.empty(B, n_mel, F, dtype=torch.float32, device=dev)
    energy = torch.empty(B, F, dtype=torch.float32, device=dev)
    mag = torch.empty(B * F, 516, dtype=torch.float32, device=dev) if want_mag else None
    lib = _lib.load()
    _lib.check(lib.ctts_mel_spectrogram(_p(_f32c(y, "y")), _p(lens), _p(_f32c(window, "window")), _p(ws), _p(mel), _p(energy), _p(mag), 516, B, N,
                                        int(n_fft), int(hop), int(n_mel), float(clip), int(kmax),
                                        None if range_flag is None else range_flag.data_ptr(),      # device or pinned host memory
                                        _stream()), "ctts_mel_spectrogram")
    return mel, energy, mag
