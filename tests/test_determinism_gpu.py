"""GPU: run-to-run BIT reproducibility (round 4).  The reference's train loop (train.py:102-125) is deterministic for a fixed seed; the
round-3 product was not - fp32 atomics (split-K weight gradients, LayerNorm / BatchNorm / bias column sums, embedding backward, the mel
L1 sums) added in arrival order, and two runs of the same step differed in the 7th digit and drifted apart under Adam
(profiles/r04_diag_determinism_*_before.txt).  Every cross-workgroup sum now goes through a workspace in a fixed order
(csrc/ctts_common.h; split-K: csrc/gemm.hip splitk_reduce_kernel).  These tests repeat each reduction several times from the same inputs -
with other work in between, so that the arrival order of the workgroups changes - and demand bit-identical results, plus agreement with
float64; the train step is repeated from the same seed (eager twice, hipGraph replay) and must reproduce losses AND the whole gradient
arena bit for bit."""
import pytest
import torch

import ctts_amd  # noqa: F401
from ctts_amd import kernels as K
from ctts_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _noise():
    """unrelated work that perturbs the timing of the next launch's workgroups"""
    a = torch.randn(1537, 911, device=DEV)
    (a @ a.t()).sum().item()


def _repeat_equal(fn, n=4, name=""):
    ref = [t.clone() for t in fn()]
    for i in range(n):
        if i & 1:
            _noise()
        out = fn()
        for a, b in zip(ref, out):
            assert torch.equal(a, b), f"{name}: run {i + 1} differs from run 0 by {(a.double() - b.double()).abs().max().item():.3e}"
    return ref


def _close(a, b, tol, name):
    err = (a.double().cpu() - b.double().cpu()).abs().max().item()
    scale = max(1.0, b.abs().max().item())
    assert err <= tol * scale, f"{name}: {err:.3e} vs tol {tol * scale:.3e}"


@pytest.mark.parametrize("M,N,Kd,split,layout", [
    (1024, 256, 16384, 32, "TN"),        # FFN linear weight gradient: 64 tiles x 32 splits
    (256, 256, 16000, 16, "TN"),         # under-filled -> two-group kernel with split-K
    (130, 70, 2048, 4, "TN"),            # ragged edges
    (768, 256, 2048, 8, "TN"),
    (2048, 256, 2304, 4, "NT"),          # split data gradient
])
def test_split_k_gemm_is_bit_reproducible_and_matches_fp64(M, N, Kd, split, layout):
    g = torch.Generator().manual_seed(M + N + Kd)
    if layout == "TN":
        A = (torch.rand(Kd, M, generator=g) - 0.5).to(DEV)
        B = (torch.rand(Kd, N, generator=g) - 0.5).to(DEV)
        ref = 0.5 * (A.double().t() @ B.double())
        args = (M, N, Kd, M, N, N + 2, False, False)
    else:
        A = (torch.rand(M, Kd, generator=g) - 0.5).to(DEV)
        B = (torch.rand(N, Kd, generator=g) - 0.5).to(DEV)
        ref = 0.5 * (A.double() @ B.double().t())
        args = (M, N, Kd, Kd, Kd, N + 2, True, True)
    base = torch.rand(M + 1, N + 2, generator=g).to(DEV)          # split_k > 1 means C += alpha A B

    def run():
        Cb = base.clone()
        K.gemm(A, B, Cb, *args, split_k=split, alpha=0.5)
        return [Cb]
    out = _repeat_equal(run, name=f"split-K {layout} {M}x{N}x{Kd}/{split}")[0]
    _close(out[:M, :N] - base[:M, :N], ref, 3e-6 * max(1, Kd / 16), "split-K vs fp64")
    assert torch.equal(out[M:], base[M:]) and torch.equal(out[:, N:], base[:, N:]), "written outside [:M, :N]"


@pytest.mark.parametrize("split,overwrite", [(2, False), (2, True), (1, True)])
def test_split_k_batched_attention_gradient_is_reproducible(split, overwrite):
    """dQ = dS K with per-batch length limits and split_k = 2 (ops.self_attention's backward).  split_overwrite: the reduce launch WRITES
    the whole [T, dh] block of every batch - zeros for the query rows beyond the batch's length - so the caller passes an uninitialised
    tensor (here: NaN) instead of a zero-filled one; unsplit (split 1, the default of ops.self_attention since round 4) the tile kernel
    writes those zeros itself."""
    nb, T, dh = 6, 640, 128
    g = torch.Generator().manual_seed(5)
    lens = torch.tensor([640, 512, 333, 64, 7, 600], dtype=torch.int32, device=DEV)
    dS = (torch.rand(nb, T, T, generator=g) - 0.5).to(DEV)
    Km = (torch.rand(nb, T, dh, generator=g) - 0.5).to(DEV)

    def run():
        dQ = torch.full((nb, T, dh), float("nan"), device=DEV) if overwrite else torch.zeros(nb, T, dh, device=DEV)
        K.gemm(dS, Km, dQ, T, dh, T, T, dh, dh, True, False, nb0=nb, nb1=1, sA=(T * T, 0), sB=(T * dh, 0), sC=(T * dh, 0), lens=lens,
               lim=(1, 0, 1), split_k=split, split_overwrite=overwrite)
        return [dQ]
    out = _repeat_equal(run, name="batched split-K")[0]
    assert torch.isfinite(out).all()
    for b in range(nb):
        L = int(lens[b])
        _close(out[b, :L], dS[b, :L, :L].double() @ Km[b, :L].double(), 2e-5, f"dQ b={b}")
        assert L == T or float(out[b, L:].abs().max()) == 0.0


def test_split_k_overwrite_conv_data_gradient_writes_padded_tiles_as_zero():
    """Conv1d data gradient with ragged rows, split_k = 3 + split_overwrite into a NaN-filled dX: valid rows vs float64 through autograd of
    F.conv1d, whole 64-row tiles of padding exactly zero, everything finite (ops._LinearConv.backward no longer zero-fills dX)."""
    import torch.nn.functional as F
    B, T, Cin, N, ks = 3, 256, 64, 128, 9
    pad = (ks - 1) // 2
    lens = torch.tensor([256, 100, 37], dtype=torch.int32, device=DEV)
    g = torch.Generator().manual_seed(9)
    w = ((torch.rand(N, Cin, ks, generator=g) - 0.5) * 0.2).to(DEV)
    mask = (torch.arange(T, device=DEV)[None, :] < lens[:, None]).float()[..., None]
    dZ = (torch.rand(B, T, N, generator=g) - 0.5).to(DEV) * mask
    wd = torch.empty(Cin, ks * N, device=DEV)
    K.conv_weight_repack(w.contiguous(), wd, N, Cin, ks, 1)
    M = B * T

    def run():
        dX = torch.full((B, T, Cin), float("nan"), device=DEV)
        K.gemm(dZ, wd, dX, M, Cin, ks * N, N, ks * N, Cin, True, True, conv=(T, pad, N), row_lens=lens, row_T=T, row_halo=pad,
               tile_map=K.row_tile_map(lens, T, pad, M), split_k=3, split_overwrite=True, use_sk=False)
        return [dX]
    dX = _repeat_equal(run, name="conv dgrad split overwrite")[0]
    assert torch.isfinite(dX).all()
    x = torch.zeros(B, Cin, T, dtype=torch.float64, device=DEV, requires_grad=True)
    F.conv1d(x, w.double(), padding=pad).backward(dZ.double().transpose(1, 2))
    ref = x.grad.transpose(1, 2)
    for b in range(B):
        L = int(lens[b])
        _close(dX[b, :L], ref[b, :L], 2e-5, f"conv dgrad b={b}")
        t0 = (L + pad + 63) // 64 * 64
        assert t0 >= T or float(dX[b, t0:].abs().max()) == 0.0


def test_reduction_tickets_return_to_zero():
    """every ordered reduction leaves its ticket words at zero (the workspace is zero-filled once, by the caller)"""
    x = torch.randn(16384, 1024, device=DEV)
    K.colsum(x); K.colstats(x); K.epilogue_bwd(x, want_bias=True)
    g = torch.ones(1024, device=DEV)
    y, mean, rstd = K.layernorm_fwd(x, g, g, 1e-5)
    K.layernorm_bwd(x, x, g, mean, rstd)
    torch.cuda.synchronize()
    ws = K.gemm_workspace(x.device)
    lo, hi = 16384, 16384 + 4 * 65536 + 4 * 65536 + 4 * 1024           # csrc/ctts_common.h: GEMM ticket area (unused), level-1 and level-2 tickets
    assert int(ws[lo:hi].view(torch.int32).abs().max()) == 0


@pytest.mark.parametrize("rows,C", [(16384, 256), (2048, 256), (16384, 1024), (999, 80), (16384, 1), (655360, 32), (37, 512)])
def test_column_sums_are_bit_reproducible(rows, C):
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.rand(rows, C, generator=g) - 0.5).to(DEV)
    w = (torch.rand(rows, generator=g) - 0.5).to(DEV)
    acc0 = torch.rand(C, generator=g).to(DEV)

    def run():
        a = K.colsum(x, scale=0.5)
        b = K.colsum(x, acc_into=acc0.clone())
        c = K.weighted_colsum(x, w)
        return [a, b, c]
    a, b, c = _repeat_equal(run, name=f"colsum {rows}x{C}")
    _close(a, 0.5 * x.double().sum(0), 2e-6 * max(1.0, rows ** 0.5), "colsum")
    _close(b, acc0.double() + x.double().sum(0), 2e-6 * max(1.0, rows ** 0.5), "colsum acc")
    _close(c, (w.double()[:, None] * x.double()).sum(0), 2e-6 * max(1.0, rows ** 0.5), "weighted colsum")


@pytest.mark.parametrize("rows,C,act", [(16384, 1024, 2), (2048, 256, 1), (300, 11, 0)])
def test_epilogue_backward_bias_gradient_is_bit_reproducible(rows, C, act):
    g = torch.Generator().manual_seed(rows * 3 + C)
    dy = (torch.rand(rows, C, generator=g) - 0.5).to(DEV)
    z = (torch.rand(rows, C, generator=g) - 0.5).to(DEV)
    rs = (torch.rand(rows, generator=g) > 0.2).float().to(DEV)

    def run():
        dz, gm, db = K.epilogue_bwd(dy, rowscale=rs, z=z if act else None, act=act, want_gm=True, want_bias=True)
        return [dz, gm, db]
    dz, gm, db = _repeat_equal(run, name="epilogue_bwd")
    _close(db, dz.double().sum(0), 2e-6 * rows ** 0.5, "dbias = colsum(dZ)")


@pytest.mark.parametrize("rows,C", [(16384, 256), (4000, 384), (16384, 1024), (50, 256)])
def test_layernorm_backward_parameter_sums_are_bit_reproducible(rows, C):
    g = torch.Generator().manual_seed(rows + 7 * C)
    x = torch.randn(rows, C, generator=g).to(DEV)
    dy = torch.randn(rows, C, generator=g).to(DEV)
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV)
    beta = torch.zeros(C, device=DEV)
    y, mean, rstd = K.layernorm_fwd(x, gamma, beta, 1e-5)

    def run():
        return list(K.layernorm_bwd(dy, x, gamma, mean, rstd))
    dx, dg, db = _repeat_equal(run, name="layernorm_bwd")[:3]
    xh = (x.double() - mean.double()[:, None]) * rstd.double()[:, None]
    _close(dg, (dy.double() * xh).sum(0), 3e-6 * rows ** 0.5, "dgamma")
    _close(db, dy.double().sum(0), 3e-6 * rows ** 0.5, "dbeta")


@pytest.mark.parametrize("rows,C", [(16384, 512), (16000, 256), (655360, 32), (16384, 80)])
def test_batchnorm_sums_are_bit_reproducible(rows, C):
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g).to(DEV)

    def run():
        return [K.colstats(x)]
    s = _repeat_equal(run, name="colstats")[0]
    _close(s[:C], x.double().sum(0), 1e-9 * rows, "sum x")
    _close(s[C:], (x.double() ** 2).sum(0), 1e-9 * rows, "sum x^2")


@pytest.mark.parametrize("V,C,n", [(361, 256, 16384), (256, 256, 16384), (360, 256, 2048), (12, 64, 100)])
def test_embedding_backward_is_bit_reproducible(V, C, n):
    g = torch.Generator().manual_seed(V + n)
    ids = torch.randint(0, V, (n,), generator=g)
    ids[::3] = 1                                       # a popular row (the unvoiced pitch bin)
    ids = ids.to(DEV)
    dy = torch.randn(n, C, generator=g).to(DEV)
    acc0 = torch.rand(V, C, generator=g).to(DEV)

    def run():
        return [K.embedding_bwd(ids, dy, V, padding_idx=0), K.embedding_bwd(ids, dy, V, padding_idx=0, acc_into=acc0.clone())]
    dw, dwa = _repeat_equal(run, name="embedding_bwd")
    ref = torch.zeros(V, C, dtype=torch.float64)
    ref.index_add_(0, ids.cpu(), dy.double().cpu())
    ref[0] = 0
    _close(dw, ref, 1e-5, "embedding_bwd")
    _close(dwa, acc0.double().cpu() + ref, 1e-5, "embedding_bwd accumulate")
    assert float(dw[0].abs().max()) == 0.0


def test_deferred_sums_through_a_partial_sink_match_fp64_and_are_reproducible():
    """kernels.PartialSink (what trainer.TrainStep installs around every backward stage): the producers only write per-workgroup partials,
    ONE ctts_partial_sums launch adds them - in index order - into the destinations.  Against float64, against the immediate mode
    (rounding only: the association differs), bit-reproducible, and nothing is added before flush()."""
    g = torch.Generator().manual_seed(11)
    rows, C, N = 16384, 256, 1024
    x = (torch.rand(rows, C, generator=g) - 0.5).to(DEV)
    dyb = (torch.rand(rows, N, generator=g) - 0.5).to(DEV)
    w = (torch.rand(rows, generator=g) - 0.5).to(DEV)
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV)
    y, mean, rstd = K.layernorm_fwd(x, gamma, torch.zeros(C, device=DEV), 1e-5)
    A = (torch.rand(rows, N, generator=g) - 0.5).to(DEV)             # dZ [rows, N]: weight gradient dW[N, C] = dZ^T x over the rows

    def run(deferred):
        outs = [torch.full((C,), 0.25, device=DEV), torch.full((N,), 0.25, device=DEV), torch.full((C,), 0.25, device=DEV),
                torch.full((C,), 0.25, device=DEV), torch.full((C,), 0.25, device=DEV), torch.full((N, C), 0.25, device=DEV)]
        sink = K.PartialSink()
        sink.FLUSH_BYTES = 0                             # this test looks at the sink before its end-of-stage flush
        prev = K.set_partial_sink(sink if deferred else None)
        try:
            K.colsum(x, scale=0.5, acc_into=outs[0])
            K.epilogue_bwd(dyb, want_bias=True, bias_scale=2.0, bias_acc_into=outs[1])
            K.weighted_colsum(x, w, scale=0.5, acc_into=outs[2])
            K.layernorm_bwd(x, x, gamma, mean, rstd, acc_into=(outs[3], outs[4]))
            K.gemm(A, x, outs[5], N, C, rows, N, C, C, False, False, split_k=31, alpha=0.5, defer=True)
            if deferred:
                assert len(sink.tasks) == 6
                torch.cuda.synchronize()
                assert all(float((o - 0.25).abs().max()) == 0.0 for o in outs), "a destination changed before flush()"
                sink.flush()
                assert not sink.tasks and not sink.keep
        finally:
            K.set_partial_sink(prev)
        return outs
    d = _repeat_equal(lambda: run(True), name="deferred sums")
    i = run(False)
    xh = (x.double() - mean.double()[:, None]) * rstd.double()[:, None]
    refs = [0.25 + 0.5 * x.double().sum(0), 0.25 + 2.0 * dyb.double().sum(0), 0.25 + 0.5 * (w.double()[:, None] * x.double()).sum(0),
            0.25 + (x.double() * xh).sum(0), 0.25 + x.double().sum(0), 0.25 + 0.5 * (A.double().t() @ x.double())]
    for k, (a, b, r) in enumerate(zip(d, i, refs)):
        tol = 3e-6 * rows ** 0.5 if k < 5 else 3e-6 * rows / 16
        _close(a, r, tol, f"deferred sum {k} vs fp64")
        _close(a, b, tol, f"deferred vs immediate {k}")


def _train(c5, block, use_graph, n=3, canonical=False):
    from ctts_amd.configs import get_configs
    from ctts_amd.loss import CompTransTTSLoss, ScheduledOptim
    from ctts_amd.synthetic import make_batch, make_unsup_batch, to_device, as_model_args
    from ctts_amd.trainer import TrainStep
    torch.manual_seed(1234)
    pre, mc, tc = get_configs()
    mc["block_type"] = block
    if c5:
        mc["prosody_modeling"]["model_type"] = "liu2021"
        mc["duration_modeling"]["learn_alignment"] = True
    model = ctts_amd.CompTransTTS(pre, mc, tc).to(DEV)
    model.train()
    loss_fn, optim = CompTransTTSLoss(pre, mc, tc).to(DEV), ScheduledOptim(model, tc, mc, 50000, capturable=True)
    cap = 1000 if block == "conformer" else None
    lens = None if canonical else [60, 41, 33, 17]
    batch = to_device((make_unsup_batch if c5 else make_batch)(lens, 8, seed=3, max_mel_cap=cap), DEV)
    step = TrainStep(model, loss_fn, optim, as_model_args(batch), world=1, use_graph=use_graph)
    if c5:
        step.step_no = 100001
    if use_graph:
        step.capture(warmup=2)
    else:
        for _ in range(2):
            step.optim.update_learning_rate()
            step._eager()
    losses, grads = [], None
    for i in range(n):
        step()
        losses.append(float(step.loss_val))
        if i == 0:
            torch.cuda.synchronize()
            grads = step.flat_grad.clone()
    torch.cuda.synchronize()
    return losses, grads, step.fadam.flat_param.clone()


@pytest.mark.parametrize("block,c5", [("transformer_fs2", False), ("transformer_fs2", True), ("conformer", False)])
def test_train_step_is_bit_reproducible_eager_and_graph(block, c5):
    """VERDICT r03 'Done' criterion: two eager runs from the same seed produce bit-identical losses - and so does hipGraph replay, and so
    do the whole gradient arena after the first step and the parameters after the last (fs2, C5 = liu2021 + learn_alignment, conformer)."""
    e1 = _train(c5, block, False)
    _noise()
    e2 = _train(c5, block, False)
    g1 = _train(c5, block, True)
    for tag, other in (("eager #2", e2), ("graph replay", g1)):
        assert e1[0] == other[0], f"{tag}: losses differ {e1[0]} vs {other[0]}"
        assert torch.equal(e1[1], other[1]), f"{tag}: gradient arena differs by {(e1[1] - other[1]).abs().max().item():.3e}"
        assert torch.equal(e1[2], other[2]), f"{tag}: parameters differ by {(e1[2] - other[2]).abs().max().item():.3e}"


def test_canonical_batch_train_step_is_bit_reproducible():
    """the B = 16 / T = 1024 batch bench.py times: every large launch (stream-K hand-off, 32-way split-K, 256-stripe column sums) in play"""
    a = _train(False, "transformer_fs2", False, n=2, canonical=True)
    _noise()
    b = _train(False, "transformer_fs2", False, n=2, canonical=True)
    assert a[0] == b[0], (a[0], b[0])
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])


def test_workspace_error_probe_reads_clean_and_raises_on_a_set_word():
    from ctts_amd._lib import CttsError
    x = torch.randn(4096, 256, device=DEV)
    K.colsum(x)
    probe = K.WorkspaceErrorProbe()
    probe.poll_and_check()                               # clean
    ws = K.gemm_workspace(x.device)
    ws.view(torch.int32)[2048] = 7
    with pytest.raises(CttsError, match="hand-off"):
        probe.poll_and_check()
    assert int(ws.view(torch.int32)[2048]) == 0          # re-zeroed
    probe.poll_and_check()


def test_grad_acc_step_follows_the_reference_loop():
    """train.py:112-125 with optimizer.grad_acc_step = k > 1 (VERDICT r03 missing #5; TrainStep used to ignore it): the loss is divided
    by k (so every gradient is EXACTLY the k = 1 gradient times 1/k - a power of two here), clipping happens only on steps with
    step % k == 0, the optimizer steps and the arena is zeroed on every step (the reference's `step_and_update_lr` / `zero_grad` sit
    outside its `if`: it never sums gradients over batches).  Eager and hipGraph replay (two optimizer graphs: with / without the clip
    coefficient) are bit-identical over four steps."""
    from ctts_amd.configs import get_configs
    from ctts_amd.loss import CompTransTTSLoss, ScheduledOptim
    from ctts_amd.synthetic import make_batch, to_device, as_model_args
    from ctts_amd.trainer import TrainStep

    def run(k, use_graph, n, warm):
        torch.manual_seed(1234)
        pre, mc, tc = get_configs()
        model = ctts_amd.CompTransTTS(pre, mc, tc).to(DEV)
        model.train()
        loss_fn, optim = CompTransTTSLoss(pre, mc, tc).to(DEV), ScheduledOptim(model, tc, mc, 50000, capturable=True)
        batch = to_device(make_batch([60, 41, 33, 17], 8, seed=3), DEV)
        step = TrainStep(model, loss_fn, optim, as_model_args(batch), world=1, use_graph=use_graph, grad_acc_step=k)
        if use_graph:
            step.capture(warmup=warm)        # `warm` REAL eager steps (lazy tables, workspaces), then the graphs
            assert (step.g_opt_noclip is not None) == (k > 1)
        else:
            for _ in range(warm):
                step.optim.update_learning_rate()
                step._eager()
        losses, clips, g1 = [], [], None
        for i in range(n):
            clips.append(step._clip_now())
            step()
            losses.append(float(step.loss_val))
            if i == 0:
                torch.cuda.synchronize()
                g1 = step.flat_grad.clone()
        torch.cuda.synchronize()
        return losses, g1, step.fadam.flat_param.clone(), clips
    l1, g1, _, c1 = run(1, False, 1, 0)
    l2, g2, p2, c2 = run(2, False, 4, 0)
    assert c1 == [True] and c2 == [False, True, False, True]           # step_no starts at 50001
    # scaling by a power of two commutes with every rounding (short of the subnormal range): the k = 2 gradients are the k = 1 gradients / 2
    assert l1[0] == l2[0] and torch.allclose(g2, 0.5 * g1, rtol=1e-6, atol=1e-20), "gradients of step 1 are not half the k = 1 gradients"
    le, ge, pe, _ = run(2, False, 4, 2)
    lg, gg, pg, _ = run(2, True, 4, 2)
    assert lg == le and torch.equal(gg, ge) and torch.equal(pg, pe), "hipGraph replay differs from eager with grad_acc_step = 2"
