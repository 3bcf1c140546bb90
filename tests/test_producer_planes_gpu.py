"""GPU: operand planes written by their PRODUCERS (round 6).  A plane set written by LayerNorm forward, BatchNorm
apply / backward or the producer-epilogue backward of the weight-stationary GEMM must be BIT-IDENTICAL to what ctts_split_planes makes
from the fp32 tensor the same launch wrote (one arithmetic: csrc/planes_common.h) - then the plane-kernel GEMMs that consume it give the
bits they gave before, and a train step with the producers on equals the step that splits every operand in a launch of its own."""
import numpy as np
import pytest
import torch

import ctts_amd
from ctts_amd import kernels as K
from ctts_amd import ops
from ctts_amd.configs import get_configs
from ctts_amd.synthetic import make_batch, to_device, as_model_args

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0") if torch.cuda.is_available() else None


def _same_planes(pl, t2d):
    ref = K.split_planes([t2d.contiguous()])[0]
    assert pl is not None and pl.shape == ref.shape
    assert torch.equal(pl.view(torch.int16), ref.view(torch.int16))


def _nasty(rows, C, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, C, generator=g) * torch.exp(torch.randn(rows, C, generator=g) * 3)
    return x.to(DEV)


@pytest.mark.parametrize("rows,C,p_drop,scaled", [(515, 256, 0.0, False), (2048, 256, 0.2, True), (77, 512, 0.1, True), (64, 1024, 0.0, False),
                                                   (33, 32, 0.0, True)])
def test_layernorm_forward_writes_the_plane_set_of_its_output(rows, C, p_drop, scaled):
    x = _nasty(rows, C, 1)
    gamma, beta = torch.randn(C, device=DEV), torch.randn(C, device=DEV)
    rs = (torch.rand(rows, device=DEV) > 0.3).float() if scaled else None
    seed = torch.full((1,), 1234, dtype=torch.int64, device=DEV) if p_drop > 0 else None
    y0, m0, r0 = K.layernorm_fwd(x, gamma, beta, 1e-5, p_drop, seed, 7, rs)
    y1, m1, r1 = K.layernorm_fwd(x, gamma, beta, 1e-5, p_drop, seed, 7, rs, want_planes=True)
    assert torch.equal(y0, y1) and torch.equal(m0, m1) and torch.equal(r0, r1)
    assert K.planes_of(y0) is None
    _same_planes(K.planes_of(y1), y1)
    # the attachment is void once the tensor was written (version) or for another tensor
    y2 = y1.clone()
    assert K.planes_of(y2) is None
    y1.add_(1.0)
    assert K.planes_of(y1) is None


def test_layernorm_planes_are_refused_without_the_layout():
    x = torch.randn(8, 48, device=DEV)               # C % 32 != 0: no plane layout - the wrapper does not ask, the C entry point refuses
    y, _, _ = K.layernorm_fwd(x, torch.ones(48, device=DEV), torch.zeros(48, device=DEV), 1e-5, want_planes=True)
    assert K.planes_of(y) is None
    from ctts_amd import _lib
    lib = _lib.load()
    pl = torch.empty(8 * 48 * 3, dtype=torch.bfloat16, device=DEV)
    m = torch.empty(8, device=DEV)
    rc = lib.ctts_layernorm_fwd(x.data_ptr(), y.data_ptr(), y.data_ptr(), y.data_ptr(), m.data_ptr(), m.data_ptr(), 8, 48, 1e-5, 0.0, None, 0,
                                None, pl.data_ptr(), None)
    assert rc != 0 and b"C % 32" in lib.ctts_last_error()


def test_batchnorm_wide_kernels_equal_the_scalar_kernels_element_for_element():
    """the 8-channels-per-thread kernels (every launch with C % 8 == 0 and 16-byte aligned operands since round 6) against the scalar
    kernels they replace, reached through a misaligned view (C % 8 != 0 is the other way in): same arithmetic, same bits"""
    rows, C = 777, 40
    x = _nasty(rows, C, 9)
    gamma, beta = torch.randn(C, device=DEV), torch.randn(C, device=DEV)
    mean, rstd = x.mean(0), torch.rsqrt(x.var(0, unbiased=False) + 1e-5)
    seed = torch.full((1,), 17, dtype=torch.int64, device=DEV)
    dy = _nasty(rows, C, 10)
    wide = (K.bn_apply(x, mean, rstd, gamma, beta, K.ACT_TANH, 0.5, seed, 3),) + tuple(K.bn_bwd(dy, x, mean, rstd, gamma, beta, K.ACT_TANH, 0.5, seed, 3, True))
    # the same operands 4 bytes off a 16-byte boundary: the library falls back to the scalar kernels
    def off(t):
        buf = torch.empty(t.numel() + 1, device=DEV)
        v = buf[1:].view(t.shape)
        v.copy_(t)
        return v
    g2, b2, m2, r2 = off(gamma), off(beta), off(mean), off(rstd)
    assert g2.data_ptr() % 16 != 0
    scalar = (K.bn_apply(x, m2, r2, g2, b2, K.ACT_TANH, 0.5, seed, 3),) + tuple(K.bn_bwd(dy, x, m2, r2, g2, b2, K.ACT_TANH, 0.5, seed, 3, True))
    for a, b in zip(wide, scalar):
        assert torch.equal(a, b)


@pytest.mark.parametrize("rows,C,act,p_drop", [(16384, 512, K.ACT_TANH, 0.5), (1000, 512, K.ACT_NONE, 0.0), (333, 64, K.ACT_TANH, 0.5)])
def test_batchnorm_apply_and_backward_write_plane_sets(rows, C, act, p_drop):
    x = _nasty(rows, C, 2)
    gamma, beta = torch.randn(C, device=DEV), torch.randn(C, device=DEV)
    mean, rstd = x.mean(0), torch.rsqrt(x.var(0, unbiased=False) + 1e-5)
    seed = torch.full((1,), 99, dtype=torch.int64, device=DEV) if p_drop > 0 else None
    y0 = K.bn_apply(x, mean, rstd, gamma, beta, act, p_drop, seed, 3)
    y1, pl = K.bn_apply(x, mean, rstd, gamma, beta, act, p_drop, seed, 3, want_planes=True)
    assert torch.equal(y0, y1)
    _same_planes(pl, y1)
    dy = _nasty(rows, C, 3)
    for batch_stats in (True, False):
        a = K.bn_bwd(dy, x, mean, rstd, gamma, beta, act, p_drop, seed, 3, batch_stats)
        b = K.bn_bwd(dy, x, mean, rstd, gamma, beta, act, p_drop, seed, 3, batch_stats, want_planes=True)
        for u, v in zip(a, b):
            assert torch.equal(u, v)
        assert K.planes_of(a[0]) is None
        _same_planes(K.planes_of(b[0]), b[0])


def test_weight_stationary_backward_epilogue_writes_the_planes_of_dz():
    """the EpiLink launch of the decoder FFN (ops._LinearConv.backward, link_role 2): dX = drop_mask / (1 - p) * gelu'(Z) * alpha * (dY W) on
    the K = 256 weight-stationary kernel, ragged rows with the device tile schedule - dX unchanged by C_planes, the plane set equal to
    ctts_split_planes(dX) INCLUDING the zero tiles of wholly padded rows, and ignored-by-contract on a launch the kernel does not take."""
    B, T, Kd, N = 8, 1024, 256, 1024
    M = B * T
    g = torch.Generator().manual_seed(5)
    lens = torch.tensor([1024, 700, 64, 0, 513, 1, 900, 128], dtype=torch.int32, device=DEV)
    valid = (torch.arange(T, device=DEV)[None, :] < lens[:, None]).reshape(M, 1).float()
    dY = torch.randn(M, Kd, generator=g).to(DEV) * valid
    W = (torch.randn(Kd, N, generator=g) * 0.05).to(DEV)
    Z = torch.randn(M, N, generator=g).to(DEV)
    seed = torch.full((1,), 4321, dtype=torch.int64, device=DEV)
    pr = ops.PadRows(lens, T)
    kw = dict(alpha=0.37, epi_bwd=True, Z=Z, ldz=N, act=K.ACT_GELU, p_drop=0.1, seed=seed, drop_offset=11, tile_map=pr.tile_map(0, M),
              row_lens=lens, row_T=T)
    d0 = torch.full((M, N), float("nan"), device=DEV)
    K.gemm(dY, W, d0, M, N, Kd, Kd, N, N, True, False, **kw)
    pl = K.new_planes(M, N, DEV)
    pl.view(torch.int16).fill_(0x7FC0)               # NaN patterns: every piece the launch owes must be overwritten
    assert K.gemm_takes_weight_stationary(dY, W, d0, M, N, Kd, Kd, N, N, True, False, c_planes=pl, **kw)
    d1 = torch.full((M, N), float("nan"), device=DEV)
    K.gemm(dY, W, d1, M, N, Kd, Kd, N, N, True, False, c_planes=pl, **kw)
    assert torch.isfinite(d1).all() and torch.equal(d0, d1)
    _same_planes(pl, d1)
    assert float(d1[3 * T:4 * T].abs().max()) == 0.0          # the empty utterance: zero tiles, zero planes
    # a forward epilogue never writes planes: the kernel does not take such a descriptor - callers must ask first
    fw = dict(alpha=1.0, bias=torch.zeros(N, device=DEV))
    assert not K.gemm_takes_weight_stationary(dY, W, d0, M, N, Kd, Kd, N, N, True, False, c_planes=pl, **fw)


def _run_steps(producers, block="transformer_fs2", lens=None, n=2):
    from ctts_amd.loss import CompTransTTSLoss, ScheduledOptim
    from ctts_amd.trainer import TrainStep
    prev = K.PRODUCER_PLANES
    K.PRODUCER_PLANES = producers
    try:
        torch.manual_seed(1234)
        pre, mc, tc = get_configs()
        mc["block_type"] = block
        model = ctts_amd.CompTransTTS(pre, mc, tc).to(DEV)
        model.train()
        loss_fn, optim = CompTransTTSLoss(pre, mc, tc).to(DEV), ScheduledOptim(model, tc, mc, 50000, capturable=True)
        batch = to_device(make_batch(lens, seed=1234, max_mel_cap=1000 if block == "conformer" else None), DEV)
        step = TrainStep(model, loss_fn, optim, as_model_args(batch), world=1, use_graph=False)
        losses = []
        for _ in range(n):
            step.optim.update_learning_rate()
            step._eager()
            losses.append(float(step.loss_val))
        torch.cuda.synchronize()
        return losses, step.fadam.flat_param.clone(), step
    finally:
        K.PRODUCER_PLANES = prev


@pytest.mark.parametrize("block", ["transformer_fs2", "conformer"])
def test_canonical_step_with_producer_planes_equals_the_step_that_splits_separately(block):
    """BASELINE configs[1] / [2] at full size, dropout on: two steps with the producers writing planes and two steps with
    CTTS_PRODUCER_PLANES=0 semantics give the same losses and the same parameters bit for bit (identical plane bits -> identical GEMMs)."""
    a = _run_steps(True, block)
    b = _run_steps(False, block)
    assert a[0] == b[0], (a[0], b[0])
    assert torch.equal(a[1], b[1])


def test_split_launches_per_canonical_fs2_step_are_down_to_the_weights_and_the_encoder():
    """ctts_split_planes launches per fs2 step (round 5: 29).  What remains (tools/dbg_splits.py lists them with call
    sites): ONE launch for all 24 weight sets of the step, the dZ of the four encoder FFN layers (their EpiLink GEMM has 2,048 rows - below
    the weight-stationary kernel's threshold, and only that kernel's epilogue writes planes) and the input / dZ of the one k = 5 predictor
    convolution on the 16,384 decoder rows (fed by an add, not by a LayerNorm): 7."""
    from torch.profiler import profile, ProfilerActivity
    _, _, step = _run_steps(True, n=2)
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        step.optim.update_learning_rate()
        step._eager()
        torch.cuda.synchronize()
    names = [e.name for e in prof.events() if getattr(e, "device_type", None) is not None and "cuda" in str(e.device_type).lower()]
    n_split = sum(1 for n in names if "split_planes_kernel" in n)
    print(f"split_planes_kernel launches per step: {n_split}; device launches {len(names)}")
    assert 1 <= n_split <= 8, n_split
