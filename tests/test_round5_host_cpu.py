"""CPU: host-side logic added in round 5 that needs no GPU - the plane layout contract of ctts_split_planes (numpy restatement), the
step-scoped weight caches, the deferred-sum overlap guard, gradient-accumulation fusion through a GEMM-major Conv2d weight, the plane
kernel's ISA invariants (hipcc cross-compiles here)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import ctts_amd  # noqa: F401
from ctts_amd import kernels as K
from ctts_amd import model as M
from ctts_amd import ops
from ctts_amd import prosody as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def split_planes_ref(x):
    """include/ctts.h ctts_split_planes restated with torch's round-to-nearest-even bf16 conversion: [rows, cols] fp32 ->
    [rows, cols / 32, 3, 32] bf16, piece q of element (r, c) at [r, c // 32, q, c % 32]; hi clamped to the largest bf16 when a finite
    value would round to infinity; non-finite values keep hi and get zero remainders."""
    x = x.float()
    hi = x.bfloat16()
    over = torch.isinf(hi.float()) & torch.isfinite(x)
    bf_max = torch.tensor(3.3895313892515355e38).bfloat16()
    hi = torch.where(over, torch.where(x > 0, bf_max, -bf_max), hi)
    fin = torch.isfinite(x)
    r1 = torch.where(fin, x - hi.float(), torch.zeros_like(x))
    mid = r1.bfloat16()
    lo = (r1 - mid.float()).bfloat16()
    rows, cols = x.shape
    return torch.stack([hi, mid, lo], 0).view(3, rows, cols // 32, 32).permute(1, 2, 0, 3).contiguous()


def test_plane_layout_contract_and_exactness_of_the_split():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(7, 96, generator=g) * torch.exp(torch.randn(7, 96, generator=g) * 5)
    x[0, :4] = torch.tensor([3.4028234e38, -3.4e38, float("inf"), float("nan")])
    pl = split_planes_ref(x)
    assert tuple(pl.shape) == (7, 3, 3, 32)
    pieces = [K.planes_piece(pl, q).float().double() for q in range(3)]
    fin = torch.isfinite(x)
    assert torch.equal((pieces[0] + pieces[1] + pieces[2])[fin], x.double()[fin])              # nothing dropped, FLT_MAX included
    assert torch.isfinite(pieces[0][0, :2]).all() and pieces[0][0, 2] == float("inf") and torch.isnan(pieces[0][0, 3])
    assert float(pieces[1][0, 2]) == 0.0 and float(pieces[2][0, 3]) == 0.0
    # the flat address of include/ctts.h: planes[row * 3 * ld + (k / 32) * 96 + q * 32 + k % 32]
    flat = pl.reshape(-1).view(torch.int16).numpy()
    r, k, q = 5, 77, 1
    assert flat[r * 3 * 96 + (k // 32) * 96 + q * 32 + k % 32] == K.planes_piece(pl, q)[r, k].view(torch.int16).item()
    # the pieces shrink by 2^-8 per level (round to nearest); the two clamped values carry a slightly larger remainder
    body = fin.clone()
    body[0, :4] = False
    assert float((pieces[1].abs() - pieces[0].abs() * 2.0 ** -8)[body].max()) <= 0
    assert float((pieces[2].abs() - pieces[0].abs() * 2.0 ** -16)[body].max()) <= 0


def test_plane_shape_prefilter_follows_the_kernel_rules():
    prev = K.gemm_bf16_split_enable(True)
    try:
        assert K.plane_shape_ok(16384, 1024, 2304, 256)            # decoder FFN conv forward
        assert K.plane_shape_ok(16384, 256, 9216, 1024)            # its data gradient
        assert not K.plane_shape_ok(16384, 80, 2560, 512)          # N < 256 (last PostNet layer)
        assert not K.plane_shape_ok(16384, 512, 400, 80)           # K not a multiple of 32, cin not a multiple of 32
        assert not K.plane_shape_ok(2048, 256, 768, 256)           # too few (tile, K-block) units
        K.gemm_bf16_split_enable(False)
        assert not K.plane_shape_ok(16384, 1024, 2304, 256)        # fp32-MFMA-only descriptors never take planes
    finally:
        K.gemm_bf16_split_enable(prev)


def test_dgrad_cache_hands_out_only_entries_made_from_the_current_weight_version():
    w = torch.nn.Parameter(torch.zeros(4, 3, 5))
    cache = ops._DgradCache()
    wd = torch.zeros(3, 20)
    # entries carry ops._wstamp(w) = (autograd version, epoch of raw-pointer updates: dp.FlatAdam) since round 6 - tests/test_round6_host_cpu.py
    cache[w.data_ptr()] = (wd, ops._wstamp(w))
    assert cache.take(w, (3, 20)) is wd and cache.take(w, (3, 20)) is None          # popped
    cache[w.data_ptr()] = (wd, ops._wstamp(w))
    with torch.no_grad():
        w.add_(1.0)                                                                  # an in-place update after the preparation
    assert cache.take(w, (3, 20)) is None
    cache[w.data_ptr()] = (wd, ops._wstamp(w))
    assert cache.take(w, (4, 20)) is None                                            # wrong shape


def test_partial_sink_flushes_before_a_second_sum_into_the_same_destination(monkeypatch):
    sink = K.PartialSink()
    flushed = []
    monkeypatch.setattr(K.PartialSink, "flush", lambda self: (flushed.append(len(self.tasks)), self.tasks.clear()))
    dst, other = torch.zeros(64), torch.zeros(64)
    src = torch.zeros(4, 64)
    sink.add(src, 4, 64, 64, dst, 1.0)
    sink.add(src, 4, 64, 64, other, 1.0)
    assert flushed == [] and len(sink.tasks) == 2
    sink.add(src, 4, 64, 32, dst, 1.0, dst_off=16)            # overlaps the first task's destination range
    assert flushed == [2] and len(sink.tasks) == 1


def test_gradient_accumulation_fusion_sees_through_the_gemm_major_conv2d_weight():
    """prosody._Conv2dParams keeps the reference shape [Cout,Cin,3,3] over [Cout][kh][kw][Cin] memory; the GEMM operand is the view that
    walks that memory in order, and ops._grad_of must hand the kernels the SAME walk over the (equally strided) gradient."""
    p = P._Conv2dParams(4, 8).weight
    assert tuple(p.shape) == (8, 4, 3, 3) and p.stride() == (36, 1, 12, 4)
    with torch.no_grad():
        p.copy_(torch.arange(8 * 4 * 9, dtype=torch.float32).view(8, 4, 3, 3))
    flat = torch.zeros(p.numel())
    p.grad = flat.as_strided(p.size(), p.stride())
    wf = p.permute(0, 2, 3, 1).reshape(8, 36)
    assert wf.data_ptr() == p.data_ptr() and wf.is_contiguous()                        # a view: no copy per call
    prev = ops.grad_accumulation_fusion()
    ops.set_grad_accumulation_fusion(True)
    try:
        g = ops._grad_of(wf)
        assert g is not None and g.data_ptr() == flat.data_ptr() and tuple(g.shape) == (8, 36) and g.is_contiguous()
        g.add_(wf.detach())                                                            # "the kernel accumulates"
        assert torch.equal(p.grad, p.detach())                                         # lands where autograd would have put it
        assert ops._grad_of(p) is not None
    finally:
        ops.set_grad_accumulation_fusion(prev)
    sd = {"weight": torch.randn(8, 4, 3, 3)}
    m = P._Conv2dParams(4, 8)
    m.load_state_dict({"weight": sd["weight"], "bias": torch.zeros(8)})
    assert torch.equal(m.state_dict()["weight"], sd["weight"])                          # reference layout in and out


def test_side_loss_scope_is_inert_without_a_marker_and_on_host_tensors():
    with ops.side_loss_scope():
        ops.mark_ready(torch.zeros(3))
        assert ops.take_ready(torch.zeros(3)) is None
    assert not ops._READY


def test_mask_aux_host_path_still_computes_lengths():
    lens = torch.tensor([4, 1])
    mask = torch.arange(5)[None, :] >= lens[:, None]
    assert M.mask_aux(mask)[1].tolist() == [4, 1]


def test_plane_kernel_loops_hold_only_dma_ds_read_and_mfma():
    """tools/check_pl_isa.py: every K loop of gemm_pl_kernel (two instruction orders x conv / plain) has 48 MFMAs, 24 ds_read_b128, 9 LDS-DMA
    instructions, one barrier, no scratch access and no compiler-inserted vmcnt wait."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_pl_isa.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if "gemm_desc" in l]
    assert len(lines) == 4 and all(l.startswith("ok") for l in lines), r.stdout


def test_weight_gradient_plane_kernel_loops_hold_only_dma_transpose_reads_and_mfma():
    """tools/check_pl_isa.py plw: every K loop of gemm_plw_kernel (csrc/gemm_plw.hip) has 48 MFMAs, 48 ds_read_b64_tr_b16 (two per 8-deep
    operand), 9 LDS-DMA instructions, one barrier, no scratch access and no compiler-inserted vmcnt wait."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_pl_isa.py"), "plw"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if "gemm_desc" in l]
    assert len(lines) == 4 and all(l.startswith("ok") for l in lines), r.stdout


def test_weight_gradient_plane_prefilter():
    """kernels.plane_wgrad_shape_ok mirrors plw_try's shape rules (the library's answer stays authoritative on the GPU)"""
    from ctts_amd import kernels as K
    prev = K.gemm_bf16_split_enable(True)
    try:
        if K.PLANES_ENABLED and K.PLW_ENABLED and K.SK_ENABLED:
            assert K.plane_wgrad_shape_ok(1024, 2304, 16384, 256)          # decoder FFN conv
            assert K.plane_wgrad_shape_ok(512, 2560, 16384, 512)           # PostNet conv
            assert not K.plane_wgrad_shape_ok(256, 1280, 16384, 256)       # 10 output tiles: the split-K kernel's parallel reduce wins
            assert not K.plane_wgrad_shape_ok(512, 400, 16384, 80)         # cin = 80
            assert not K.plane_wgrad_shape_ok(1000, 2304, 16384, 256)      # M % 128
        K.gemm_bf16_split_enable(False)
        assert not K.plane_wgrad_shape_ok(1024, 2304, 16384, 256)
    finally:
        K.gemm_bf16_split_enable(prev)


def test_arithmetic_switch_values_and_amp_selector():
    """kernels.gemm_bf16_split_enable: False / True / 2 / "amp" (3) / 4 -> ctts_gemm_desc.bf16_split; amp_split() = what a launch under
    torch.amp.autocast gets (one-term arithmetic with the current thresholds; None when the bf16 kernels are off)"""
    from ctts_amd import kernels as K
    prev = K.gemm_bf16_split_enable(True)
    try:
        assert K.BF16_SPLIT == 1 and K.amp_split() == 3
        assert K.gemm_bf16_split_enable(2) == 1 and K.BF16_SPLIT == 2 and K.amp_split() == 4
        assert K.gemm_bf16_split_enable("amp") == 2 and K.BF16_SPLIT == 3 and K.amp_split() == 3
        assert K.gemm_bf16_split_enable(4) == 3 and K.BF16_SPLIT == 4 and K.amp_split() == 4
        assert K.gemm_bf16_split_enable(False) == 4 and K.BF16_SPLIT == 0 and K.amp_split() is None
    finally:
        K.gemm_bf16_split_enable(prev)
