"""GPU parity tests of the pre-split operand path (round 5): ctts_split_planes (exact three-way bf16 split, include/ctts.h) and the
persistent plane kernel gemm_pl_kernel (csrc/gemm_pl.hip) behind ctts_gemm_desc.A_planes / B_planes, plus the arithmetic DOMAIN of all
bf16-split kernels (both-operand exactness, tiny / huge / infinite / NaN operands) and the per-descriptor arithmetic
switch (ctts_gemm_desc.bf16_split).  References: float64 on the host (the oracle of a GEMM), the fp32-MFMA kernels
of the same library on the same launch, and torch's own round-to-nearest-even bf16 conversion for the split.
Replaces (reference): nn.Conv1d of the FFN `transformer_fs2.py:220-239` and of PostNet `modules.py:140-148` (forward and data gradient)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from ctts_amd import kernels as K
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def planes_to_f64(pl):
    """plane set [rows, cols / 32, 3, 32] bf16 -> the three pieces as float64 [rows, cols]"""
    return [K.planes_piece(pl, q).float().double().cpu() for q in range(3)]


# ---------------------------------------------------------------------------------------------------------------- the split itself
def test_split_planes_is_the_exact_round_to_nearest_split():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(300, 288, generator=g) * torch.exp(torch.randn(300, 288, generator=g) * 6)
    x[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 2.0 ** -100, 2.0 ** -120])
    pl = K.split_planes([x.to(DEV)])[0]
    assert pl.dtype == torch.bfloat16 and tuple(pl.shape) == (300, 9, 3, 32)
    hi, mid, lo = planes_to_f64(pl)
    xd = x.double()
    assert torch.equal(hi + mid + lo, xd)                                            # exact: nothing is dropped by the split
    assert torch.equal(K.planes_piece(pl, 0).cpu(), x.bfloat16())                                    # hi = RNE(x)
    r1 = (xd - hi).float()
    assert torch.equal(K.planes_piece(pl, 1).cpu(), r1.bfloat16())                                   # mid = RNE(x - hi)
    assert float((mid.abs() - hi.abs() * 2.0 ** -8).max()) <= 0 and float((lo.abs() - hi.abs() * 2.0 ** -16).max()) <= 0


def test_split_planes_domain_huge_finite_values_infinities_and_nans():
    big = torch.tensor([3.3961e38, 3.4e38, 3.4028234e38, -3.4028234e38, 3.3895e38, float("inf"), -float("inf"), float("nan")])
    x = torch.zeros(8, 32)
    x[0, :8] = big
    pl = K.split_planes([x.to(DEV)])[0]
    hi, mid, lo = planes_to_f64(pl)
    hi, mid, lo, x = hi[:, :8], mid[:, :8], lo[:, :8], x[:, :8]
    fin = torch.isfinite(x[0])
    # finite values that would round to infinity: hi clamped to the largest bf16, the remainder stays exact
    assert torch.isfinite(hi[0][fin]).all() and torch.equal((hi + mid + lo)[0][fin], x[0].double()[fin])
    assert float(hi[0][fin].abs().max()) == float(torch.tensor(3.3895313892515355e38))
    # infinities and NaN: hi carries them, the remainders are zero
    assert hi[0][5] == float("inf") and hi[0][6] == -float("inf") and torch.isnan(hi[0][7])
    assert torch.equal(mid[0][5:], torch.zeros(3, dtype=torch.float64)) and torch.equal(lo[0][5:], torch.zeros(3, dtype=torch.float64))


def test_split_planes_many_tasks_per_launch_and_bad_arguments():
    mats = [rnd(37 + i, 32 * (1 + i % 5), seed=i).to(DEV) for i in range(30)]           # > 24: two launches
    for m, pl in zip(mats, K.split_planes(mats)):
        hi, mid, lo = planes_to_f64(pl)
        assert torch.equal(hi + mid + lo, m.double().cpu())
    with pytest.raises(Exception):
        K.split_planes([rnd(4, 24).to(DEV)])                                        # cols % 32 != 0


# ---------------------------------------------------------------------------------------------------------------- the plane GEMM
def _plane_gemm(A, Bm, M, N, Kd, lda=None, out=None, expect=True, **kw):
    lda = Kd if lda is None else lda
    out = torch.full((M, N), float("nan"), device=DEV) if out is None else out
    ap, bp = K.split_planes([A.reshape(-1, lda).contiguous(), Bm])
    args = (A, Bm, out, M, N, Kd, lda, Kd, N, True, True)
    took = K.gemm_takes_planes(*args, a_planes=ap, b_planes=bp, **kw)
    assert took == expect, f"plane kernel eligibility: got {took}, expected {expect}"
    K.gemm(*args, a_planes=ap, b_planes=bp, **kw)
    return out


def _err_word():
    return max(int(ws.view(torch.int32)[2048].item()) for ws in K._SK_WS.values())


def test_plane_gemm_is_exact_on_integers_with_both_operands_multi_piece_and_fp32_class_on_random_data():
    """(1) 12-bit x 12-bit integers on BOTH sides (hi and mid pieces non-zero on both operands: exercises hi*hi, hi*mid, mid*hi and mid*mid -
    a swapped entry of the kernel's term table for these terms fails here; VERDICT r04 weak #1a), at most four non-zero products per
    output so that every fp32 partial sum is an exact integer below 2^24; (2) 18-bit integers against sparse +-1 in both roles (all three
    pieces of one operand: lo*hi and hi*lo); (3) random data: error against float64 not above the fp32-MFMA kernels' on the same launch,
    two runs bit-identical, hand-off error word clean."""
    M, N, Kd = 4096, 512, 1024
    g = torch.Generator().manual_seed(5)

    def i12(rows):
        return torch.randint(-(1 << 11), (1 << 11) + 1, (rows, Kd), generator=g).float()

    def few(rows, per_row, vals):
        out = torch.zeros(rows, Kd)
        idx = torch.stack([torch.randperm(Kd, generator=g)[:per_row] for _ in range(rows)])
        out.scatter_(1, idx, vals(rows, per_row))
        return out

    v12 = lambda r, c: torch.randint(-(1 << 11), (1 << 11) + 1, (r, c), generator=g).float()
    pm1 = lambda r, c: (torch.randint(0, 2, (r, c), generator=g) * 2 - 1).float()
    i18 = lambda rows: torch.randint(-(1 << 17), (1 << 17) + 1, (rows, Kd), generator=g).float()
    cases = [(i12(M), few(N, 4, v12)), (few(M, 4, v12), i12(N)), (i18(M), few(N, 32, pm1)), (few(M, 32, pm1), i18(N))]
    for A, Bm in cases:
        ref = A.double() @ Bm.double().t()
        assert float(ref.abs().max()) <= 2 ** 24
        got = _plane_gemm(A.to(DEV), Bm.to(DEV), M, N, Kd, bf16_split=2)
        assert torch.equal(got.double().cpu(), ref), float((got.double().cpu() - ref).abs().max())
    A, Bm = torch.randn(M, Kd, generator=g).to(DEV), (torch.randn(N, Kd, generator=g) * 0.1).to(DEV)
    ref = A.double() @ Bm.double().t()
    p1 = _plane_gemm(A, Bm, M, N, Kd, bf16_split=2)
    assert torch.equal(p1, _plane_gemm(A, Bm, M, N, Kd, bf16_split=2))
    f32 = torch.full((M, N), float("nan"), device=DEV)
    K.gemm(A, Bm, f32, M, N, Kd, Kd, Kd, N, True, True, bf16_split=0)
    e6, e32 = float((p1.double() - ref).abs().max()), float((f32.double() - ref).abs().max())
    print(f"plane kernel max |err| vs fp64 {e6:.3e}, fp32 MFMA {e32:.3e}")
    assert e6 <= 1.25 * e32 + 1e-7, (e6, e32)
    assert _err_word() == 0


@pytest.mark.parametrize("case", ["gelu_drop", "plain", "relu", "swish", "bdrs"])
def test_plane_gemm_conv_view_ragged_rows_and_epilogues_equal_the_fp32_kernels(case):
    """The launch of the decoder FFN convolution in small (im2col view on A, ragged utterances, every epilogue the kernel carries) on the
    plane kernel and on the fp32-MFMA kernels: equal to accumulation-order noise, identical zeros (the 64-row zero rule included: an
    utterance of 65..128 valid rows leaves the upper half of its 128-row tile to the zero writers)."""
    B_, T, Cin, N, ks = 16, 512, 128, 768, 5
    M, Kd = B_ * T, ks * Cin
    lens = torch.tensor([512, 200, 129, 64, 330, 1, 448, 449, 384, 385, 511, 65, 63, 128, 300, 256], dtype=torch.int32, device=DEV)
    x, w, bias = rnd(B_, T, Cin, seed=301).to(DEV), rnd(N, Kd, seed=302, scale=0.05).to(DEV), rnd(N, seed=303).to(DEV)
    R = rnd(B_, T, N, seed=304).to(DEV)
    rs = (torch.arange(T, device=DEV)[None, :] < lens[:, None]).float().reshape(-1).contiguous()
    seed = torch.full((1,), 7, dtype=torch.int64, device=DEV)
    kw = dict(conv=(T, ks // 2, Cin), alpha=0.5, bias=bias, row_lens=lens, row_T=T, row_halo=0)
    act = 0
    if case == "gelu_drop":
        act = 2
        kw.update(act=2, p_drop=0.2, seed=seed, drop_offset=3)
    elif case in ("relu", "swish"):
        act = 1 if case == "relu" else 4
        kw.update(act=act)
    elif case == "bdrs":
        kw.update(p_drop=0.2, seed=seed, drop_offset=3, R=R, ldr=N, rowscale=rs)
    outs = {}
    for planes in (True, False):
        out = torch.full((B_, T, N), float("nan"), device=DEV)
        Z = torch.full((B_, T, N), float("nan"), device=DEV) if act else None
        k2 = dict(kw)
        if act:
            k2.update(Z=Z, ldz=N)
        if planes:
            _plane_gemm(x, w, M, N, Kd, lda=Cin, out=out, bf16_split=2, **k2)
        else:
            K.gemm(x, w, out, M, N, Kd, Cin, Kd, N, True, True, bf16_split=0, tile_map=K.row_tile_map(lens, T, 0, M), **k2)
        outs[planes] = (out, Z)
    (o6, z6), (o32, z32) = outs[True], outs[False]
    assert torch.isfinite(o6).all() and (o6 == 0).eq(o32 == 0).all()
    assert float((o6 - o32).abs().max()) <= 2e-5
    if act:
        assert float((z6 - z32).abs().max()) <= 2e-5 and (z6 == 0).eq(z32 == 0).all()
    assert _err_word() == 0


@pytest.mark.parametrize("ragged", [False, True])
def test_plane_gemm_data_gradient_shape_long_reduction_with_halo_vs_fp64(ragged):
    """The FFN conv data gradient in small: N = 256 (ONE n-tile), K = 9 x 512 walked channel-block-major, every tile cut between
    workgroups (stream-K hand-off in a fixed order), row halo, split_overwrite semantics (every element written).  Against float64."""
    B_, T, Cout, Cin, ks = 4, 256, 512, 256, 9
    M, Kd = B_ * T, ks * Cout
    lens = torch.tensor([256, 130, 64, 200], dtype=torch.int32, device=DEV)
    dz = rnd(B_, T, Cout, seed=11).to(DEV)
    if ragged:
        dz = dz * (torch.arange(T, device=DEV)[None, :, None] < lens[:, None, None])
    wd = rnd(Cin, Kd, seed=12, scale=0.03).to(DEV)
    kw = dict(conv=(T, ks // 2, Cout), alpha=0.7, split_overwrite=True)
    if ragged:
        kw.update(row_lens=lens, row_T=T, row_halo=ks // 2)
    got = _plane_gemm(dz, wd, M, Cin, Kd, lda=Cout, bf16_split=2, **kw).view(B_, T, Cin)
    dzp = torch.nn.functional.pad(dz.double(), (0, 0, ks // 2, ks // 2))
    cols = torch.cat([dzp[:, kk:kk + T] for kk in range(ks)], dim=-1)                 # [B, T, (tap, c)]
    ref = 0.7 * (cols @ wd.double().t())
    if ragged:                                                                       # wholly padded 64-row blocks (beyond len + halo) are zero
        t = torch.arange(T, device=DEV)[None, :]
        live = ((t // 64) * 64 < (lens[:, None] + ks // 2))
        assert float(got[~live].abs().max()) == 0.0
        ref = ref * live[..., None]
    err = float((got.double() - ref).abs().max())
    print(f"dgrad-shaped plane GEMM: max |err| vs fp64 {err:.3e} (|ref| max {float(ref.abs().max()):.2f})")
    assert err <= 2e-5 * max(1.0, float(ref.abs().max()))
    again = _plane_gemm(dz, wd, M, Cin, Kd, lda=Cout, bf16_split=2, **kw).view(B_, T, Cin)
    assert torch.equal(got, again) and _err_word() == 0


def test_plane_gemm_is_refused_where_it_does_not_apply_and_the_launch_still_runs():
    """Descriptors the plane kernel does not carry (bf16_split = 0, an epilogue outside its set, N not a multiple of 128) run on the other
    kernels from the fp32 operands - same result."""
    M, N, Kd = 2048, 512, 512
    A, Bm = rnd(M, Kd, seed=1).to(DEV), rnd(N, Kd, seed=2, scale=0.1).to(DEV)
    ref = (A.double() @ Bm.double().t()).float()
    for kw in (dict(bf16_split=0), dict(bf16_split=2, act=3)):                        # tanh is not among the lean epilogues it carries
        out = _plane_gemm(A, Bm, M, N, Kd, expect=False, **kw)
        want = torch.tanh(ref) if kw.get("act") == 3 else ref
        assert float((out - want).abs().max()) <= 3e-5
    out = _plane_gemm(A, Bm[:500].contiguous(), M, 500, Kd, expect=False, bf16_split=2)
    assert float((out - ref[:, :500]).abs().max()) <= 3e-5


# ---------------------------------------------------------------------------------------------------------------- arithmetic domain
def _nt(A, Bm, M, N, Kd, mode, **kw):
    """the same NT launch on: 'f32' fp32 MFMA, 'x6' in-kernel split, 'pl' pre-split planes"""
    out = torch.full((M, N), float("nan"), device=DEV)
    if mode == "pl":
        return _plane_gemm(A, Bm, M, N, Kd, out=out, bf16_split=2, **kw)
    if mode == "x6":
        assert K.gemm_takes_bf16_split(A, Bm, out, M, N, Kd, Kd, Kd, N, True, True, bf16_split=2, **kw)
    K.gemm(A, Bm, out, M, N, Kd, Kd, Kd, N, True, True, bf16_split=2 if mode == "x6" else 0, **kw)
    return out


def _tn(dz, x, Mo, No, Kred, mode):
    out = torch.full((Mo, No), float("nan"), device=DEV)
    args = (dz, x, out, Mo, No, Kred, Mo, No, No, False, False)
    if mode == "x6":
        assert K.gemm_takes_bf16_split(*args, bf16_split=2)
    K.gemm(*args, bf16_split=2 if mode == "x6" else 0)
    return out


@pytest.mark.parametrize("mode", ["x6", "pl"])
def test_bf16_split_domain_nt_tiny_huge_inf_nan(mode):
    """include/ctts.h DOMAIN of the bf16-split kernels, NT launches (VERDICT r04 weak #1b):
    operands scaled by 2^-110 against 2^120 (products of normal size), a column of 3.0e38, FLT_MAX-sized values, one Inf, one NaN."""
    M, N, Kd = 2048, 256, 256
    g = torch.Generator().manual_seed(21)
    A0, B0 = torch.randn(M, Kd, generator=g), torch.randn(N, Kd, generator=g)
    # (a) tiny x large: the lo piece of the tiny operand leaves bf16's normal range - the product keeps >= 16 significant bits
    A, Bm = (A0 * 2.0 ** -110).to(DEV), (B0 * 2.0 ** 120).to(DEV)
    ref = (A.double() @ Bm.double().t())
    got, f32 = _nt(A, Bm, M, N, Kd, mode), _nt(A, Bm, M, N, Kd, "f32")
    assert torch.isfinite(got).all() and torch.isfinite(f32).all()
    scale = float(ref.abs().max())
    e, e32 = float((got.double() - ref).abs().max()) / scale, float((f32.double() - ref).abs().max()) / scale
    print(f"[{mode}] 2^-110 x 2^120: rel err {e:.3e} (fp32 MFMA {e32:.3e})")
    assert e <= 2.0 ** -13
    # (b) a column of 3.0e38 (below the band where hi rounds to infinity) against 2^-100-sized partners: finite, fp32-class
    A = A0.clone(); A[:, 7] = 3.0e38
    A, Bm = A.to(DEV), (B0 * 2.0 ** -100).to(DEV)
    ref = A.double() @ Bm.double().t()
    got = _nt(A, Bm, M, N, Kd, mode)
    assert torch.isfinite(got).all()
    assert float((got.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    # (c) FLT_MAX-sized finite values: exact through the planes (hi clamped); treated like an infinity by the in-kernel split
    A = A0.clone(); A[5, 3] = 3.4028234e38
    A, Bm = A.to(DEV), (B0 * 2.0 ** -100).to(DEV)
    got, f32 = _nt(A, Bm, M, N, Kd, mode), _nt(A, Bm, M, N, Kd, "f32")
    assert torch.isfinite(f32).all()
    if mode == "pl":
        ref = A.double() @ Bm.double().t()
        assert torch.isfinite(got).all() and float((got.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    else:
        assert torch.isfinite(got[torch.arange(M) != 5]).all() and not torch.isfinite(got[5]).any()
    # (d) one Inf and one NaN: the finite / non-finite pattern of the result equals the fp32-MFMA kernels'
    A = A0.clone(); A[9, 100] = float("inf"); A[1000, 0] = float("nan")
    Bz = B0.clone(); Bz[17, 100] = 0.0                                               # inf * 0 = NaN in fp32 as well
    A, Bm = A.to(DEV), Bz.to(DEV)
    got, f32 = _nt(A, Bm, M, N, Kd, mode), _nt(A, Bm, M, N, Kd, "f32")
    assert torch.equal(torch.isfinite(got), torch.isfinite(f32))
    assert not torch.isfinite(got[9]).any() and not torch.isfinite(got[1000]).any() and torch.isfinite(got[10]).all()


def test_bf16_split_domain_tn_tiny_huge_inf_nan_and_both_operand_exactness():
    """the same for the weight-gradient kernel (gemm_x6tn_kernel), plus 12-bit x 12-bit integers on both sides (TN layout)."""
    if os.environ.get("CTTS_X6_TN", "1") == "0":
        pytest.skip("CTTS_X6_TN=0")
    Mo, No, Kred = 256, 256, 4096
    g = torch.Generator().manual_seed(22)
    # exact: dense 12-bit integers against <= 4 non-zero 12-bit integers per output column (all partial sums < 2^24)
    dz = torch.randint(-(1 << 11), (1 << 11) + 1, (Kred, Mo), generator=g).float()
    x = torch.zeros(Kred, No)
    for n in range(No):
        rows = torch.randperm(Kred, generator=g)[:4]
        x[rows, n] = torch.randint(-(1 << 11), (1 << 11) + 1, (4,), generator=g).float()
    for a, b in ((dz, x), (x[:, :Mo].contiguous(), dz[:, :No].contiguous())):
        ref = a.double().t() @ b.double()
        assert float(ref.abs().max()) <= 2 ** 24
        got = _tn(a.to(DEV), b.to(DEV), Mo, No, Kred, "x6")
        assert torch.equal(got.double().cpu(), ref), float((got.double().cpu() - ref).abs().max())
    D0, X0 = torch.randn(Kred, Mo, generator=g), torch.randn(Kred, No, generator=g)
    dzd, xd = (D0 * 2.0 ** -110).to(DEV), (X0 * 2.0 ** 120).to(DEV)
    ref = dzd.double().t() @ xd.double()
    got = _tn(dzd, xd, Mo, No, Kred, "x6")
    assert torch.isfinite(got).all() and float((got.double() - ref).abs().max()) <= 2.0 ** -13 * float(ref.abs().max())
    D = D0.clone(); D[77, 5] = float("inf"); D[99, 200] = float("nan")
    dzd, xd = D.to(DEV), X0.to(DEV)
    got, f32 = _tn(dzd, xd, Mo, No, Kred, "x6"), _tn(dzd, xd, Mo, No, Kred, "f32")
    assert torch.equal(torch.isfinite(got), torch.isfinite(f32))
    assert not torch.isfinite(got[5]).any() and not torch.isfinite(got[200]).any() and torch.isfinite(got[6]).all()


def test_arithmetic_is_chosen_per_descriptor():
    """ctts_gemm_desc.bf16_split replaces the process-wide switch: two launches of one process differ, a
    zero-initialised descriptor is plain fp32, and kernels.gemm_bf16_split_enable only sets this module's default."""
    M, N, Kd = 8192, 768, 512
    A, Bm, out = rnd(M, Kd, seed=1).to(DEV), rnd(N, Kd, seed=2).to(DEV), torch.empty(M, N, device=DEV)
    args = (A, Bm, out, M, N, Kd, Kd, Kd, N, True, True)
    assert K.gemm_takes_bf16_split(*args, bf16_split=1) and not K.gemm_takes_bf16_split(*args, bf16_split=0)
    prev = K.gemm_bf16_split_enable(False)
    try:
        assert not K.gemm_takes_bf16_split(*args) and K.gemm_takes_bf16_split(*args, bf16_split=1)
    finally:
        K.gemm_bf16_split_enable(prev)
    assert K.gemm_takes_bf16_split(*args) == (prev >= 1)
