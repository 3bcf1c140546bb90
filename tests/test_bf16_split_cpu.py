"""CPU: the arithmetic of the bf16-split GEMM kernels (csrc/gemm.hip gemm_x6_kernel / gemm_x6tn_kernel), restated in numpy.

An fp32 value is split by truncation into three bf16 values, x = hi + mid + lo EXACTLY (as long as the residuals stay normal numbers:
|x| >= 2^-100, far below anything a network holds); a product is the six largest of the nine cross terms.  These tests pin the claims
the kernels' comments and DESIGN.md make: exactness of the split, bf16 representability of the pieces, and the size of what the three
dropped terms (mid * lo, lo * mid, lo * lo) can contribute: at most 2^-21 of the product (truncation leaves |mid| <= 2^-7 |hi|,
|lo| <= 2^-15 |hi|), typically below 2^-24 = the rounding of one fp32 multiply-add."""
import numpy as np


def split3(x):
    x = np.asarray(x, dtype=np.float32)

    def trunc(v):
        return (v.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    hi = trunc(x)
    r1 = (x - hi).astype(np.float32)          # exact in fp32
    mid = trunc(r1)
    r2 = (r1 - mid).astype(np.float32)        # exact
    lo = trunc(r2)
    return hi, mid, lo, (r2 - lo).astype(np.float32)


def test_three_way_split_is_exact_and_every_piece_is_a_bf16_number():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(200000).astype(np.float32) * np.float32(10.0) ** rng.integers(-20, 20, 200000).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 2.0 ** -100, 1 + 2.0 ** -23, 1 - 2.0 ** -24], dtype=np.float32)])
    x = x[(x == 0) | (np.abs(x) >= 2.0 ** -100)]
    hi, mid, lo, rest = split3(x)
    for piece in (hi, mid, lo):
        assert not np.any(piece.view(np.uint32) & np.uint32(0xFFFF)), "a piece is not representable in bf16"
    # 8 + 8 + 8 significand bits cover the 24 of an fp32 number: nothing is left, and the sum reproduces x bit for bit
    assert np.all(rest == 0)
    back = (hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)).astype(np.float32)
    assert np.array_equal(back, x)                       # (value equality: -0.0 comes back as +0.0)
    # the pieces shrink by at least 2^-7 and 2^-15 (truncation): the dropped terms are <= 2 * 2^-7 * 2^-15 + 2^-30 of hi * hi
    nz = hi != 0
    assert np.all(np.abs(mid[nz]) <= np.abs(hi[nz]) * 2.0 ** -7) and np.all(np.abs(lo[nz]) <= np.abs(hi[nz]) * 2.0 ** -15)


def test_six_terms_reproduce_the_fp32_product_to_2_pow_minus_21_worst_case_and_2_pow_minus_24_typically():
    rng = np.random.default_rng(1)
    x = rng.standard_normal(300000).astype(np.float32)
    y = (rng.standard_normal(300000) * 3).astype(np.float32)
    xh, xm, xl, _ = split3(x)
    yh, ym, yl, _ = split3(y)
    f = np.float64
    six = (xl.astype(f) * yh + xh.astype(f) * yl + xm.astype(f) * ym) + (xm.astype(f) * yh + xh.astype(f) * ym) + xh.astype(f) * yh
    exact = x.astype(f) * y.astype(f)
    dropped = np.abs(six - exact)
    bound = np.abs(exact) * 2.0 ** -21 * 1.01    # mid * lo + lo * mid + lo * lo with |mid| <= 2^-7 |hi|, |lo| <= 2^-15 |hi|
    assert np.all(dropped <= bound), float((dropped / np.maximum(np.abs(exact), 1e-300)).max())
    # typical size: an order of magnitude below the rounding of ONE fp32 multiply-add (2^-24)
    rel = dropped / np.abs(exact)
    assert float(np.median(rel)) < 2.0 ** -24


def test_products_of_short_significands_are_exact_in_every_kept_term():
    """what the GPU test relies on: operands with <= 24 significant bits against +-1 / small integers give exact integer partial sums"""
    rng = np.random.default_rng(2)
    a = rng.integers(-(1 << 17), (1 << 17) + 1, 4096).astype(np.float32)
    b = rng.integers(-1, 2, 4096).astype(np.float32)
    ah, am, al, _ = split3(a)
    bh, bm, bl, _ = split3(b)
    assert np.all(bm == 0) and np.all(bl == 0)
    six = (al * bh + ah * bl + am * bm) + (am * bh + ah * bm) + ah * bh          # all in fp32
    assert np.array_equal(six, a * b)
