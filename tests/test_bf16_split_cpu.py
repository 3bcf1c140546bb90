"""CPU: the arithmetic of the bf16-split GEMM kernels (csrc/gemm.hip gemm_x6_kernel / gemm_x6tn_kernel), restated in numpy.

An fp32 value is split into three bf16 values, each the round-to-nearest-even of what is left (v_cvt_pk_bf16_f32): x = hi + mid + lo
EXACTLY (as long as the residuals stay normal numbers: |x| >= 2^-100, far below anything a network holds); a product is the six largest
of the nine cross terms.  These tests pin the claims the kernels' comments and DESIGN.md make: exactness of the split, bf16
representability of the pieces, and the size of what the three dropped terms (mid * lo, lo * mid, lo * lo) can contribute: at most
2^-24 of the product - ONE fp32 rounding - (|mid| <= 2^-8 |hi|, |lo| <= 2^-16 |hi|), 2^-29 in the median."""
import numpy as np


def split3(x):
    x = np.asarray(x, dtype=np.float32)

    def rne(v):                                 # float32 -> nearest bf16 (ties to even), as a float32
        u = v.view(np.uint32).astype(np.uint64)
        return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)
    hi = rne(x)
    r1 = (x - hi).astype(np.float32)          # exact in fp32
    mid = rne(r1)
    r2 = (r1 - mid).astype(np.float32)        # exact
    lo = rne(r2)
    return hi, mid, lo, (r2 - lo).astype(np.float32)


def test_three_way_split_is_exact_and_every_piece_is_a_bf16_number():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(200000).astype(np.float32) * np.float32(10.0) ** rng.integers(-20, 20, 200000).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 2.0 ** -100, 1 + 2.0 ** -23, 1 - 2.0 ** -24, 1.99609375, 255.5], dtype=np.float32)])
    x = x[(x == 0) | (np.abs(x) >= 2.0 ** -100)]
    hi, mid, lo, rest = split3(x)
    for piece in (hi, mid, lo):
        assert not np.any(piece.view(np.uint32) & np.uint32(0xFFFF)), "a piece is not representable in bf16"
    # 8 + 8 + 8 significand bits cover the 24 of an fp32 number: nothing is left, and the sum reproduces x bit for bit
    assert np.all(rest == 0)
    back = (hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)).astype(np.float32)
    assert np.array_equal(back, x)                       # (value equality: -0.0 comes back as +0.0)
    # the pieces shrink by at least 2^-8 and 2^-16 (rounding to nearest): the dropped terms are <= 2 * 2^-8 * 2^-16 + 2^-32 of hi * hi
    nz = hi != 0
    assert np.all(np.abs(mid[nz]) <= np.abs(hi[nz]) * 2.0 ** -8) and np.all(np.abs(lo[nz]) <= np.abs(hi[nz]) * 2.0 ** -16)


def test_six_terms_reproduce_the_fp32_product_to_one_fp32_rounding():
    rng = np.random.default_rng(1)
    x = rng.standard_normal(300000).astype(np.float32)
    y = (rng.standard_normal(300000) * 3).astype(np.float32)
    xh, xm, xl, _ = split3(x)
    yh, ym, yl, _ = split3(y)
    f = np.float64
    six = (xl.astype(f) * yh + xh.astype(f) * yl + xm.astype(f) * ym) + (xm.astype(f) * yh + xh.astype(f) * ym) + xh.astype(f) * yh
    exact = x.astype(f) * y.astype(f)
    dropped = np.abs(six - exact)
    bound = np.abs(exact) * 2.0 ** -23 * 1.01    # mid * lo + lo * mid + lo * lo with |mid| <= 2^-8 |hi|, |lo| <= 2^-16 |hi| (|hi| may exceed |x| by 2^-8)
    assert np.all(dropped <= bound), float((dropped / np.maximum(np.abs(exact), 1e-300)).max())
    # measured: max 2^-24.2 = the rounding of ONE fp32 multiply-add, median 2^-29
    rel = dropped / np.abs(exact)
    assert float(rel.max()) < 2.0 ** -24 * 1.05 and float(np.median(rel)) < 2.0 ** -28


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
