"""GPU parity tests of the weight-gradient plane kernel gemm_plw_kernel (csrc/gemm_plw.hip): the TN form of ctts_gemm
(C[m][n] (+)= alpha * sum_k A[k][m] B[k + tap - pad][c], both operands reduction-major, im2col view on B) on the ROW-MAJOR pre-split plane
sets of ctts_split_planes, transposed by the LDS (ds_read_b64_tr_b16).  References: float64 on the host and the other kernels of the
library on the same launch.  Replaces (reference): the weight gradients of nn.Conv1d in the FFN `transformer_fs2.py:220-239` and in
PostNet `modules.py:140-148` (autograd of F.conv1d)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from ctts_amd import kernels as K
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def ref_wgrad(dZ, X, ksize, pad, T):
    """float64: dW[n, tap * cin + c] = sum_{b,t} dZ[b,t,n] X[b,t+tap-pad,c] (zero outside [0, T)); ksize = 0: dZ^T X"""
    dZ, X = dZ.double().cpu(), X.double().cpu()
    if not ksize:
        return dZ.t() @ X
    Bn = dZ.shape[0] // T
    dz, x = dZ.view(Bn, T, -1), X.view(Bn, T, -1)
    out = []
    for tap in range(ksize):
        s = tap - pad
        xs = torch.zeros_like(x)
        lo, hi = max(0, -s), min(T, T - s)
        if hi > lo:
            xs[:, lo:hi] = x[:, lo + s:hi + s]
        out.append(torch.einsum("btn,btc->nc", dz, xs))
    return torch.cat(out, dim=1)


def plane_wgrad(dZ, X, cout, cin, ksize=0, pad=0, T=0, out=None, expect=True, planes=True, **kw):
    rows = dZ.shape[0]
    Kd = max(ksize, 1) * cin
    out = torch.full((cout, Kd), float("nan"), device=DEV) if out is None else out
    conv = dict(conv=(T, pad, cin), conv_on_b=True) if ksize else {}
    pk = {}
    if planes:
        ap, bp = K.split_planes([dZ, X])
        pk = dict(a_planes=ap, b_planes=bp)
        took = K.gemm_takes_planes(dZ, X, out, cout, Kd, rows, cout, cin, Kd, False, False, **pk, **conv, **kw)
        assert took == expect, f"weight-gradient plane kernel eligibility: got {took}, expected {expect}"
    K.gemm(dZ, X, out, cout, Kd, rows, cout, cin, Kd, False, False, **pk, **conv, **kw)
    return out


def close(got, ref, tol=2e-6):
    scale = float(ref.abs().max())
    err = float((got.double().cpu() - ref).abs().max())
    assert err <= tol * scale, f"max |err| {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("rows,cout,cin", [(512, 128, 256), (96, 256, 512), (4096, 384, 256)])
def test_plain_tn_product_matches_float64(rows, cout, cin):
    dZ, X = rnd(rows, cout, seed=1).to(DEV), rnd(rows, cin, seed=2).to(DEV)
    got = plane_wgrad(dZ, X, cout, cin, bf16_split=2, split_k=2, split_overwrite=True)
    close(got, ref_wgrad(dZ, X, 0, 0, 0))


@pytest.mark.parametrize("rows,T,ksize,pad", [(1000, 0, 0, 0), (2 * 1000, 1000, 5, 2), (3 * 50, 50, 3, 1)])
def test_reduction_length_not_a_multiple_of_the_k_block(rows, T, ksize, pad):
    """the last 32-row K-block is partial (rows beyond the end read the hardware's out-of-range zero); with a conv view and T % 32 != 0
    every utterance boundary falls inside a K-block (conformer: T = 1000)"""
    cout, cin = 128, 256
    dZ, X = rnd(rows, cout, seed=21).to(DEV), rnd(rows, cin, seed=22).to(DEV)
    got = plane_wgrad(dZ, X, cout, cin, ksize, pad, T, bf16_split=2, split_k=2, split_overwrite=True)
    close(got, ref_wgrad(dZ, X, ksize, pad, T))


def test_transpose_detecting_operands():
    """one-hot operands: every (k, m) x (k, n) pairing lands in exactly one output element (guide rule: symmetric inputs hide transposes)"""
    rows, cout, cin = 256, 128, 256
    dZ, X = torch.zeros(rows, cout), torch.zeros(rows, cin)
    for k in range(rows):
        dZ[k, (7 * k + 3) % cout] = 1.0 + k
        X[k, (11 * k + 5) % cin] = 1.0 + 0.5 * k
    got = plane_wgrad(dZ.to(DEV), X.to(DEV), cout, cin, bf16_split=2, split_k=2, split_overwrite=True)
    assert torch.equal(got.double().cpu(), ref_wgrad(dZ, X, 0, 0, 0))


@pytest.mark.parametrize("T,ksize,pad,cin,cout,nb", [(64, 3, 1, 256, 128, 4), (40, 5, 2, 256, 128, 5), (96, 9, 4, 256, 256, 3), (64, 5, 4, 512, 128, 2),
                                                      (32, 3, 0, 256, 128, 3), (64, 9, 4, 512, 128, 2), (64, 8, 3, 256, 128, 2), (64, 2, 0, 256, 128, 3),
                                                      (64, 12, 5, 256, 128, 2), (32, 4, 1, 256, 128, 4), (64, 1, 0, 256, 128, 2)])
def test_conv_weight_gradient_dense_rows(T, ksize, pad, cin, cout, nb):
    """taps shift B by rows; rows shifted across an utterance boundary (also inside a 32-row K-block when T % 32 != 0) contribute zero.
    With T % 32 == 0 the n-tiles are FOLDED (8 / 4 / 2 taps x 32 / 64 / 128 channels share one B image of 32 + taps - 1 rows) and the
    taps beyond the folded ones get one-tap tiles: k = 9 (8 + 1), 5 (4 + 1), 3 (2 + 1), 8, 2, 12 (8 + 4 one-tap), 4, 1 (nothing to fold);
    cin = 512: two one-tap tiles per tap, 16 folded tiles"""
    rows = nb * T
    dZ, X = rnd(rows, cout, seed=3).to(DEV), rnd(rows, cin, seed=4).to(DEV)
    got = plane_wgrad(dZ, X, cout, cin, ksize, pad, T, bf16_split=2, split_k=2, split_overwrite=True, alpha=0.5)
    close(got, 0.5 * ref_wgrad(dZ, X, ksize, pad, T))


def test_conv_weight_gradient_ragged_rows_and_accumulation():
    T, ksize, pad, cin, cout = 128, 9, 4, 256, 256
    lens = [128, 17, 0, 75, 33, 0]
    rows = len(lens) * T
    mask = torch.cat([(torch.arange(T) < L).float() for L in lens])[:, None]
    dZ, X = (rnd(rows, cout, seed=5) * mask).to(DEV), rnd(rows, cin, seed=6).to(DEV)
    rl = dict(row_lens=torch.tensor(lens, dtype=torch.int32, device=DEV), row_T=T, row_halo=0)
    ref = ref_wgrad(dZ, X, ksize, pad, T)
    got = plane_wgrad(dZ, X, cout, cin, ksize, pad, T, bf16_split=2, split_k=2, split_overwrite=True, **rl)
    close(got, ref)
    # split_k > 1 without split_overwrite: C += alpha A^T B, added by the tile's owner in place
    base = rnd(cout, ksize * cin, seed=7).to(DEV)
    acc = plane_wgrad(dZ, X, cout, cin, ksize, pad, T, out=base.clone(), bf16_split=2, split_k=2, alpha=2.0, **rl)
    close(acc, base.double().cpu() + 2.0 * ref)
    # all utterances empty: the product is zero / the target unchanged
    rl0 = dict(row_lens=torch.zeros(len(lens), dtype=torch.int32, device=DEV), row_T=T, row_halo=0)
    z = plane_wgrad(dZ * 0, X, cout, cin, ksize, pad, T, bf16_split=2, split_k=2, split_overwrite=True, **rl0)
    assert torch.equal(z, torch.zeros_like(z))
    keep = plane_wgrad(dZ * 0, X, cout, cin, ksize, pad, T, out=base.clone(), bf16_split=2, split_k=2, **rl0)
    assert torch.equal(keep, base)


def test_cut_tiles_hand_their_partial_sums_over_in_a_fixed_order():
    """long reduction, few tiles: every tile is cut into many pieces (slab + flag hand-off); bit-identical on repetition"""
    T, ksize, pad, cin, cout, nb = 1024, 5, 2, 256, 256, 16
    rows = nb * T
    dZ, X = rnd(rows, cout, seed=8).to(DEV), rnd(rows, cin, seed=9).to(DEV)
    a = plane_wgrad(dZ, X, cout, cin, ksize, pad, T, bf16_split=2, split_k=2, split_overwrite=True)      # 2: below the kernel's tile-count threshold
    b = plane_wgrad(dZ, X, cout, cin, ksize, pad, T, bf16_split=2, split_k=2, split_overwrite=True)
    assert torch.equal(a, b)
    close(a, ref_wgrad(dZ, X, ksize, pad, T), tol=3e-6)
    K.WorkspaceErrorProbe().poll_and_check()


def test_full_size_ffn_weight_gradient_against_the_other_kernels():
    """the launch of the fs2 step (decoder FFN conv k = 9, 256 -> 1024, 16 x 1024 ragged rows): plane kernel vs the in-kernel-split kernel
    vs fp32 MFMA on the same operands"""
    from ctts_amd.synthetic import CANONICAL_SRC_LENS
    T, ksize, pad, cin, cout, nb = 1024, 9, 4, 256, 1024, 16
    lens = torch.tensor([8 * v for v in CANONICAL_SRC_LENS], dtype=torch.int32)
    rows = nb * T
    mask = (torch.arange(T)[None, :] < lens[:, None]).float().reshape(-1, 1)
    dZ, X = (rnd(rows, cout, seed=10) * mask).to(DEV), rnd(rows, cin, seed=11).to(DEV)
    rl = dict(row_lens=lens.to(DEV), row_T=T, row_halo=0)
    pl = plane_wgrad(dZ, X, cout, cin, ksize, pad, T, bf16_split=1, split_k=4, split_overwrite=True, **rl)
    x6 = plane_wgrad(dZ, X, cout, cin, ksize, pad, T, planes=False, bf16_split=1, split_k=4, split_overwrite=True, **rl)
    f32 = plane_wgrad(dZ, X, cout, cin, ksize, pad, T, planes=False, bf16_split=0, split_k=4, split_overwrite=True, **rl)
    scale = float(f32.abs().max())
    assert float((pl - x6).abs().max()) <= 2e-6 * scale and float((pl - f32).abs().max()) <= 4e-6 * scale


def test_ineligible_descriptors_stay_on_the_other_kernels():
    dZ, X = rnd(512, 128, seed=12).to(DEV), rnd(512, 128, seed=13).to(DEV)
    got = plane_wgrad(dZ, X, 128, 128, expect=False, bf16_split=2, split_k=2, split_overwrite=True)       # N = 128: not a 256-column tile
    close(got, ref_wgrad(dZ, X, 0, 0, 0))
    dZ, X = rnd(4 * 64, 128, seed=14).to(DEV), rnd(4 * 64, 128, seed=15).to(DEV)
    got = plane_wgrad(dZ, X, 128, 128, 3, 1, 64, expect=False, bf16_split=2, split_k=2, split_overwrite=True)      # cin % 256 != 0
    close(got, ref_wgrad(dZ, X, 3, 1, 64))
