"""GPU: the batched, length-limited attention products with a long reduction (O = P V, dV = P^T dO, dQ = dS K: ops._SelfAttention) at the
decoder shape against float64 - every utterance length that cuts a tile, a 16-byte chunk or a K-block (0, 1, 3, 63, 64, 65, 127, 511, 513,
1000, 1024), the "write everything" zero rule outside the limits (NaN-filled outputs), bit-identical repetition.  Written in round 6 next
to two alternative kernels for these launches (32 x 64 two-group tiles; 64 x 64 tiles with eight waves) that passed it and were measured
away: it stays as the direct test of the launches the train step makes."""
import pytest
import torch

import ctts_amd  # noqa: F401
from ctts_amd import kernels as K

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0") if torch.cuda.is_available() else None

LENS = [1024, 0, 1, 3, 63, 64, 65, 127, 511, 513, 1000, 770]


def _setup(T=1024, H=2, dh=128):
    B = len(LENS)
    C = H * dh
    g = torch.Generator().manual_seed(7)
    lens = torch.tensor(LENS, dtype=torch.int32, device=DEV)
    P = torch.rand(B, H, T, T, generator=g).to(DEV)
    qkv = torch.randn(B, T, 3 * C, generator=g).to(DEV)
    dO = torch.randn(B, T, C, generator=g).to(DEV)
    return B, T, H, dh, C, lens, P, qkv, dO


def test_pv_and_dv_match_float64_for_every_length():
    B, T, H, dh, C, lens, P, qkv, dO = _setup()
    C3 = 3 * C
    sP = (H * T * T, T * T)
    out = torch.full((B, T, C), float("nan"), device=DEV)
    K.gemm(P, qkv, out, T, dh, T, T, C3, C, True, False, b_off=2 * C, nb0=B, nb1=H, sA=sP, sB=(T * C3, dh), sC=(T * C, dh), lens=lens,
           lim=(1, 0, 1), split_overwrite=True)
    dqkv = torch.full((B, T, C3), float("nan"), device=DEV)
    K.gemm(P, dO, dqkv, T, dh, T, T, C, C3, False, False, c_off=2 * C, nb0=B, nb1=H, sA=sP, sB=(T * C, dh), sC=(T * C3, dh), lens=lens,
           lim=(1, 0, 1), alpha=0.5, split_overwrite=True)
    torch.cuda.synchronize()
    V = qkv[:, :, 2 * C:].view(B, T, H, dh).double()
    worst_o = worst_v = 0.0
    for b, L in enumerate(LENS):
        for hh in range(H):
            Pb = P[b, hh, :L, :L].double()
            ref_o = Pb @ V[b, :L, hh]                                           # [L, dh]
            got_o = out[b, :L, hh * dh:(hh + 1) * dh].double()
            ref_v = 0.5 * (Pb.t() @ dO[b, :L, hh * dh:(hh + 1) * dh].double())
            got_v = dqkv[b, :L, 2 * C + hh * dh:2 * C + (hh + 1) * dh].double()
            if L:
                worst_o = max(worst_o, float((got_o - ref_o).abs().max() / ref_o.abs().max().clamp(min=1e-30)))
                worst_v = max(worst_v, float((got_v - ref_v).abs().max() / ref_v.abs().max().clamp(min=1e-30)))
        # "write everything": rows beyond the utterance's length are zeros, not the NaN the buffers were filled with
        assert float(out[b, L:].abs().max()) == 0.0 if L < T else True
        assert float(dqkv[b, L:, 2 * C:].abs().max()) == 0.0 if L < T else True
    assert torch.isfinite(out).all() and torch.isfinite(dqkv[:, :, 2 * C:]).all()
    assert worst_o < 2e-6 and worst_v < 2e-6, (worst_o, worst_v)                 # fp32 MFMA products, 1,024-deep fp32 accumulation
    out2 = torch.empty_like(out)
    K.gemm(P, qkv, out2, T, dh, T, T, C3, C, True, False, b_off=2 * C, nb0=B, nb1=H, sA=sP, sB=(T * C3, dh), sC=(T * C, dh), lens=lens,
           lim=(1, 0, 1), split_overwrite=True)
    assert torch.equal(out, out2)                                                # fixed summation order: bit-identical repetition


def test_dq_shaped_product_with_alpha_and_offsets():
    """dQ = scale * dS K (NN, A = the [T, T] map, B = the K slice of the packed projection, C = the Q slice of dqkv)"""
    B, T, H, dh, C, lens, dS, qkv, _ = _setup()
    C3 = 3 * C
    sP = (H * T * T, T * T)
    dqkv = torch.full((B, T, C3), float("nan"), device=DEV)
    K.gemm(dS, qkv, dqkv, T, dh, T, T, C3, C3, True, False, b_off=C, c_off=0, nb0=B, nb1=H, sA=sP, sB=(T * C3, dh), sC=(T * C3, dh),
           lens=lens, lim=(1, 0, 1), alpha=0.25, split_overwrite=True)
    torch.cuda.synchronize()
    Kk = qkv[:, :, C:2 * C].view(B, T, H, dh).double()
    worst = 0.0
    for b, L in enumerate(LENS):
        if not L:
            continue
        for hh in range(H):
            ref = 0.25 * (dS[b, hh, :L, :L].double() @ Kk[b, :L, hh])
            got = dqkv[b, :L, hh * dh:(hh + 1) * dh].double()
            worst = max(worst, float((got - ref).abs().max() / ref.abs().max()))
    assert worst < 2e-6, worst
    assert torch.isfinite(dqkv[:, :, :C]).all()
