"""Where the unfused fs2 attention's time goes: each of its batched products (ops._SelfAttention) timed alone at the decoder shape
(B = 16, T = 1024, 2 heads x 128) for three length distributions with the SAME padded shape: the canonical ragged lengths, all-1024
(dense) and a uniform length with the canonical batch's sum of squares - separating the kernels' own efficiency from the imbalance
between utterances.  HIP events on the launch stream; TFLOP/s of VALID work."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctts_amd  # noqa
from ctts_amd import kernels as K
from ctts_amd.synthetic import CANONICAL_SRC_LENS
DEV = torch.device("cuda:0")
B, T, H, C = 16, 1024, 2, 256
dh, C3 = C // H, 3 * C
scale = dh ** -0.5


def timeit(fn, iters=30, warm=10):
    for _ in range(warm):
        fn()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def run(name, lens_list):
    lens = torch.tensor(lens_list, dtype=torch.int32, device=DEV)
    v2 = sum(int(l) ** 2 for l in lens_list)
    gf = 2 * dh * H * v2 / 1e9
    qkv = torch.randn(B, T, C3, device=DEV)
    dO = torch.randn(B, T, C, device=DEV)
    S = torch.empty(B, H, T, T, device=DEV)
    P = torch.rand(B, H, T, T, device=DEV)
    out = torch.empty(B, T, C, device=DEV)
    dqkv = torch.empty_like(qkv)
    sP = (H * T * T, T * T)
    prods = {
        "S = Q K^T (NT, K=128)": lambda: K.gemm(qkv, qkv, S, T, T, dh, C3, C3, T, True, True, a_off=0, b_off=C, nb0=B, nb1=H, sA=(T * C3, dh), sB=(T * C3, dh), sC=sP, lens=lens, lim=(1, 1, 0), alpha=scale),
        "O = P V   (NN, K=T)": lambda: K.gemm(P, qkv, out, T, dh, T, T, C3, C, True, False, b_off=2 * C, nb0=B, nb1=H, sA=sP, sB=(T * C3, dh), sC=(T * C, dh), lens=lens, lim=(1, 0, 1), split_overwrite=True),
        "dV = P^T dO (TN, K=T)": lambda: K.gemm(P, dO, dqkv, T, dh, T, T, C, C3, False, False, c_off=2 * C, nb0=B, nb1=H, sA=sP, sB=(T * C, dh), sC=(T * C3, dh), lens=lens, lim=(1, 0, 1), split_overwrite=True),
        "softmax fwd": lambda: K.softmax_fwd(S, lens, B, H, T),
    }
    print(f"== {name}: sum len^2 = {v2}, {gf:.2f} GFLOP of valid work per product")
    for k, fn in prods.items():
        us = timeit(fn)
        print(f"   {k:24s} {us:7.1f} us" + ("" if "softmax" in k else f"   {gf / us * 1e3:6.1f} TFLOP/s valid"))


canon = [8 * s for s in CANONICAL_SRC_LENS]
v2 = sum(l * l for l in canon)
uni = int(round((v2 / B) ** 0.5 / 8)) * 8
run("canonical ragged", canon)
if "--only-canonical" in sys.argv:
    sys.exit(0)
run("dense", [T] * B)
run(f"uniform {uni}", [uni] * B)
run("uniform 512", [512] * B)
