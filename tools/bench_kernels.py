"""Micro-benchmarks of the hot GEMM / conv shapes at the canonical batch (B=16, Tm=1024).
Usage (GPU box): python tools/bench_kernels.py   -> TFLOP/s per shape, fp32 MFMA roof = 157.3."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctts_amd
from ctts_amd import kernels as K, ops

dev = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    B, T = 16, 1024
    M = B * T
    rows = []
    shapes = [("qkv   NT", M, 768, 256), ("oproj NT", M, 256, 256), ("ffn2  NT", M, 256, 1024), ("mel   NT", M, 80, 256)]
    for name, m, n, k in shapes:
        A, Bm, C = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev), torch.empty(m, n, device=dev)
        t = timeit(lambda: K.gemm(A, Bm, C, m, n, k, k, k, n, True, True))
        rows.append((name, m, n, k, t, 2 * m * n * k / t / 1e12))
    for name, cin, cout, ks in [("ffn1 conv fwd", 256, 1024, 9), ("post conv fwd", 512, 512, 5), ("post0 conv fwd", 80, 512, 5)]:
        x = torch.randn(B, T, cin, device=dev)
        wf = torch.randn(cout, ks * cin, device=dev)
        C = torch.empty(B, T, cout, device=dev)
        kd = ks * cin
        t = timeit(lambda: K.gemm(x, wf, C, M, cout, kd, cin, kd, cout, True, True, conv=(T, ks // 2, cin)))
        rows.append((name, M, cout, kd, t, 2 * M * cout * kd / t / 1e12))
    # dgrad (NN) and wgrad (TN split-K) of ffn2 and the ffn1 conv
    dz = torch.randn(M, 256, device=dev); w2 = torch.randn(256, 1024, device=dev); dx = torch.empty(M, 1024, device=dev)
    t = timeit(lambda: K.gemm(dz, w2, dx, M, 1024, 256, 256, 1024, 1024, True, False))
    rows.append(("ffn2 dgrad NN", M, 1024, 256, t, 2 * M * 1024 * 256 / t / 1e12))
    g = torch.randn(M, 1024, device=dev); dw = torch.zeros(256, 1024, device=dev)
    sk = max(2, ops._split_k_for(256, 1024, M))
    t = timeit(lambda: K.gemm(dz, g, dw, 256, 1024, M, 256, 1024, 1024, False, False, split_k=sk))
    rows.append((f"ffn2 wgrad TN sk{sk}", 256, 1024, M, t, 2 * M * 1024 * 256 / t / 1e12))
    x = torch.randn(B, T, 256, device=dev); dzc = torch.randn(M, 1024, device=dev); dwf = torch.zeros(1024, 2304, device=dev)
    sk = max(2, ops._split_k_for(1024, 2304, M))
    t = timeit(lambda: K.gemm(dzc, x, dwf, 1024, 2304, M, 1024, 256, 2304, False, False, conv=(T, 4, 256), conv_on_b=True, split_k=sk))
    rows.append((f"ffn1 wgrad TN sk{sk}", 1024, 2304, M, t, 2 * M * 1024 * 2304 / t / 1e12))
    wd = torch.randn(256, 9 * 1024, device=dev); dxc = torch.empty(B, T, 256, device=dev)
    t = timeit(lambda: K.gemm(dzc, wd, dxc, M, 256, 9 * 1024, 1024, 9 * 1024, 256, True, True, conv=(T, 4, 1024)))
    rows.append(("ffn1 dgrad conv", M, 256, 9216, t, 2 * M * 256 * 9216 / t / 1e12))
    # attention pieces
    H, dh, C3 = 2, 128, 768
    qkv = torch.randn(B, T, C3, device=dev); S = torch.empty(B, H, T, T, device=dev)
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    t = timeit(lambda: K.gemm(qkv, qkv, S, T, T, dh, C3, C3, T, True, True, b_off=256, nb0=B, nb1=H, sA=(T * C3, dh), sB=(T * C3, dh),
                              sC=(H * T * T, T * T), lens=lens, lim=(1, 1, 0)))
    rows.append(("QK^T batched", T, T, dh, t, 2 * B * H * T * T * dh / t / 1e12))
    t = timeit(lambda: K.softmax_fwd(S, lens, B, H, T))
    rows.append(("softmax (GB/s)", B * H * T, T, 0, t, 2 * S.numel() * 4 / t / 1e9))
    out = torch.zeros(B, T, 256, device=dev)
    t = timeit(lambda: K.gemm(S, qkv, out, T, dh, T, T, C3, 256, True, False, b_off=512, nb0=B, nb1=H, sA=(H * T * T, T * T),
                              sB=(T * C3, dh), sC=(T * 256, dh), lens=lens, lim=(1, 0, 1)))
    rows.append(("PV batched", T, dh, T, t, 2 * B * H * T * T * dh / t / 1e12))
    xx = torch.randn(M, 256, device=dev); gm = torch.ones(256, device=dev); bt = torch.zeros(256, device=dev)
    t = timeit(lambda: K.layernorm_fwd(xx, gm, bt, 1e-5))
    rows.append(("layernorm fwd (GB/s)", M, 256, 0, t, 2 * xx.numel() * 4 / t / 1e9))
    for r in rows:
        print(f"{r[0]:24s} M={r[1]:6d} N={r[2]:5d} K={r[3]:5d}  {r[4]*1e6:9.1f} us  {r[5]:8.2f}")


if __name__ == "__main__":
    main()
