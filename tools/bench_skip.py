"""Micro-benchmark: FFN conv k9 forward (dominant kernel) dense vs padded-row skipping, with / without the fused epilogue."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctts_amd
from ctts_amd import kernels as K
from ctts_amd.synthetic import CANONICAL_SRC_LENS

dev = "cuda"
B, T, cin, cout, ks = 16, 1024, 256, 1024, 9
M = B * T
x = torch.randn(B, T, cin, device=dev); wf = torch.randn(cout, ks * cin, device=dev) * 0.02
bias = torch.zeros(cout, device=dev); out = torch.empty(B, T, cout, device=dev); Z = torch.empty_like(out)
seed = torch.zeros(1, dtype=torch.int64, device=dev)
lens = torch.tensor([8 * s for s in CANONICAL_SRC_LENS], dtype=torch.int32, device=dev)
valid = int(lens.sum())


def t(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, ep, rl in [("dense bare", False, False), ("dense epilogue", True, False), ("skip bare", False, True), ("skip epilogue", True, True)]:
    kw = dict(conv=(T, ks // 2, cin))
    if ep:
        kw.update(alpha=ks ** -0.5, bias=bias, Z=Z, ldz=cout, act=K.ACT_GELU, p_drop=0.1, seed=seed, drop_offset=1)
    if rl:
        kw.update(row_lens=lens, row_T=T, row_halo=0)
    us = t(lambda: K.gemm(x, wf, out, M, cout, ks * cin, cin, ks * cin, cout, True, True, **kw))
    rows = valid if rl else M
    print(f"{name:16s} {us:8.1f} us   {2 * rows * cout * ks * cin / us / 1e6:7.1f} TFLOP/s (rows credited: {rows})")
