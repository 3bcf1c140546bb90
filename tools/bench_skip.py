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

# how much of the ragged-batch time is the skipping mechanism itself?  dense GEMM over exactly the rows the skipping version executes
rows_exec = int(sum((int(l) + 63) // 64 * 64 for l in lens.tolist()))
xd = torch.randn(1, rows_exec, cin, device=dev); outd = torch.empty(1, rows_exec, cout, device=dev); Zd = torch.empty_like(outd)
for name, ep in [("dense-compact bare", False), ("dense-compact epilogue", True)]:
    kw = dict(conv=(rows_exec, ks // 2, cin))
    if ep:
        kw.update(alpha=ks ** -0.5, bias=bias, Z=Zd, ldz=cout, act=K.ACT_GELU, p_drop=0.1, seed=seed, drop_offset=1)
    us = t(lambda: K.gemm(xd, wf, outd, rows_exec, cout, ks * cin, cin, ks * cin, cout, True, True, **kw))
    print(f"{name:24s} {us:8.1f} us   rows executed {rows_exec}  -> {2 * rows_exec * cout * ks * cin / us / 1e6:7.1f} TFLOP/s dense")
for name, ep, rl in [("skip epilogue (again)", True, True), ("dense epilogue (again)", True, False)]:
    kw = dict(conv=(T, ks // 2, cin), alpha=ks ** -0.5, bias=bias, Z=Z, ldz=cout, act=K.ACT_GELU, p_drop=0.1, seed=seed, drop_offset=1)
    if rl:
        kw.update(row_lens=lens, row_T=T, row_halo=0)
    us = t(lambda: K.gemm(x, wf, out, M, cout, ks * cin, cin, ks * cin, cout, True, True, **kw), iters=50)
    print(f"{name:24s} {us:8.1f} us")

tmap = K.row_tile_map(lens, T, 0, M)
kw = dict(conv=(T, ks // 2, cin), alpha=ks ** -0.5, bias=bias, Z=Z, ldz=cout, act=K.ACT_GELU, p_drop=0.1, seed=seed, drop_offset=1,
          row_lens=lens, row_T=T, row_halo=0)
out2 = torch.empty_like(out)
K.gemm(x, wf, out, M, cout, ks * cin, cin, ks * cin, cout, True, True, **kw)
K.gemm(x, wf, out2, M, cout, ks * cin, cin, ks * cin, cout, True, True, tile_map=tmap, **kw)
print("tile-map result identical:", bool(torch.equal(out, out2)), "active tiles", int(tmap[0]), "of", tmap.numel() - 1)
for name, tm_ in [("skip epilogue", None), ("skip+schedule epilogue", tmap), ("skip epilogue", None), ("skip+schedule epilogue", tmap)]:
    us = t(lambda: K.gemm(x, wf, out, M, cout, ks * cin, cin, ks * cin, cout, True, True, tile_map=tm_, **kw), iters=50)
    print(f"{name:24s} {us:8.1f} us   {2 * valid * cout * ks * cin / us / 1e6:7.1f} TFLOP/s algorithmic")
