"""LayerNorm forward / backward at the decoder's activation size [16384, 256] (and [16000, 256], dres + dropout as the step uses them)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctts_amd
from ctts_amd import kernels as K
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n * 1e3
for rows, C in ((16384, 256), (16000, 256), (16384, 1024)):
    x = torch.randn(rows, C, device="cuda"); dy = torch.randn(rows, C, device="cuda"); dres = torch.randn(rows, C, device="cuda")
    g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    y, mean, rstd = K.layernorm_fwd(x, g, b, 1e-5)[:3]
    f = timeit(lambda: K.layernorm_fwd(x, g, b, 1e-5))
    bw = timeit(lambda: K.layernorm_bwd(dy, x, g, mean, rstd))
    bw2 = timeit(lambda: K.layernorm_bwd(dy, x, g, mean, rstd, dres=dres))
    mb = x.numel() * 4 / 1e6
    print(f"layernorm [{rows},{C}]: fwd {f:6.1f} us ({2 * mb / f / 1e6 * 1e6 / 1e6:.2f} TB/s)   bwd {bw:6.1f} us ({3 * mb / bw:.2f} MB/us = TB/s)   bwd+dres {bw2:6.1f} us ({4 * mb / bw2:.2f} TB/s)")
