"""Depthwise Conv1d k=31 of the conformer conv module (conformer.py:522-560) at the canonical decoder size: forward and both gradients."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctts_amd
from ctts_amd import ops
B, T, C = 16, 1000, 256
x = torch.randn(B, T, C, device="cuda", requires_grad=True)
w = (torch.randn(C, 1, 31, device="cuda") * 0.2).requires_grad_()
go = torch.randn(B, T, C, device="cuda")
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n * 1e3
with torch.no_grad():
    f = timeit(lambda: ops.depthwise_conv1d(x, w))
def fb():
    x.grad = None; w.grad = None
    ops.depthwise_conv1d(x, w).backward(go)
fbt = timeit(fb)
print(f"dwconv k=31 [{B},{T},{C}]: fwd {f:7.1f} us   fwd+bwd {fbt:7.1f} us   (fwd moves {2 * x.numel() * 4 / 1e6:.0f} MB: {2 * x.numel() * 4 / f / 1e6:.2f} TB/s)")
