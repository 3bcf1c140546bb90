"""Micro-benchmark of the GRU recurrence kernels (H = 32, T = 1024, B = 16, two sequences per utterance as in ops.gru_group)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctts_amd
from ctts_amd import kernels as K

dev = "cuda"
B, T, H, nd = 16, 1024, 32, 2
gi = torch.randn(B, T, nd * 3 * H, device=dev)
whh = torch.randn(nd, 3 * H, H, device=dev) * 0.1
bhh = torch.zeros(nd, 3 * H, device=dev)


def t(fn, iters=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


out, gates = K.gru_fwd(gi, whh, bhh, H, nd, rev_mask=0)
dout = torch.randn_like(out)
print(f"gru fwd {t(lambda: K.gru_fwd(gi, whh, bhh, H, nd, rev_mask=0)):8.1f} us   bwd {t(lambda: K.gru_bwd(dout, out, gates, whh, H, nd, 0)):8.1f} us"
      f"   ({'multi-wave' if os.environ.get('CTTS_GRU_MULTIWAVE') else 'single-wave'} kernels, {T} steps)")
