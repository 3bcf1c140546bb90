"""Micro-benchmark: fused flat clip+Adam (csrc/optim.hip) vs torch clip_grad_norm_(foreach) + fused Adam on the fs2 parameter set."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctts_amd
from ctts_amd.configs import get_configs
from ctts_amd.dp import FlatGradArena, FlatAdam

dev = "cuda"
pre, mc, tc = get_configs()


def t(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


m = ctts_amd.CompTransTTS(pre, mc, tc).to(dev)
arena = FlatGradArena(m.parameters())
arena.flat.normal_()
opt = torch.optim.Adam(arena.params, lr=torch.tensor(1e-4, device=dev), betas=(0.9, 0.98), eps=1e-9, capturable=True, fused=True)


def torch_step():
    torch.nn.utils.clip_grad_norm_(arena.params, 1.0, foreach=True)
    opt.step()


print(f"torch clip+Adam (eager)  {t(torch_step):8.1f} us")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    torch_step()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    torch_step()
print(f"torch clip+Adam (graph)  {t(g.replay):8.1f} us")
fa = FlatAdam(arena, 1e-4, betas=(0.9, 0.98), eps=1e-9, max_norm=1.0)
print(f"fused flat clip+Adam     {t(fa.step):8.1f} us   ({arena.flat.numel()} params, {arena.flat.numel() * 32 / 1e6:.0f} MB of traffic)")
