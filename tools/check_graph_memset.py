"""Regression check (GPU box): zero-initialisation inside a captured hipGraph.  With hipMemsetAsync in the C entry points the third float
of the accumulator kept stale data from the second replay on (ROCm 7.2 memset graph nodes); ctts_zero_async (a kernel) must give exact
results on every replay.  python tools/check_graph_memset.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctts_amd
from ctts_amd import kernels as K
dev = "cuda"
for Cc in (3, 11, 64):
    x = torch.randn(512, Cc, device=dev)
    K.colsum(x); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        res = K.colsum(x)
    for it in range(3):
        x.normal_()
        g.replay()
        torch.cuda.synchronize()
        print("C", Cc, "replay", it, "res", res[:3].tolist(), "want", x.sum(0)[:3].tolist())
# same with a torch-zeroed accumulate target
x = torch.randn(512, 64, device=dev)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    buf = torch.zeros(64, device=dev)
    K.colsum(x, acc_into=buf)
for it in range(3):
    x.normal_(); g.replay(); torch.cuda.synchronize()
    print("torch-zero replay", it, float((buf - x.sum(0)).abs().max()))
