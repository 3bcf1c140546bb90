"""Phase clocks of the weight-stationary GEMM (CTTS_WS_DEBUG=1): python tools/dbg_ws.py [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctts_amd import kernels as K
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
M = 16000
x = torch.randn(M, 256, device="cuda"); w = torch.randn(N, 256, device="cuda") * 0.05; C = torch.empty(M, N, device="cuda")
ws = K.gemm_workspace(torch.device("cuda"))
f = lambda: K.gemm(x, w, C, M, N, 256, 256, 256, N, True, True)
for _ in range(40):
    f()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(30):
        f()
g.replay(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
g.replay()
e.record(); torch.cuda.synchronize()
print(f"N={N} debug={os.environ.get('CTTS_WS_DEBUG')} {s.elapsed_time(e)/30*1e3:.1f} us")
if int(os.environ.get("CTTS_WS_DEBUG", "0")) & 1:
    t = ws.view(torch.int64)[-8192:].view(1024, 8)[:((N + 127) // 128) * 64].double().cpu()
    names = ["total", "prologue", "wait1", "wait2", "comp0", "comp1", "epilogue", "tiles"]
    print("mean per WG (cycles):", {n: round(float(v), 1) for n, v in zip(names, t.mean(0))})
    print("max  per WG (cycles):", {n: round(float(v), 1) for n, v in zip(names, t.max(0).values)})
    ws.view(torch.int64)[-8192:].zero_()
