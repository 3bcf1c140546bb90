"""Weight-stationary K = 256 GEMM (csrc/gemm_ws.hip) vs the other kernels of ctts_gemm on the same descriptor: bit difference of every
output and time.  python tools/bench_ws.py [iters] [name-filter]   (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctts_amd import kernels as K
from ctts_amd.synthetic import CANONICAL_SRC_LENS

dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
only = sys.argv[2] if len(sys.argv) > 2 else None
torch.manual_seed(0)


def timeit(fn, per_graph=20):
    """GPU time per call: the calls are captured into a graph (a Python ctts_gemm call costs more CPU time than these kernels run)"""
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(per_graph):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    reps = max(1, iters // per_graph)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (reps * per_graph) * 1e-3


def run(name, fn, flops, outs):
    if only and only not in name:
        return
    res = []
    for on in (False, True):
        K.gemm_ws_enable(on)
        for o in outs:
            o.fill_(float("nan"))
        fn(); torch.cuda.synchronize()
        res.append([o.clone() for o in outs])
    diff = max(float((a - b).abs().max()) if not (torch.isnan(a).any() or torch.isnan(b).any()) else float("nan")
               for a, b in zip(*res))
    scale = max(float(a.abs().max()) for a in res[0])
    t = []
    for on in (False, True):
        K.gemm_ws_enable(on)
        t.append(timeit(fn))
    K.gemm_ws_enable(True)
    print(f"{name:40s} other {t[0]*1e6:8.1f} us {flops/t[0]/1e12:7.2f} TF | ws {t[1]*1e6:8.1f} us {flops/t[1]/1e12:7.2f} TF | x{t[0]/t[1]:5.2f} | "
          f"maxdiff {diff:.2e} (scale {scale:.2e})", flush=True)


B, T = 16, 1000
M = B * T
lens = torch.tensor([min(T, int(7.8 * v)) for v in CANONICAL_SRC_LENS], dtype=torch.int32, device=dev)
nvalid = int(lens.sum())
tmap = K.row_tile_map(lens, T, 0, M)
x = torch.randn(M, 256, device=dev)
seed = torch.zeros(1, dtype=torch.int64, device=dev)

for N in (256, 512, 768, 1024):
    w = torch.randn(N, 256, device=dev) * 0.05          # [out, in]: NT
    wt = w.t().contiguous()                              # [in, out]: NN
    bias = torch.randn(N, device=dev) * 0.1
    C = torch.empty(M, N, device=dev); Z = torch.empty(M, N, device=dev); R = torch.randn(M, N, device=dev)
    fl = 2 * M * N * 256
    run(f"NT plain            N={N}", lambda: K.gemm(x, w, C, M, N, 256, 256, 256, N, True, True), fl, [C])
    run(f"NN plain            N={N}", lambda: K.gemm(x, wt, C, M, N, 256, 256, N, N, True, False), fl, [C])
    run(f"NT bias+swish+Z+drop N={N}", lambda: K.gemm(x, w, C, M, N, 256, 256, 256, N, True, True, bias=bias, Z=Z, ldz=N,
                                                       act=K.ACT_SWISH, p_drop=0.1, seed=seed, drop_offset=3), fl, [C, Z])
    run(f"NT bias+drop+R ragged N={N}", lambda: K.gemm(x, w, C, M, N, 256, 256, 256, N, True, True, bias=bias, alpha=0.5, p_drop=0.1,
                                                        seed=seed, drop_offset=5, R=R, ldr=N, row_lens=lens, row_T=T, row_halo=0,
                                                        tile_map=tmap), 2 * nvalid * N * 256, [C])
    run(f"NN epi_bwd gelu     N={N}", lambda: K.gemm(x, wt, C, M, N, 256, 256, N, N, True, False, Z=R, ldz=N, act=K.ACT_GELU,
                                                      p_drop=0.1, seed=seed, drop_offset=7, epi_bwd=True), fl, [C])

# edges: N not a multiple of 128 / 32, M not a multiple of 64, strided C
for (m, n) in ((4100, 80), (5003, 200), (16000, 336)):
    w = torch.randn(n, 256, device=dev) * 0.05
    xx = torch.randn(m, 256, device=dev)
    Cb = torch.empty(m, n + 8, device=dev)
    run(f"NT edge M={m} N={n} ldc={n + 8}", lambda: K.gemm(xx, w, Cb, m, n, 256, 256, 256, n + 8, True, True), 2 * m * n * 256, [Cb[:, :n]])
