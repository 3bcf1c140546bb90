"""Where do the ~100 us of the encoder-sized plane launches go?  (tools; GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctts_amd import kernels as K
from ctts_amd.synthetic import CANONICAL_SRC_LENS
dev = "cuda"
torch.manual_seed(0)


def timeit(fn, per_graph=10, reps=4):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(per_graph):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (reps * per_graph) * 1e3


def case(name, M, N, Kd, cin=None, T=None, lens=None, epi=False):
    x = torch.randn(M, cin or Kd, device=dev)
    w = torch.randn(N, Kd, device=dev) * 0.02
    out = torch.empty(M, N, device=dev)
    kw = {}
    if cin:
        kw["conv"] = (T, (Kd // cin) // 2, cin)
    if lens is not None:
        kw.update(row_lens=lens, row_T=T, row_halo=0)
    if epi:
        kw.update(bias=torch.zeros(N, device=dev), Z=torch.empty(M, N, device=dev), ldz=N, act=K.ACT_GELU, p_drop=0.1,
                  seed=torch.zeros(1, dtype=torch.int64, device=dev), drop_offset=1, alpha=9 ** -0.5)
    ap, bp = K.split_planes([x, w])
    args = (x, w, out, M, N, Kd, cin or Kd, Kd, N, True, True)
    took = K.gemm_takes_planes(*args, a_planes=ap, b_planes=bp, bf16_split=2, **kw)
    t = timeit(lambda: K.gemm(*args, a_planes=ap, b_planes=bp, bf16_split=2, **kw))
    print(f"{name:44s} took={int(took)}  {t:7.1f} us", flush=True)


lens = torch.tensor(list(CANONICAL_SRC_LENS), dtype=torch.int32, device=dev)
case("2048 x 1024 x 64 (2 K-blocks, whole tiles)", 2048, 1024, 64)
case("2048 x 1024 x 736 (23 K-blocks, whole tiles)", 2048, 1024, 736)
case("2048 x 1024 x 2304 dense plain", 2048, 1024, 2304)
case("2048 x 1024 x 2304 conv dense", 2048, 1024, 2304, cin=256, T=128)
case("2048 x 1024 x 2304 conv ragged", 2048, 1024, 2304, cin=256, T=128, lens=lens)
case("2048 x 1024 x 2304 conv ragged + epilogue", 2048, 1024, 2304, cin=256, T=128, lens=lens, epi=True)
case("2048 x 256 x 9216 plain", 2048, 256, 9216)
case("16384 x 1024 x 64 (2 K-blocks)", 16384, 1024, 64)
