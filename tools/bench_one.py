"""Single-shape GEMM loop for PMC collection: python tools/bench_one.py ffn1|dgrad|wgrad|qkv [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctts_amd import kernels as K, ops
dev = "cuda"
which = sys.argv[1] if len(sys.argv) > 1 else "ffn1"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
B, T = 16, 1024
M = B * T
if which == "ffn1":
    x = torch.randn(B, T, 256, device=dev); wf = torch.randn(1024, 2304, device=dev); C = torch.empty(B, T, 1024, device=dev)
    fn = lambda: K.gemm(x, wf, C, M, 1024, 2304, 256, 2304, 1024, True, True, conv=(T, 4, 256)); fl = 2 * M * 1024 * 2304
elif which == "ffn1_step":      # the dominant kernel exactly as the train step launches it (bench.py measure_dominant_kernel)
    from ctts_amd.synthetic import CANONICAL_SRC_LENS
    x = torch.randn(B, T, 256, device=dev); wf = torch.randn(1024, 2304, device=dev) * 0.02; C = torch.empty(B, T, 1024, device=dev)
    Z = torch.empty_like(C); bias = torch.zeros(1024, device=dev); seed = torch.zeros(1, dtype=torch.int64, device=dev)
    lens = torch.tensor([8 * v for v in CANONICAL_SRC_LENS], dtype=torch.int32, device=dev)
    kw = dict(conv=(T, 4, 256), alpha=9 ** -0.5, bias=bias, Z=Z, ldz=1024, act=K.ACT_GELU, p_drop=0.1, seed=seed, drop_offset=1,
              row_lens=lens, row_T=T, row_halo=0)
    planes = {}
    if K.BF16_SPLIT >= 1 and K.plane_shape_ok(M, 1024, 2304, 256):
        ap, bp = K.split_planes([x.view(M, 256), wf])
        if K.gemm_takes_planes(x, wf, C, M, 1024, 2304, 256, 2304, 1024, True, True, a_planes=ap, b_planes=bp, **kw):
            planes = dict(a_planes=ap, b_planes=bp)
    tmap = None if planes else K.row_tile_map(lens, T, 0, M)
    fn = lambda: K.gemm(x, wf, C, M, 1024, 2304, 256, 2304, 1024, True, True, tile_map=tmap, **kw, **planes)
    fl = 2 * int(lens.sum()) * 1024 * 2304
elif which == "wgrad_step":     # the FFN conv weight gradient as the train step launches it: ragged rows, planes of dZ and x at hand, += into the gradient
    from ctts_amd.synthetic import CANONICAL_SRC_LENS
    lens = torch.tensor([8 * v for v in CANONICAL_SRC_LENS], dtype=torch.int32, device=dev)
    mask = (torch.arange(T, device=dev)[None, :] < lens[:, None]).float().reshape(-1, 1)
    x = torch.randn(M, 256, device=dev); dz = torch.randn(M, 1024, device=dev) * mask; C = torch.zeros(1024, 2304, device=dev)
    kw = dict(conv=(T, 4, 256), conv_on_b=True, split_k=4, alpha=9 ** -0.5, row_lens=lens, row_T=T, row_halo=0)
    planes = {}
    if K.BF16_SPLIT >= 1 and K.plane_wgrad_shape_ok(1024, 2304, M, 256):
        ap, bp = K.split_planes([dz, x])
        if K.gemm_takes_planes(dz, x, C, 1024, 2304, M, 1024, 256, 2304, False, False, a_planes=ap, b_planes=bp, **kw):
            planes = dict(a_planes=ap, b_planes=bp)
    fn = lambda: K.gemm(dz, x, C, 1024, 2304, M, 1024, 256, 2304, False, False, **kw, **planes)
    fl = 2 * int(lens.sum()) * 1024 * 2304
elif which == "dgrad":
    dz = torch.randn(M, 1024, device=dev); wd = torch.randn(256, 9216, device=dev); C = torch.empty(B, T, 256, device=dev)
    fn = lambda: K.gemm(dz, wd, C, M, 256, 9216, 1024, 9216, 256, True, True, conv=(T, 4, 1024)); fl = 2 * M * 256 * 9216
elif which == "wgrad":
    x = torch.randn(B, T, 256, device=dev); dz = torch.randn(M, 1024, device=dev); C = torch.zeros(1024, 2304, device=dev)
    fn = lambda: K.gemm(dz, x, C, 1024, 2304, M, 1024, 256, 2304, False, False, conv=(T, 4, 256), conv_on_b=True, split_k=4); fl = 2 * M * 1024 * 2304
elif which in ("proj", "proj_sk2", "proj_nn", "proj_nn_sk2"):      # conformer-sized projections: M = B*T, N = K = 256 (latency-bound)
    sk = 2 if which.endswith("sk2") else 1
    x = torch.randn(M, 256, device=dev); C = torch.zeros(M, 256, device=dev)
    if "nn" in which:
        w = torch.randn(256, 256, device=dev)
        fn = lambda: K.gemm(x, w, C, M, 256, 256, 256, 256, 256, True, False, split_k=sk)
    else:
        w = torch.randn(256, 256, device=dev)
        fn = lambda: K.gemm(x, w, C, M, 256, 256, 256, 256, 256, True, True, split_k=sk)
    fl = 2 * M * 256 * 256
elif which == "sq":
    n = 4096
    x = torch.randn(n, n, device=dev); w = torch.randn(n, n, device=dev); C = torch.empty(n, n, device=dev)
    fn = lambda: K.gemm(x, w, C, n, n, n, n, n, n, True, True); fl = 2 * n ** 3
else:
    x = torch.randn(M, 256, device=dev); w = torch.randn(768, 256, device=dev); C = torch.empty(M, 768, device=dev)
    fn = lambda: K.gemm(x, w, C, M, 768, 256, 256, 256, 768, True, True); fl = 2 * M * 768 * 256
for _ in range(30):          # steady state: the first ~30 launches after an idle period run 15 % slower (clock ramp)
    fn()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters):
    fn()
e.record(); torch.cuda.synchronize()
t = s.elapsed_time(e) / iters * 1e-3
print(f"{which}: {t*1e6:.1f} us  {fl/t/1e12:.2f} TFLOP/s")
