"""Plane kernel (csrc/gemm_pl.hip: pre-split bf16 operands, persistent, LDS-DMA) against gemm_x6_kernel (split inside the GEMM) and the
fp32-MFMA kernels on the big NT launches of the fs2 step, graph-timed; split launches timed separately.
    python tools/bench_pl.py [iters] [name-filter]        (GPU box)
Prints one line per shape: us and dense-equivalent TFLOP/s per path, the max |difference| between the paths, the hand-off error word."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctts_amd import kernels as K
from ctts_amd.synthetic import CANONICAL_SRC_LENS

dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
only = sys.argv[2] if len(sys.argv) > 2 else None
torch.manual_seed(0)


def timeit(fn, per_graph=10):
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
    reps = max(1, iters // per_graph)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (reps * per_graph) * 1e-3


def err_word():
    return max(int(ws.view(torch.int32)[2048].item()) for ws in K._SK_WS.values())


def pl_clock():
    """GHz of the shader clock during the last plane-kernel launch (CTTS_PL_DEBUG & 16)"""
    if not (int(os.environ.get("CTTS_PL_DEBUG", "0")) & 16):
        return ""
    ws = K.gemm_workspace(torch.device(dev))
    c = ws.view(torch.int64)[(2048 + 2) // 2:(2048 + 2) // 2 + 2].tolist()
    return f" clk {c[0] / max(c[1], 1) * 0.1:.3f} GHz ({c[0]} cycles)" if c[1] else ""


def run(name, A, Bm, out_shape, M, N, Kd, lda, flops, kw, tmap=None):
    if only and only not in name:
        return
    A2 = A.reshape(-1, lda)
    ap, bp = K.split_planes([A2, Bm])
    outs = {}

    def make(mode):
        C = torch.full(out_shape, float("nan"), device=dev)
        k2 = dict(kw)
        if "Z" in k2:
            k2["Z"] = torch.empty(out_shape, device=dev)
        if mode == "pl":
            k2.update(a_planes=ap, b_planes=bp, bf16_split=1)
            assert K.gemm_takes_planes(A, Bm, C, M, N, Kd, lda, Kd, N, True, True, **k2), name
        elif mode == "x6":
            k2.update(bf16_split=1, tile_map=tmap)
        else:
            k2.update(bf16_split=0, tile_map=tmap)
        outs[mode] = C
        return lambda: K.gemm(A, Bm, C, M, N, Kd, lda, Kd, N, True, True, **k2)
    fns = {m: make(m) for m in ("f32", "x6", "pl")}
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    d_x6 = float((outs["pl"] - outs["x6"]).abs().max())
    d_32 = float((outs["pl"] - outs["f32"]).abs().max())
    t = {m: timeit(f) for m, f in fns.items()}
    clk = pl_clock()
    t_sa = timeit(lambda: K.split_planes([A2]))
    t_sb = timeit(lambda: K.split_planes([Bm]))
    print(f"{name:26s} " + " | ".join(f"{m} {t[m]*1e6:7.1f} us {flops/t[m]/1e12:6.1f} TF" for m in ("f32", "x6", "pl")) +
          f" | split A {t_sa*1e6:5.1f} us B {t_sb*1e6:5.1f} us | pl-x6 {d_x6:.2e} pl-f32 {d_32:.2e} (|out| {float(outs['f32'].abs().max()):.2f}) err {err_word()}{clk}",
          flush=True)


B, T = 16, 1024
M = B * T
lens = torch.tensor([8 * v for v in CANONICAL_SRC_LENS], dtype=torch.int32, device=dev)
nvalid = int(lens.sum())
seed = torch.zeros(1, dtype=torch.int64, device=dev)

# decoder FFN conv forward exactly as the step launches it (ragged, bias + GELU + dropout + pre-activation store)
x = torch.randn(B, T, 256, device=dev)
wf = torch.randn(1024, 2304, device=dev) * 0.02
bias = torch.zeros(1024, device=dev)
run("ffn1 fwd step", x, wf, (B, T, 1024), M, 1024, 2304, 256, 2 * nvalid * 1024 * 2304,
    dict(conv=(T, 4, 256), alpha=9 ** -0.5, bias=bias, Z=True, ldz=1024, act=K.ACT_GELU, p_drop=0.1, seed=seed, drop_offset=1,
         row_lens=lens, row_T=T, row_halo=0), tmap=K.row_tile_map(lens, T, 0, M))
run("ffn1 fwd dense", x, wf, (B, T, 1024), M, 1024, 2304, 256, 2 * M * 1024 * 2304, dict(conv=(T, 4, 256)))
# its data gradient (N = 256, K = 9216, halo 4)
dz = torch.randn(B, T, 1024, device=dev) * (torch.arange(T, device=dev)[None, :, None] < lens[:, None, None])
wd = torch.randn(256, 9216, device=dev) * 0.02
run("ffn1 dgrad step", dz, wd, (B, T, 256), M, 256, 9216, 1024, 2 * nvalid * 256 * 9216,
    dict(conv=(T, 4, 1024), alpha=9 ** -0.5, row_lens=lens, row_T=T, row_halo=4, split_overwrite=True), tmap=K.row_tile_map(lens, T, 4, M))
# PostNet conv k=5 512 -> 512 (dense rows), forward and data gradient are the same shape
xp = torch.randn(B, T, 512, device=dev)
wp = torch.randn(512, 2560, device=dev) * 0.02
run("postnet conv", xp, wp, (B, T, 512), M, 512, 2560, 512, 2 * M * 512 * 2560, dict(conv=(T, 2, 512), bias=torch.zeros(512, device=dev)))
# FFN linear 2 forward (k = 1: N = 256, K = 1024; bias + dropout + residual + row scale) and a plain big square
h = torch.randn(M, 1024, device=dev)
w2 = torch.randn(256, 1024, device=dev) * 0.03
R = torch.randn(M, 256, device=dev)
rs = (torch.arange(T, device=dev)[None, :] < lens[:, None]).float().reshape(-1).contiguous()
run("ffn2 fwd step", h, w2, (M, 256), M, 256, 1024, 1024, 2 * nvalid * 256 * 1024,
    dict(bias=torch.zeros(256, device=dev), p_drop=0.1, seed=seed, drop_offset=2, R=R, ldr=256, rowscale=rs, row_lens=lens, row_T=T, row_halo=0),
    tmap=K.row_tile_map(lens, T, 0, M))
# FFN linear 2 data gradient with the producer's epilogue backward (N = 1024, K = 256: "a2zdB"); in the step it runs NN on the
# weight-stationary kernel (85 us) - here NT on a transposed weight, to see what the plane kernel would make of it
dy2 = torch.randn(M, 256, device=dev) * rs[:, None]
w2t = torch.randn(1024, 256, device=dev) * 0.03
Zp = torch.randn(M, 1024, device=dev)
run("ffn2 dgrad step (a2zdB)", dy2, w2t, (M, 1024), M, 1024, 256, 256, 2 * nvalid * 1024 * 256,
    dict(epi_bwd=True, Z=True, ldz=1024, act=K.ACT_GELU, p_drop=0.1, seed=seed, drop_offset=1, row_lens=lens, row_T=T, row_halo=0),
    tmap=K.row_tile_map(lens, T, 0, M))
xq = torch.randn(M, 256, device=dev)
wq = torch.randn(768, 256, device=dev) * 0.05
run("qkv fwd step", xq, wq, (M, 768), M, 768, 256, 256, 2 * nvalid * 768 * 256,
    dict(bias=torch.zeros(768, device=dev), row_lens=lens, row_T=T, row_halo=0), tmap=K.row_tile_map(lens, T, 0, M))
# encoder-sized launches (16 x 128 phoneme rows, about half of them valid): few tiles, every tile cut between many workgroups
Te = 128
lens_en = torch.tensor(list(CANONICAL_SRC_LENS), dtype=torch.int32, device=dev)
nval_e = int(lens_en.sum())
xe = torch.randn(B, Te, 256, device=dev)
run("enc ffn1 fwd step", xe, wf, (B, Te, 1024), B * Te, 1024, 2304, 256, 2 * nval_e * 1024 * 2304,
    dict(conv=(Te, 4, 256), alpha=9 ** -0.5, bias=bias, Z=True, ldz=1024, act=K.ACT_GELU, p_drop=0.1, seed=seed, drop_offset=1,
         row_lens=lens_en, row_T=Te, row_halo=0), tmap=K.row_tile_map(lens_en, Te, 0, B * Te))
dze = torch.randn(B, Te, 1024, device=dev) * (torch.arange(Te, device=dev)[None, :, None] < lens_en[:, None, None])
run("enc ffn1 dgrad step", dze, wd, (B, Te, 256), B * Te, 256, 9216, 1024, 2 * nval_e * 256 * 9216,
    dict(conv=(Te, 4, 1024), alpha=9 ** -0.5, row_lens=lens_en, row_T=Te, row_halo=4, split_overwrite=True), tmap=K.row_tile_map(lens_en, Te, 4, B * Te))
n = 4096
run("square 4096", torch.randn(n, n, device=dev), torch.randn(n, n, device=dev) * 0.02, (n, n), n, n, n, n, 2 * n ** 3, dict())


# ---- weight gradients (TN, csrc/gemm_plw.hip): the plane sets are the ones the forward / data-gradient launches made, so no split is timed
def run_tn(name, dZm, Xm, cout, cin, ksize, pad, T, flops, kw):
    if only and only not in name:
        return
    rows, Kd = dZm.shape[0], max(ksize, 1) * cin
    ap, bp = K.split_planes([dZm, Xm])
    conv = dict(conv=(T, pad, cin), conv_on_b=True) if ksize else {}
    outs = {}

    def make(mode):
        Cm = torch.full((cout, Kd), float("nan"), device=dev)
        k2 = dict(kw, **conv)
        if mode == "pl":
            k2.update(a_planes=ap, b_planes=bp, bf16_split=2)      # 2: also the shapes below the kernel's tile-count threshold (shown for the record)
            assert K.gemm_takes_planes(dZm, Xm, Cm, cout, Kd, rows, cout, cin, Kd, False, False, **k2), name
        else:
            k2.update(bf16_split=1 if mode == "x6" else 0)
        outs[mode] = Cm
        return lambda: K.gemm(dZm, Xm, Cm, cout, Kd, rows, cout, cin, Kd, False, False, **k2)
    fns = {m: make(m) for m in ("f32", "x6", "pl")}
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    d_x6 = float((outs["pl"] - outs["x6"]).abs().max())
    d_32 = float((outs["pl"] - outs["f32"]).abs().max())
    t = {m: timeit(f) for m, f in fns.items()}
    print(f"{name:26s} " + " | ".join(f"{m} {t[m]*1e6:7.1f} us {flops/t[m]/1e12:6.1f} TF" for m in ("f32", "x6", "pl")) +
          f" | pl-x6 {d_x6:.2e} pl-f32 {d_32:.2e} (|out| {float(outs['f32'].abs().max()):.2f}) err {err_word()}{pl_clock()}", flush=True)


mask = (torch.arange(T, device=dev)[None, :] < lens[:, None]).float().reshape(-1, 1)
dzw = torch.randn(M, 1024, device=dev) * mask
xw = torch.randn(M, 256, device=dev)
rlw = dict(row_lens=lens, row_T=T, row_halo=0)
run_tn("ffn1 wgrad step", dzw, xw, 1024, 256, 9, 4, T, 2 * nvalid * 1024 * 2304, dict(split_k=4, split_overwrite=True, **rlw))
run_tn("ffn1 wgrad dense", dzw, xw, 1024, 256, 9, 4, T, 2 * M * 1024 * 2304, dict(split_k=4, split_overwrite=True))
run_tn("postnet wgrad", torch.randn(M, 512, device=dev), torch.randn(M, 512, device=dev), 512, 512, 5, 2, T, 2 * M * 512 * 2560,
       dict(split_k=7, split_overwrite=True))
run_tn("k5 256 wgrad", torch.randn(M, 256, device=dev), torch.randn(M, 256, device=dev), 256, 256, 5, 2, T, 2 * M * 256 * 1280,
       dict(split_k=26, split_overwrite=True))
Me = 2048
lens_e = torch.tensor(list(CANONICAL_SRC_LENS), dtype=torch.int32, device=dev)
mask_e = (torch.arange(128, device=dev)[None, :] < lens_e[:, None]).float().reshape(-1, 1)
run_tn("enc ffn1 wgrad", torch.randn(Me, 1024, device=dev) * mask_e, torch.randn(Me, 256, device=dev), 1024, 256, 9, 4, 128,
       2 * int(lens_e.sum()) * 1024 * 2304, dict(split_k=4, split_overwrite=True, row_lens=lens_e, row_T=128, row_halo=0))
run_tn("ffn2 wgrad (k=1)", torch.randn(M, 256, device=dev) * mask, torch.randn(M, 1024, device=dev), 256, 1024, 0, 0, 0, 2 * nvalid * 256 * 1024,
       dict(split_k=32, split_overwrite=True, **rlw))
