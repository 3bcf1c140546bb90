"""Stream-K persistent GEMM (csrc/gemm_sk.hip) vs the tile-per-workgroup kernels (csrc/gemm.hip): same descriptor, both paths,
max-abs difference and time.  python tools/bench_sk.py [iters]   (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctts_amd import kernels as K
from ctts_amd.synthetic import CANONICAL_SRC_LENS

dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only = sys.argv[2] if len(sys.argv) > 2 else None
torch.manual_seed(0)


def timeit(fn, per_graph=10):
    """GPU time per call; the calls are replayed from a graph (a Python ctts_gemm call costs more CPU time than the small kernels run)"""
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


def sk_clock():
    """GHz of the shader clock during the last stream-K launch (CTTS_SK_DEBUG & 16), else None"""
    if not (int(os.environ.get("CTTS_SK_DEBUG", "0")) & 16):
        return ""
    ws = K.gemm_workspace(torch.device(dev))
    c = ws.view(torch.int64)[(2048 + 2) // 2:(2048 + 2) // 2 + 2].tolist()
    return f" clk {c[0] / max(c[1], 1) * 0.1:.3f} GHz" if c[1] else ""


def run(name, make, flops, outs):
    """make(use_sk) -> callable; outs() -> list of output tensors to compare"""
    if only and only not in name:
        return
    f0, f1 = make(False), make(True)
    for o in outs():
        o.zero_()
    f0(); torch.cuda.synchronize()
    ref = [o.clone() for o in outs()]
    for o in outs():
        o.zero_()
    f1(); torch.cuda.synchronize()
    diff = max(float((a - b).abs().max()) for a, b in zip(ref, outs()))
    scale = max(float(a.abs().max()) for a in ref)
    t0, t1 = timeit(f0), timeit(f1)
    print(f"{name:34s} old {t0*1e6:8.1f} us {flops/t0/1e12:7.2f} TF | sk {t1*1e6:8.1f} us {flops/t1/1e12:7.2f} TF | x{t0/t1:5.2f} | "
          f"maxdiff {diff:.2e} (scale {scale:.2e}) err {err_word()}{sk_clock()}", flush=True)


B, T = 16, 1024
M = B * T
lens = torch.tensor([8 * v for v in CANONICAL_SRC_LENS], dtype=torch.int32, device=dev)
nvalid = int(lens.sum())
tmap0 = K.row_tile_map(lens, T, 0, M)
tmap4 = K.row_tile_map(lens, T, 4, M)

# ---- decoder FFN conv forward (NT, conv on A), dense rows
x = torch.randn(B, T, 256, device=dev); wf = torch.randn(1024, 2304, device=dev) * 0.02; C = torch.empty(B, T, 1024, device=dev)
run("ffn1 fwd dense", lambda sk: (lambda: K.gemm(x, wf, C, M, 1024, 2304, 256, 2304, 1024, True, True, conv=(T, 4, 256), use_sk=sk)),
    2 * M * 1024 * 2304, lambda: [C])
# ---- the same exactly as the train step launches it
Z = torch.empty_like(C); bias = torch.randn(1024, device=dev) * 0.1; seed = torch.zeros(1, dtype=torch.int64, device=dev)
run("ffn1 fwd step (ragged, epilogue)", lambda sk: (lambda: K.gemm(
    x, wf, C, M, 1024, 2304, 256, 2304, 1024, True, True, conv=(T, 4, 256), alpha=9 ** -0.5, bias=bias, Z=Z, ldz=1024, act=K.ACT_GELU,
    p_drop=0.1, seed=seed, drop_offset=1, row_lens=lens, row_T=T, row_halo=0, tile_map=tmap0, use_sk=sk)),
    2 * nvalid * 1024 * 2304, lambda: [C, Z])
# ---- data gradient (NT conv, K = 9216, halo 4)
dz = torch.randn(M, 1024, device=dev); wd = torch.randn(256, 9216, device=dev) * 0.02; dX = torch.empty(B, T, 256, device=dev)
run("ffn1 dgrad dense", lambda sk: (lambda: K.gemm(dz, wd, dX, M, 256, 9216, 1024, 9216, 256, True, True, conv=(T, 4, 1024), use_sk=sk)),
    2 * M * 256 * 9216, lambda: [dX])
dXz = torch.zeros(B, T, 256, device=dev)


def dgrad_step(sk):
    if sk:
        return lambda: K.gemm(dz, wd, dX, M, 256, 9216, 1024, 9216, 256, True, True, conv=(T, 4, 1024), alpha=1 / 3, row_lens=lens, row_T=T,
                              row_halo=4, tile_map=tmap4, use_sk=True)
    def f():
        dX.zero_()
        K.gemm(dz, wd, dX, M, 256, 9216, 1024, 9216, 256, True, True, conv=(T, 4, 1024), alpha=1 / 3, row_lens=lens, row_T=T, row_halo=4,
               split_k=3, tile_map=tmap4, use_sk=False)
    return f


run("ffn1 dgrad step (ragged)", dgrad_step, 2 * nvalid * 256 * 9216, lambda: [dX])
# ---- weight gradient (TN, conv on B, accumulate)
dW = torch.zeros(1024, 2304, device=dev)


def wgrad(sk, ragged):
    kw = dict(row_lens=lens, row_T=T, tile_map=tmap0) if ragged else {}
    def f():
        dW.zero_()
        K.gemm(dz, x, dW, 1024, 2304, M, 1024, 256, 2304, False, False, conv=(T, 4, 256), conv_on_b=True, split_k=4, alpha=0.5, use_sk=sk, **kw)
    return f


# the dZ of padded rows is zero in the step; emulate for the ragged comparison
rowmask = (torch.arange(T, device=dev)[None, :] < lens[:, None]).reshape(M, 1).float()
run("ffn1 wgrad dense", lambda sk: wgrad(sk, False), 2 * M * 1024 * 2304, lambda: [dW])
dz.mul_(rowmask)
run("ffn1 wgrad step (ragged)", lambda sk: wgrad(sk, True), 2 * nvalid * 1024 * 2304, lambda: [dW])
dz = torch.randn(M, 1024, device=dev)

# ---- PostNet Conv1d(512 -> 512, k = 5): dense rows (BatchNorm needs the padded rows), forward / data gradient and weight gradient
xp = torch.randn(B, T, 512, device=dev); wp = torch.randn(512, 2560, device=dev) * 0.02; Cp = torch.empty(B, T, 512, device=dev)
run("postnet conv fwd", lambda sk: (lambda: K.gemm(xp, wp, Cp, M, 512, 2560, 512, 2560, 512, True, True, conv=(T, 2, 512), use_sk=sk)),
    2 * M * 512 * 2560, lambda: [Cp])
dzp = torch.randn(M, 512, device=dev); dWp = torch.zeros(512, 2560, device=dev)


def wgrad_post(sk):
    def f():
        dWp.zero_()
        K.gemm(dzp, xp, dWp, 512, 2560, M, 512, 512, 2560, False, False, conv=(T, 2, 512), conv_on_b=True, split_k=7, use_sk=sk)
    return f


run("postnet conv wgrad", wgrad_post, 2 * M * 512 * 2560, lambda: [dWp])
# ---- encoder FFN conv (16 x 128 phoneme rows, ragged)
Te = 128; Me = B * Te
lens_e = torch.tensor(list(CANONICAL_SRC_LENS), dtype=torch.int32, device=dev)
tmap_e = K.row_tile_map(lens_e, Te, 0, Me)
xe = torch.randn(B, Te, 256, device=dev); Ce = torch.empty(B, Te, 1024, device=dev)
run("encoder ffn1 fwd (2048 rows)", lambda sk: (lambda: K.gemm(xe, wf, Ce, Me, 1024, 2304, 256, 2304, 1024, True, True, conv=(Te, 4, 256),
                                                                 row_lens=lens_e, row_T=Te, row_halo=0, tile_map=tmap_e, use_sk=sk)),
    2 * int(lens_e.sum()) * 1024 * 2304, lambda: [Ce])

# ---- linears
for name, m, n, k in [("qkv NT 256->768", M, 768, 256), ("ffn2 NT 1024->256", M, 256, 1024), ("conf FF1 NT 256->1024", 16000, 1024, 256),
                      ("conf FF2 NT 1024->256", 16000, 256, 1024), ("proj NT 256->256", 16000, 256, 256), ("mel NT 256->80", M, 80, 256),
                      ("sq 4096", 4096, 4096, 4096)]:
    A = torch.randn(m, k, device=dev); Bm = torch.randn(n, k, device=dev) * 0.05; Cc = torch.empty(m, n, device=dev)
    run(name, lambda sk, A=A, Bm=Bm, Cc=Cc, m=m, n=n, k=k: (lambda: K.gemm(A, Bm, Cc, m, n, k, k, k, n, True, True, use_sk=sk)),
        2 * m * n * k, lambda Cc=Cc: [Cc])
# NN (linear data gradient) and TN (linear weight gradient)
dy = torch.randn(M, 256, device=dev); w2 = torch.randn(256, 1024, device=dev) * 0.05; dx = torch.empty(M, 1024, device=dev)
run("ffn2 dgrad NN", lambda sk: (lambda: K.gemm(dy, w2, dx, M, 1024, 256, 256, 1024, 1024, True, False, use_sk=sk)), 2 * M * 1024 * 256, lambda: [dx])
gI = torch.randn(M, 1024, device=dev); dw2 = torch.zeros(256, 1024, device=dev)


def wlin(sk):
    def f():
        dw2.zero_()
        K.gemm(dy, gI, dw2, 256, 1024, M, 256, 1024, 1024, False, False, split_k=8, use_sk=sk)
    return f


run("ffn2 wgrad TN", wlin, 2 * M * 1024 * 256, lambda: [dw2])
# ragged M (not a multiple of 64) with rowscale / residual epilogue
m, n, k = 15000, 256, 512
A = torch.randn(m, k, device=dev); Bm = torch.randn(n, k, device=dev) * 0.05; Cc = torch.empty(m, n, device=dev)
R = torch.randn(m, n, device=dev); rs = (torch.rand(m, device=dev) > 0.2).float(); bb = torch.randn(n, device=dev)
run("epilogue R/rowscale/bias, M=15000", lambda sk: (lambda: K.gemm(A, Bm, Cc, m, n, k, k, k, n, True, True, bias=bb, R=R, ldr=n, rowscale=rs,
                                                                  act=K.ACT_RELU, use_sk=sk)), 2 * m * n * k, lambda: [Cc])

# ---- under-filled launches: 2048 phoneme rows x 256 outputs = 128 tiles of 64 x 64 on 256 CUs
nve = int(lens_e.sum())
for Kd, name in ((1024, "enc ffn2 fwd bdrs"), (1024, "enc dgrad NN"), (768, "enc qkv dgrad NN"), (512, "enc NN K=512"), (256, "enc NN K=256")):
    a_ = torch.randn(Me, Kd, device=dev); c_ = torch.empty(Me, 256, device=dev)
    if "bdrs" in name:
        w_ = torch.randn(256, Kd, device=dev) * 0.03; b_ = torch.randn(256, device=dev); r_ = torch.randn(Me, 256, device=dev)
        rs_ = (torch.arange(Te, device=dev)[None, :] < lens_e[:, None]).reshape(Me).float(); sd_ = torch.zeros(1, dtype=torch.int64, device=dev)
        run(f"{name} 2048x256x{Kd}", lambda sk, a_=a_, w_=w_, c_=c_, b_=b_, r_=r_, rs_=rs_, sd_=sd_, Kd=Kd: (lambda: K.gemm(
            a_, w_, c_, Me, 256, Kd, Kd, Kd, 256, True, True, bias=b_, p_drop=0.1, seed=sd_, drop_offset=9, R=r_, ldr=256, rowscale=rs_,
            row_lens=lens_e, row_T=Te, row_halo=0, tile_map=tmap_e, use_sk=sk)), 2 * nve * 256 * Kd, lambda c_=c_: [c_])
    else:
        w_ = torch.randn(Kd, 256, device=dev) * 0.03
        run(f"{name} 2048x256x{Kd}", lambda sk, a_=a_, w_=w_, c_=c_, Kd=Kd: (lambda: K.gemm(
            a_, w_, c_, Me, 256, Kd, Kd, 256, 256, True, False, row_lens=lens_e, row_T=Te, row_halo=0, tile_map=tmap_e, use_sk=sk)),
            2 * nve * 256 * Kd, lambda c_=c_: [c_])
xc_ = torch.randn(B, Te, 256, device=dev); wc_ = torch.randn(256, 1280, device=dev) * 0.03; cc_ = torch.empty(B, Te, 256, device=dev)
zc_ = torch.empty(B, Te, 256, device=dev); bc_ = torch.randn(256, device=dev)
run("enc predictor conv k=5 ba1z 2048x256x1280", lambda sk: (lambda: K.gemm(
    xc_, wc_, cc_, Me, 256, 1280, 256, 1280, 256, True, True, conv=(Te, 2, 256), bias=bc_, Z=zc_, ldz=256, act=K.ACT_RELU, use_sk=sk)),
    2 * Me * 256 * 1280, lambda: [cc_, zc_])
