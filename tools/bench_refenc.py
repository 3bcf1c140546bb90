"""Micro-benchmark of one liu2021 ReferenceEncoder at the canonical batch: per-layer Conv2d (patch matrix + GEMM),
BatchNorm2d+ReLU, and the GRU recurrence, forward and backward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctts_amd
from ctts_amd import ops, kernels as K

dev = "cuda"
B, T = 16, 1024


def t(fn, iters=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


filters = [4, 32, 32, 64, 64, 128, 128]
W = 80
tot_f = tot_b = 0.0
for i in range(6):
    cin, cout = filters[i], filters[i + 1]
    Wo = (W - 1) // 2 + 1
    x = torch.randn(B, T, W, cin, device=dev, requires_grad=(i > 0))
    w = (torch.randn(cout, cin, 3, 3, device=dev) * 0.1).requires_grad_(True)
    b = torch.zeros(cout, device=dev, requires_grad=True)
    gam, bet = torch.ones(cout, device=dev, requires_grad=True), torch.zeros(cout, device=dev, requires_grad=True)
    rm, rv, nb = torch.zeros(cout, device=dev), torch.ones(cout, device=dev), torch.tensor(0, device=dev)
    tf_conv = t(lambda: ops.conv2d_3x3s2(x.detach(), w.detach(), b.detach()))
    tf_im2col = t(lambda: K.im2col_3x3s2(x.detach()))
    y = ops.conv2d_3x3s2(x, w, b)
    tf_bn = t(lambda: ops.batch_norm_act(y.detach(), gam.detach(), bet.detach(), rm, rv, nb, True, act=ops.ACT_RELU))
    z = ops.batch_norm_act(y, gam, bet, rm, rv, nb, True, act=ops.ACT_RELU)
    g = torch.randn_like(z)

    def fb():
        y_ = ops.conv2d_3x3s2(x, w, b)
        z_ = ops.batch_norm_act(y_, gam, bet, rm, rv, nb, True, act=ops.ACT_RELU)
        z_.backward(g)
    t_fb = t(fb)
    flops = 2.0 * B * T * Wo * cout * 9 * cin
    print(f"L{i}: W {W:3d}->{Wo:3d} C {cin:3d}->{cout:3d}  conv fwd {tf_conv:7.1f} us (im2col {tf_im2col:6.1f}, {flops / tf_conv / 1e6:5.1f} TF)  "
          f"bn+relu fwd {tf_bn:6.1f} us   fwd+bwd total {t_fb:7.1f} us")
    tot_f += tf_conv + tf_bn
    tot_b += t_fb
    W = Wo
print(f"stack: fwd {tot_f:.0f} us, fwd+bwd {tot_b:.0f} us per encoder")
gi = torch.randn(B, T, 96, device=dev, requires_grad=True)
whh = (torch.randn(1, 96, 32, device=dev) * 0.1).requires_grad_(True); bhh = torch.zeros(1, 96, device=dev, requires_grad=True)
print(f"GRU<32> fwd {t(lambda: K.gru_fwd(gi.detach(), whh.detach(), bhh.detach(), 32, 1)):.0f} us")
