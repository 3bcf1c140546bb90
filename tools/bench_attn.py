"""Micro-benchmark + A/B check of the fused attention kernels (csrc/attn.hip) against the unfused GEMM / softmax pipeline at the
BASELINE shapes: fs2 decoder (B=16, T=1024, 2 heads x 128, canonical ragged lengths) and conformer decoder (B=16, T=1000, 8 x 32,
dropout 0.1).  HIP-event timing on the launch stream; prints one line per (op, path)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctts_amd  # noqa: E402,F401
from ctts_amd import kernels as K, ops  # noqa: E402
from ctts_amd.synthetic import CANONICAL_SRC_LENS  # noqa: E402

DEV = torch.device("cuda:0")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def relerr(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-20))


def fs2(scale_in=1.0):
    B, T, H, C = 16, 1024, 2, 256
    lens = torch.tensor([8 * s for s in CANONICAL_SRC_LENS], dtype=torch.int32, device=DEV)
    g = torch.Generator(device="cpu").manual_seed(1)
    qkv = (torch.randn(B, T, 3 * C, generator=g) * scale_in).to(DEV)
    nonpad = (torch.arange(T, device=DEV)[None, :] < lens[:, None]).float()[..., None]
    go = torch.randn(B, T, C, generator=g).to(DEV) * nonpad
    res = {}
    for fused in (False, True):
        ops.set_fused_attention(fused)
        q = qkv.clone().requires_grad_()
        y = ops.self_attention(q, lens, H)
        y.backward(go)
        res[fused] = (y.detach(), q.grad)
        tf = timeit(lambda: ops.self_attention(qkv, lens, H))
        q2 = qkv.clone().requires_grad_()

        def fb():
            q2.grad = None
            ops.self_attention(q2, lens, H).backward(go)
        tfb = timeit(fb)
        print(f"fs2 dec attention  fused={int(fused)}  scale_in={scale_in}: fwd {tf:8.1f} us   fwd+bwd {tfb:8.1f} us", flush=True)
    print(f"   fused vs unfused: out rel {relerr(res[True][0], res[False][0]):.2e}  dqkv rel {relerr(res[True][1], res[False][1]):.2e}  "
          f"finite {bool(torch.isfinite(res[True][1]).all())}", flush=True)
    valid = sum(int(l) ** 2 for l in lens.tolist())
    print(f"   algorithmic: fwd {4 * 128 * 2 * valid / 1e9:.2f} GFLOP, bwd(5 GEMMs) {10 * 128 * 2 * valid / 1e9:.2f} GFLOP")


def conformer():
    B, T, H, C, p = 16, 1000, 8, 256, 0.1
    g = torch.Generator(device="cpu").manual_seed(2)
    qu, qv, kv = [torch.randn(B, T, n, generator=g).to(DEV) for n in (C, C, 2 * C)]
    pos = torch.randn(T, C, generator=g).to(DEV)
    go = torch.randn(B, T, C, generator=g).to(DEV)
    res = {}
    for fused in ((True,) if "--fused-only" in sys.argv else (False, True)):
        ops.set_fused_attention(fused)
        drop = K.DropCtx(DEV, seed=5)
        ts = [t.clone().requires_grad_() for t in (qu, qv, kv, pos)]
        y = ops.relpos_attention(*ts, H, 1.0 / 16, p_drop=p, drop=drop)
        y.backward(go)
        res[fused] = [y.detach()] + [t.grad for t in ts]
        d2 = K.DropCtx(DEV, seed=5)
        tf = timeit(lambda: ops.relpos_attention(qu, qv, kv, pos, H, 1.0 / 16, p_drop=p, drop=d2), iters=5, warm=2)
        ts2 = [t.clone().requires_grad_() for t in (qu, qv, kv, pos)]

        def fb():
            for t in ts2:
                t.grad = None
            ops.relpos_attention(*ts2, H, 1.0 / 16, p_drop=p, drop=d2).backward(go)
        tfb = timeit(fb, iters=5, warm=2)
        print(f"conformer dec attention  fused={int(fused)}: fwd {tf:8.1f} us   fwd+bwd {tfb:8.1f} us", flush=True)
        del ts, ts2
        torch.cuda.empty_cache()
    if False in res:
        print("   fused vs unfused rel errs:", [f"{relerr(a, b):.2e}" for a, b in zip(res[True], res[False])], flush=True)
    print(f"   algorithmic (QK, QP, PV) fwd {3 * 2 * 32 * T * T * B * H / 1e9:.1f} GFLOP; train x3")


def trajectory():
    """eager train steps, dropout off: loss trajectory fused vs unfused (finds divergence of the fused gradients in the real model)"""
    from ctts_amd.configs import get_configs
    from ctts_amd.loss import CompTransTTSLoss, ScheduledOptim
    from ctts_amd.synthetic import make_batch, to_device, as_model_args
    from ctts_amd.trainer import TrainStep
    for fused in (False, True):
        ops.set_fused_attention(fused)
        torch.manual_seed(1234)
        pre, mc, tc = get_configs()
        model = ctts_amd.CompTransTTS(pre, mc, tc).to(DEV)
        for m in model.modules():
            if hasattr(m, "dropout"):
                m.dropout = 0.0
        model.train()
        loss_fn = CompTransTTSLoss(pre, mc, tc).to(DEV)
        optim = ScheduledOptim(model, tc, mc, 50000, capturable=True)
        step = TrainStep(model, loss_fn, optim, as_model_args(to_device(make_batch(), DEV)), use_graph=False)
        ls, gn = [], []
        for _ in range(6):
            step()
            ls.append(round(float(step.loss_val), 5))
            gn.append(round(float(step.fadam.total_norm), 4))
        print(f"trajectory fused={int(fused)}: loss {ls}  |g| {gn}", flush=True)
    ops.set_fused_attention(None)


if __name__ == "__main__":
    what = [a for a in sys.argv[1:] if not a.startswith("--")] or ["fs2", "conformer", "trajectory"]
    if "fs2" in what:
        fs2(1.0)
        fs2(6.0)
    if "conformer" in what:
        conformer()
    if "trajectory" in what:
        trajectory()
