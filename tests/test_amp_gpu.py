"""GPU tests of the OPTIONAL reduced-precision arithmetic (reference train.py:13,59,104-124 `--use_amp`:
`amp.autocast(args.use_amp)` + `GradScaler`): inside torch.amp.autocast the launches the plane kernels take - the Conv1d layers with many
rows, forward, data gradient and weight gradient - round their operands to bf16 and use ONE MFMA term (ctts_gemm_desc.bf16_split 3 / 4;
csrc/gemm_pl.hip, csrc/gemm_plw.hip TERMS = 1), fp32 accumulate, fp32 tensors everywhere.  Never the default and never the headline; its
tolerances are its own and are stated here:
  * a launch: EQUAL to the product of the bf16-rounded operands up to fp32 summation (2e-6 of the largest entry) - the arithmetic is
    exactly "bf16 operands, fp32 accumulate", nothing looser;
  * a layer: within 2e-2 of the fp32 layer relative to the largest entry (bf16 rounding of both operands: 2^-8 each, averaged over the
    reduction) - and measurably DIFFERENT from it (the mode is really on);
  * the canonical fs2 model: eval mel within 0.05 max-abs of the fp32 forward (measured 0.018), losses of the first two train steps
    within 3 % and the gradient norm within 5 % of the fp32-class steps (measured: 5th digit)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from ctts_amd import kernels as K
    from ctts_amd import ops
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def bf16d(x):
    return x.bfloat16().double().cpu()


def close(got, ref, tol):
    scale = float(ref.abs().max())
    err = float((got.double().cpu() - ref).abs().max())
    assert err <= tol * scale, f"max |err| {err:.3e} vs scale {scale:.3e} (tol {tol})"
    return err / scale


def test_one_term_forward_launch_is_the_product_of_the_bf16_rounded_operands():
    M, N, Kd = 1024, 512, 768
    A, Bm = rnd(M, Kd, seed=1).to(DEV), rnd(N, Kd, seed=2, scale=0.05).to(DEV)
    ap, bp = K.split_planes([A, Bm])
    out = torch.full((M, N), float("nan"), device=DEV)
    args = (A, Bm, out, M, N, Kd, Kd, Kd, N, True, True)
    assert K.gemm_takes_planes(*args, a_planes=ap, b_planes=bp, bf16_split=4)
    K.gemm(*args, a_planes=ap, b_planes=bp, bf16_split=4)
    close(out, bf16d(A) @ bf16d(Bm).t(), 2e-6)
    # and it is NOT the fp32 product
    full = A.double().cpu() @ Bm.double().cpu().t()
    assert float((out.double().cpu() - full).abs().max()) > 1e-4 * float(full.abs().max())
    # conv view on A (the FFN forward in small): T = 128, k = 3, cin = 256, ragged rows
    T, k, cin, cout, nb = 128, 3, 256, 256, 4
    x, w = rnd(nb, T, cin, seed=3).to(DEV), rnd(cout, k * cin, seed=4, scale=0.05).to(DEV)
    lens = torch.tensor([128, 77, 1, 100], dtype=torch.int32, device=DEV)
    xp, wp = K.split_planes([x.view(nb * T, cin), w])
    y = torch.full((nb, T, cout), float("nan"), device=DEV)
    kw = dict(conv=(T, 1, cin), row_lens=lens, row_T=T, row_halo=0, a_planes=xp, b_planes=wp, bf16_split=4)
    K.gemm(x, w, y, nb * T, cout, k * cin, cin, k * cin, cout, True, True, **kw)
    xb = torch.nn.functional.pad(bf16d(x), (0, 0, 1, 1))
    cols = torch.cat([xb[:, t:t + T] for t in range(k)], dim=2)                      # [nb, T, k * cin], tap-major like the GEMM-major weight
    ref = cols @ bf16d(w).t()
    for b, L in enumerate(lens.tolist()):
        close(y[b, :L], ref[b, :L], 2e-6)
        assert torch.equal(y[b, -(-L // 64) * 64:], torch.zeros_like(y[b, -(-L // 64) * 64:]))      # the zero rule is the six-term kernel's


def test_one_term_weight_gradient_launch_is_the_product_of_the_bf16_rounded_operands():
    T, k, pad, cin, cout, nb = 64, 3, 1, 256, 128, 4
    rows = nb * T
    dZ, X = rnd(rows, cout, seed=5).to(DEV), rnd(rows, cin, seed=6).to(DEV)
    ap, bp = K.split_planes([dZ, X])
    out = torch.full((cout, k * cin), float("nan"), device=DEV)
    kw = dict(conv=(T, pad, cin), conv_on_b=True, split_k=2, split_overwrite=True, a_planes=ap, b_planes=bp, bf16_split=4)
    assert K.gemm_takes_planes(dZ, X, out, cout, k * cin, rows, cout, cin, k * cin, False, False, **kw)
    K.gemm(dZ, X, out, cout, k * cin, rows, cout, cin, k * cin, False, False, **kw)
    dz, x = bf16d(dZ).view(nb, T, cout), bf16d(X).view(nb, T, cin)
    ref = []
    for tap in range(k):
        s = tap - pad
        xs = torch.zeros_like(x)
        lo, hi = max(0, -s), min(T, T - s)
        xs[:, lo:hi] = x[:, lo + s:hi + s]
        ref.append(torch.einsum("btn,btc->nc", dz, xs))
    close(out, torch.cat(ref, dim=1), 2e-6)


def _ffn_layer(autocast):
    """the decoder FFN conv at full size through ops.linear_conv (forward + backward), with and without autocast"""
    torch.manual_seed(7)
    B, T, cin, cout, k = 16, 1024, 256, 1024, 9
    x = torch.randn(B, T, cin, device=DEV, requires_grad=True)
    w = torch.nn.Parameter((torch.randn(cout, k, cin, device=DEV) * 0.02).permute(0, 2, 1))      # GEMM-major memory, Conv1d shape
    b = torch.nn.Parameter(torch.zeros(cout, device=DEV))
    with torch.amp.autocast("cuda", enabled=autocast):
        y = ops._LinearConv.apply(x, w, b, None, None, ops.ACT_NONE, 1.0, 0.0, None, 0, k, None, 0)
    g = torch.randn(B, T, cout, generator=torch.Generator().manual_seed(8)).to(DEV)
    ops.set_grad_accumulation_fusion(False)          # plain autograd: the weight gradient comes back as a tensor
    y.backward(g)
    return y.detach(), x.grad.detach(), w.grad.detach().clone()


def test_autocast_selects_the_one_term_arithmetic_for_a_conv_layer_and_only_there():
    y0, dx0, dw0 = _ffn_layer(False)
    y1, dx1, dw1 = _ffn_layer(True)
    assert y1.dtype == torch.float32 and dx1.dtype == torch.float32 and dw1.dtype == torch.float32
    for name, a, bb in (("y", y0, y1), ("dx", dx0, dx1), ("dw", dw0, dw1)):
        rel = close(bb, a.double().cpu(), 2e-2)
        assert rel > 1e-4, f"{name}: autocast left the layer in the fp32-class arithmetic (rel diff {rel:.2e})"
    y2, dx2, dw2 = _ffn_layer(False)
    assert torch.equal(y0, y2) and torch.equal(dx0, dx2) and torch.equal(dw0, dw2)      # nothing sticks after the context


def test_canonical_model_under_autocast_tracks_the_fp32_model_within_its_own_tolerance():
    import ctts_amd
    from ctts_amd.configs import get_configs
    from ctts_amd.loss import CompTransTTSLoss
    from ctts_amd.synthetic import make_batch, to_device, as_model_args
    pre, mc, tc = get_configs("LJSpeech")
    torch.manual_seed(1234)
    model = ctts_amd.CompTransTTS(pre, mc, tc).to(DEV)
    loss_fn = CompTransTTSLoss(pre, mc, tc).to(DEV)
    batch = to_device(make_batch(None, 8, seed=3), DEV)
    args = as_model_args(batch)
    model.eval()
    outs = []
    for ac in (False, True):
        with torch.no_grad(), torch.amp.autocast("cuda", enabled=ac):
            o = model(*args, step=50001)
        outs.append(o[1].float().clone())                                   # postnet mel
    d = float((outs[0] - outs[1]).abs().max())
    print("eval postnet-mel |amp - fp32| max", d)
    assert np.isfinite(d) and 1e-6 < d <= 0.05, d


def _first_step_loss(amp):
    import ctts_amd
    from ctts_amd.configs import get_configs
    from ctts_amd.loss import CompTransTTSLoss, ScheduledOptim
    from ctts_amd.synthetic import make_batch, to_device, as_model_args
    from ctts_amd.trainer import TrainStep
    prev = K.gemm_bf16_split_enable("amp" if amp else True)
    try:
        torch.manual_seed(1234)
        pre, mc, tc = get_configs("LJSpeech")
        model = ctts_amd.CompTransTTS(pre, mc, tc).to(DEV)
        model.train()
        loss_fn, optim = CompTransTTSLoss(pre, mc, tc).to(DEV), ScheduledOptim(model, tc, mc, 50000, capturable=True)
        batch = to_device(make_batch(None, 8, seed=3), DEV)
        step = TrainStep(model, loss_fn, optim, as_model_args(batch), world=1, use_graph=False)
        out = []
        for _ in range(2):
            step()
            out.append(float(step.loss_val))
        torch.cuda.synchronize()
        gn = float(step.flat_grad.norm())
        return out, gn
    finally:
        K.gemm_bf16_split_enable(prev)
        ops.set_grad_accumulation_fusion(False)


def test_canonical_train_steps_in_the_amp_arithmetic_track_the_fp32_steps():
    """the whole-model switch (kernels.gemm_bf16_split_enable("amp"): what bench.py's AMP line times): same seeds, dropout on"""
    l0, g0 = _first_step_loss(False)
    l1, g1 = _first_step_loss(True)
    print("losses fp32-class", l0, "amp", l1, "grad norms", g0, g1)
    assert all(np.isfinite(l1)) and np.isfinite(g1)
    assert l0 != l1, "the amp switch changed nothing"
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 3e-2 * abs(a), (l0, l1)
    assert abs(g0 - g1) <= 5e-2 * g0, (g0, g1)
