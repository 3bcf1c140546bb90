"""GPU (round 5): the BASELINE configurations at the size bench.py times them, and the captured path
free of stock-torch reductions.
  * ctts_cwt_pitch (csrc/pitch.hip) against the stock-torch chain it replaces (utils/pitch_tools.py:27-36,258-294);
  * no `at::native::reduce_kernel` launch in a train step of fs2 / conformer / C5 / the `ph` pitch variant (torch's multi-block reduction
    clears its semaphores with a memset node that mis-replays inside a hipGraph on this stack: DESIGN.md section 1);
  * BASELINE configs[2] (conformer) at B = 16 and the VCTK slice (configs[3]) at B = 8: the train step is finite and BIT-reproducible, and
    the eval() forward of the full batch reproduces, row for row, the forward of a 4-utterance sub-batch - which is checked against the
    CPU oracle (oracle/restate.py, pinned to the live reference: tests/golden/reference_vs_oracle_full_size.json)."""
import os

import numpy as np
import pytest
import torch

import ctts_amd
from ctts_amd import kernels as K
from ctts_amd import model as M
from ctts_amd.configs import get_configs
from ctts_amd.synthetic import make_batch, make_unsup_batch, to_device, as_model_args, CANONICAL_SRC_LENS

pytestmark = pytest.mark.gpu
DEV = "cuda"


# ---------------------------------------------------------------------------------------------------------------- pitch chain
@pytest.mark.parametrize("B,T,inference", [(16, 1024, False), (3, 77, False), (16, 1000, True), (2, 5, True)])
def test_cwt_pitch_kernel_matches_the_stock_torch_chain(B, T, inference):
    g = torch.Generator().manual_seed(B * 1000 + T)
    cwt = (torch.randn(B, T, 11, generator=g) * 0.6).to(DEV)
    mean, std = (torch.randn(B, generator=g) * 0.2 + 5.0).to(DEV), (torch.rand(B, generator=g) * 0.3 + 0.1).to(DEV)
    eps = 1e-6
    pk = dict(eps=eps, mel_min=M.F0_MEL_MIN, mel_max=M.F0_MEL_MAX, f0_bin=M.F0_BIN)
    if inference:
        f0, den, ids = K.cwt_pitch(cwt, mean, std, 0.8, uv_chan=10, nscale=10, **pk)
        ref_f0 = M.cwt2f0_norm(cwt[:, :, :10], mean, std * 0.8, T, eps)
        uv = cwt[:, :, -1] > 0
    else:
        spec = cwt[:, :, :10].contiguous()
        uvf = (torch.rand(B, T, generator=g) < 0.3).float().to(DEV)
        f0, den, ids = K.cwt_pitch(spec, mean, std, 1.0, uv=uvf, width=T, **pk)
        ref_f0 = M.cwt2f0_norm(spec, mean, std, T, eps)
        uv = uvf > 0
    ref_den = torch.where(uv, torch.zeros_like(ref_f0), 2 ** ref_f0)
    ref_ids = M.f0_to_coarse(ref_den)
    assert float((f0 - ref_f0).abs().max()) <= 2e-5, float((f0 - ref_f0).abs().max())
    assert float(((den - ref_den).abs() / (1.0 + ref_den.abs())).max()) <= 2e-5
    assert ids.dtype == torch.int64 and int(ids.min()) >= 1 and int(ids.max()) <= 255
    diff = (ids - ref_ids).abs()
    # the bins agree except where a value sits within rounding of a bin edge (the chain's reductions are summed in another order)
    assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) <= 2e-3, (int(diff.max()), float((diff > 0).float().mean()))
    assert torch.equal(den == 0, uv)
    f0b, denb, idsb = (K.cwt_pitch(cwt, mean, std, 0.8, uv_chan=10, nscale=10, **pk) if inference else
                       K.cwt_pitch(spec, mean, std, 1.0, uv=uvf, width=T, **pk))
    assert torch.equal(f0, f0b) and torch.equal(ids, idsb)


def test_cwt_pitch_single_frame_is_nan_like_torch_std():
    cwt = torch.randn(2, 1, 11).to(DEV)
    f0, den, ids = K.cwt_pitch(cwt, torch.zeros(2, device=DEV), torch.ones(2, device=DEV), 1.0, uv_chan=10, nscale=10, eps=1e-6,
                               mel_min=M.F0_MEL_MIN, mel_max=M.F0_MEL_MAX, f0_bin=M.F0_BIN)
    assert torch.isnan(f0).all()                                        # unbiased std of one sample = NaN, as in the reference


# ---------------------------------------------------------------------------------------------------------------- captured path
def _step(block="transformer_fs2", c5=False, pitch_type=None, lens=(60, 41, 33, 17), dataset="LJSpeech", use_graph=False, nodrop=False):
    from ctts_amd.loss import CompTransTTSLoss, ScheduledOptim
    from ctts_amd.trainer import TrainStep
    torch.manual_seed(1234)
    pre, mc, tc = get_configs(dataset)
    mc["block_type"] = block
    if c5:
        mc["prosody_modeling"]["model_type"] = "liu2021"
        mc["duration_modeling"]["learn_alignment"] = True
    if pitch_type:
        pre["preprocessing"]["pitch"]["pitch_type"] = pitch_type
    model = ctts_amd.CompTransTTS(pre, mc, tc).to(DEV)
    model.train()
    if nodrop:
        for sub in model.modules():
            for attr in ("dropout", "p_drop"):
                if isinstance(getattr(sub, attr, None), float):
                    setattr(sub, attr, 0.0)
    loss_fn, optim = CompTransTTSLoss(pre, mc, tc).to(DEV), ScheduledOptim(model, tc, mc, 50000, capturable=True)
    cap = 1000 if block == "conformer" else None
    extra = dict(multi_speaker=True) if dataset == "VCTK" else {}
    mk = make_unsup_batch if c5 else make_batch
    batch = to_device(mk(None if lens is None else list(lens), 8, seed=3, max_mel_cap=cap, **extra), DEV)
    step = TrainStep(model, loss_fn, optim, as_model_args(batch), world=1, use_graph=use_graph)
    if c5:
        step.step_no = 100001
    return step, model, batch


@pytest.mark.parametrize("name,kw", [("fs2 canonical", dict(lens=None)), ("conformer", dict(block="conformer")), ("C5", dict(c5=True)),
                                     ("ph pitch", dict(pitch_type="ph"))])
def test_no_stock_torch_reduction_kernel_in_a_train_step(name, kw):
    """one eager step = exactly the launches a captured step replays"""
    from torch.profiler import profile, ProfilerActivity
    step, _, _ = _step(**kw)
    for _ in range(2):
        step.optim.update_learning_rate()
        step._eager()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        step.optim.update_learning_rate()
        step._eager()
        torch.cuda.synchronize()
    names = [e.name for e in prof.events() if getattr(e, "device_type", None) is not None and "cuda" in str(e.device_type).lower()]
    assert len(names) > 100, f"profiler saw only {len(names)} device kernels"
    bad = sorted({n for n in names if "at::native::reduce_kernel" in n})
    assert not bad, f"{name}: stock-torch reductions on the step's path: {bad}"
    ours = sum(1 for n in names if "anonymous namespace" in n or "ctts" in n)
    print(f"{name}: {len(names)} device launches, {len(names) - ours} of them stock torch ({100.0 * (len(names) - ours) / len(names):.1f} %)")


@pytest.mark.parametrize("use_graph", [False, True])
def test_side_stream_branches_change_nothing_but_the_schedule(use_graph):
    """liu2021 + learn_alignment (BASELINE configs[4]) with every dropout off (the side branches change the program order in which the
    dropout call sites draw their offsets, not the arithmetic): the step with the reference encoders beside the text encoder and the CTC
    recursions beside the decoder (ops.fork_side / mark_ready) reproduces the single-stream step - losses, the whole gradient arena and
    the parameters after three steps - bit for bit, eagerly and under hipGraph replay."""
    from ctts_amd import ops
    res = {}
    for side in (True, False):
        prev = (ops.SIDE_LOSS, ops.SIDE_PROSODY)
        ops.SIDE_LOSS = ops.SIDE_PROSODY = side
        try:
            step, _, _ = _step(c5=True, use_graph=use_graph, nodrop=True)
            if use_graph:
                step.capture(warmup=2)
            else:
                for _ in range(2):
                    step.optim.update_learning_rate()
                    step._eager()
            losses = []
            for i in range(3):
                step()
                losses.append(float(step.loss_val))
                if i == 0:
                    torch.cuda.synchronize()
                    grads = step.flat_grad.clone()
            torch.cuda.synchronize()
            res[side] = (losses, grads, step.fadam.flat_param.clone())
            del step
        finally:
            ops.SIDE_LOSS, ops.SIDE_PROSODY = prev
    a, b = res[True], res[False]
    assert a[0] == b[0], (a[0], b[0])
    assert torch.equal(a[1], b[1]), float((a[1] - b[1]).abs().max())
    assert torch.equal(a[2], b[2])


# ---------------------------------------------------------------------------------------------------------------- full-size configs
def _two_runs(kw, n=2):
    outs = []
    for _ in range(2):
        step, _, _ = _step(**kw)
        for _ in range(2):
            step.optim.update_learning_rate()
            step._eager()
        losses = []
        for i in range(n):
            step()
            losses.append(float(step.loss_val))
        torch.cuda.synchronize()
        outs.append((losses, step.flat_grad.clone(), step.fadam.flat_param.clone()))
        del step
        torch.cuda.empty_cache()
    return outs


@pytest.mark.parametrize("name,kw", [("configs[2] conformer B=16", dict(block="conformer", lens=None)),
                                     ("configs[3] VCTK slice B=8", dict(dataset="VCTK", lens=CANONICAL_SRC_LENS[0::2]))])
def test_full_size_train_step_is_finite_and_bit_reproducible(name, kw):
    a, b = _two_runs(kw)
    assert all(np.isfinite(a[0])), a[0]
    assert torch.isfinite(a[1]).all() and torch.isfinite(a[2]).all()
    assert a[0] == b[0], (a[0], b[0])
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), name
    assert max(int(ws.view(torch.int32)[2048].item()) for ws in K._SK_WS.values()) == 0


def _sub_batch(batch, idx):
    """utterances `idx` of a CPU batch dict at the SAME padded lengths: the reference normalises the pitch contour over all (padded) mel
    columns and the conformer attends over all (padded) keys, so an utterance's output depends on the padded size - not on its neighbours"""
    idx = list(idx)
    out = dict(batch)
    for k in ("speakers", "texts", "src_lens", "mel_lens", "mels", "e_targets", "d_targets", "spker_embeds"):
        if torch.is_tensor(batch.get(k)):
            out[k] = batch[k][idx].contiguous()
    out["p_targets"] = {k: (v[idx].contiguous() if torch.is_tensor(v) else v) for k, v in batch["p_targets"].items()}
    return out


@pytest.mark.parametrize("block,dataset", [("conformer", "LJSpeech"), ("transformer_fs2", "VCTK")])
def test_full_size_eval_forward_rows_equal_a_sub_batch_that_matches_the_oracle(block, dataset):
    from oracle import restate as R
    torch.manual_seed(7)
    pre, mc, tc = get_configs(dataset)
    mc["block_type"] = block
    m = ctts_amd.CompTransTTS(pre, mc, tc).to(DEV).eval()
    cap = 1000 if block == "conformer" else None
    extra = dict(multi_speaker=True) if dataset == "VCTK" else {}
    lens = None if dataset == "LJSpeech" else CANONICAL_SRC_LENS[0::2]
    full = make_batch(lens, 8, seed=11, max_mel_cap=cap, **extra)
    nb = full["texts"].shape[0]
    idx = [1, nb // 3, (2 * nb) // 3, nb - 1]
    sub = _sub_batch(full, idx)

    def fwd(b):
        a = list(as_model_args(to_device(b, DEV)))
        a[7] = dict(a[7])
        with torch.no_grad():
            return m(*a)
    of, osub = fwd(full), fwd(sub)
    worst = 0.0
    for j, i in enumerate(idx):
        L = int(full["mel_lens"][i])
        for k in (0, 1):                                     # mel and postnet mel, valid frames of the utterance
            worst = max(worst, float((of[k][i, :L] - osub[k][j, :L]).abs().max()))
        Ls = int(full["src_lens"][i])
        worst = max(worst, float((of[4][i, :Ls] - osub[4][j, :Ls]).abs().max()))          # log-duration prediction
    print(f"{block} / {dataset}: B = {nb} rows vs 4-utterance sub-batch: max |diff| {worst:.2e}")
    assert worst <= 5e-5, worst
    # the sub-batch against the CPU oracle
    sd = {k: v.detach().cpu().contiguous().clone() for k, v in m.state_dict().items()}
    a = list(as_model_args(sub))
    a[7] = dict(a[7])
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    with torch.no_grad():
        fn = R.comp_trans_tts_forward_conformer if block == "conformer" else R.comp_trans_tts_forward
        ref = fn(sd, mc, pre, *a, step=50001, training=False) if block == "conformer" else fn(sd, mc, pre, *a, training=False)
    e_mel = float((osub[0].cpu() - ref[0]).abs().max())
    e_post = float((osub[1].cpu() - ref[1]).abs().max())
    print(f"sub-batch vs oracle: mel {e_mel:.2e} postnet {e_post:.2e}")
    assert e_mel <= 1e-3 and e_post <= 1e-3, (e_mel, e_post)
