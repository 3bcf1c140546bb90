"""GPU: the drop-in CompTransTTS (HIP kernels through the C ABI) against (a) the golden vectors
captured from the live reference and (b) the CPU oracle on the same seeded inputs.
Tolerance: north_star states mel max-abs <= 1e-3, LengthRegulator indices bit-exact."""
import os

import numpy as np
import pytest
import torch

import ctts_amd
from ctts_amd.configs import get_configs
from ctts_amd.synthetic import make_batch, to_device, as_model_args
from oracle import restate as R
from oracle.weights import _hash_uniform
from tests.util import load_golden, closed_form_sd, batch_from_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"
MEL_TOL = 1e-3


def build(dataset="LJSpeech", sd=None, block="transformer_fs2", vp=None):
    pre, mc, tc = get_configs(dataset)
    mc["block_type"] = block
    mc["variance_predictor"].update(vp or {})
    m = ctts_amd.CompTransTTS(pre, mc, tc)
    if sd is not None:
        m.load_state_dict(sd)
    return m.to(DEV), (pre, mc, tc)


def no_dropout(m):
    for sub in m.modules():
        if hasattr(sub, "dropout"):
            sub.dropout = 0.0


def args_from(b, dev=DEV):
    b = to_device(b, dev)
    return (b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"], b["mel_lens"], b["max_mel_len"],
            b["p_targets"], b["e_targets"], b["d_targets"], None, b["spker_embeds"])


def maxerr(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    return float(np.abs(a - np.asarray(b, dtype=np.float64)).max())


def check_against_golden(out, g):
    mel, post, p_pred, e_pred, log_d, d_rounded, src_mask, mel_mask, src_lens, mel_lens = out[:10]
    errs = {
        "mel": maxerr(mel, g["out.mel"]), "postnet_mel": maxerr(post, g["out.postnet_mel"]),
        "log_d": maxerr(log_d, g["out.log_d"]), "e_pred": maxerr(e_pred, g["out.e_pred"]),
        "cwt": maxerr(p_pred["cwt"], g["out.cwt"]), "f0_mean": maxerr(p_pred["f0_mean"], g["out.f0_mean"]),
        "f0_std": maxerr(p_pred["f0_std"], g["out.f0_std"]),
    }
    print("max-abs vs reference golden:", {k: f"{v:.2e}" for k, v in errs.items()})
    for k, v in errs.items():
        assert v <= MEL_TOL, (k, v)
    assert np.array_equal(src_mask.cpu().numpy(), g["out.src_mask"])
    assert np.array_equal(mel_mask.cpu().numpy(), g["out.mel_mask"])
    assert np.array_equal(mel_lens.cpu().numpy(), g["out.mel_lens"])
    assert np.array_equal(d_rounded.cpu().numpy(), g["out.d_rounded"])
    assert maxerr(p_pred["f0_denorm"], g["out.f0_denorm"]) < 0.05
    return errs


def test_state_dict_roundtrip_reference_keys():
    sd = closed_form_sd()
    m, _ = build(sd=sd)
    out = m.state_dict()
    assert list(sorted(out.keys())) == list(sorted(sd.keys()))
    for k in sd:
        if "_float_tensor" in k:
            continue
        assert torch.equal(out[k].cpu(), sd[k]), k


def test_g1_eval_matches_reference():
    g = load_golden("g1_fs2_eval")
    m, _ = build(sd=closed_form_sd())
    m.eval()
    with torch.no_grad():
        out = m(*args_from(batch_from_golden(g)))
    check_against_golden(out, g)
    assert maxerr(out[12]["f0"], g["out.pt_f0"]) < 1e-3


def test_g2_train_batchnorm_and_gradients_match_reference():
    g = load_golden("g2_fs2_train_nodrop")
    m, _ = build(sd=closed_form_sd())
    m.train()
    no_dropout(m)
    out = m(*args_from(batch_from_golden(g)))
    check_against_golden(out, g)
    sd_after = m.state_dict()
    for k in sd_after:
        if "running_" in k:
            assert maxerr(sd_after[k], g["bn." + k]) < 1e-4, k

    def pseudo(name, shape):
        return torch.from_numpy(_hash_uniform("probe." + name, int(np.prod(shape))).reshape(shape)).float().to(DEV)
    mel, post, p_pred, e_pred, log_d = out[:5]
    loss = ((post * pseudo("post", post.shape)).sum() + (mel * pseudo("mel", mel.shape)).sum()
            + (log_d * pseudo("logd", log_d.shape)).sum() + (e_pred * pseudo("e", e_pred.shape)).sum()
            + (p_pred["cwt"] * pseudo("cwt", p_pred["cwt"].shape)).sum()
            + (p_pred["f0_mean"] * 0.7).sum() + (p_pred["f0_std"] * -0.3).sum())
    assert abs(loss.item() - float(g["grad.loss"])) < 5e-2
    loss.backward()
    worst = ("", 0.0)
    n = 0
    for k, p in m.named_parameters():
        if "grad.stat." + k not in g:
            continue
        gs = g["grad.stat." + k]
        gr = p.grad.flatten() if p.grad is not None else torch.zeros(p.numel(), device=DEV)
        scale = max(1.0, float(gs[1]))
        e = maxerr(gr[:64], g["grad.head." + k]) / scale
        en = abs(float(gr.double().pow(2).sum().sqrt()) - gs[1]) / scale
        if max(e, en) > worst[1]:
            worst = (k, max(e, en))
        n += 1
    print("worst relative gradient error:", worst, "over", n, "parameters")
    # closed-form weights put many pre-activations of the predictor stacks next to the ReLU kink: a 1e-6 change upstream (summation
    # order of a different kernel) flips a few units, which moves a conv-bias gradient by ~2e-3 of its norm
    assert n > 150 and worst[1] < 4e-3, worst


def test_g3_inference_branch_matches_reference():
    g = load_golden("g3_fs2_infer")
    m, _ = build(sd=closed_form_sd())
    m.eval()
    b = batch_from_golden(g)
    with torch.no_grad():
        out = m(*args_from(b), p_control=1.1, e_control=0.9, d_control=2.0)
    check_against_golden(out, g)


def test_g5_vctk_multispeaker_matches_reference():
    g = load_golden("g5_vctk_eval")
    m, _ = build("VCTK", sd=closed_form_sd("VCTK"))
    m.eval()
    with torch.no_grad():
        out = m(*args_from(batch_from_golden(g)))
    check_against_golden(out, g)


def test_canonical_batch_vs_oracle_full_size():
    """BASELINE config 2 shapes (B=16, Ts<=128, Tm<=1024): HIP forward vs the CPU oracle."""
    torch.manual_seed(1)
    m, (pre, mc, tc) = build()
    m.eval()
    batch = make_batch()
    with torch.no_grad():
        out = m(*as_model_args(to_device(batch, DEV)))
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ref = R.comp_trans_tts_forward(sd, mc, pre, *as_model_args(batch), training=False)
    e_mel, e_post = maxerr(out[0], ref[0].numpy()), maxerr(out[1], ref[1].numpy())
    print(f"canonical batch: mel max-abs {e_mel:.2e}, postnet mel {e_post:.2e}")
    assert e_mel <= MEL_TOL and e_post <= MEL_TOL
    assert torch.equal(out[9].cpu(), ref[9])


def test_canonical_batch_train_mode_gradients_vs_oracle_full_size():
    """BASELINE configs[1] at FULL size (B=16, Ts<=128, Tm<=1024, 11,992 valid frames), train mode (BatchNorm batch statistics),
    dropout off: outputs and ALL parameter gradients of the reference's loss against the CPU oracle's autograd - the sizes that
    select the long-sequence code paths (attention key tiling / split reductions, 3,104 active GEMM tiles, split-K wgrad)."""
    from ctts_amd.loss import CompTransTTSLoss
    from oracle.loss_restate import RefLoss
    ops_mod = __import__("ctts_amd").ops
    ops_mod.set_grad_accumulation_fusion(False)
    torch.manual_seed(1)
    m, (pre, mc, tc) = build()
    m.train()
    no_dropout(m)
    sd = {k: v.detach().cpu().contiguous().clone() for k, v in m.state_dict().items()}
    batch = make_batch()
    args = list(as_model_args(to_device(batch, DEV)))
    args[7] = dict(args[7])
    out = m(*args, step=50001)
    inputs = [None, None] + args
    inputs[9:11] = out[-2:]
    losses = CompTransTTSLoss(pre, mc, tc).to(DEV)(inputs, out[:-2], 50001)
    losses[0].backward()
    torch.cuda.synchronize()
    trainable = {k for k, p in m.named_parameters() if p.requires_grad}
    sdg = {k: (v.requires_grad_(True) if k in trainable else v) for k, v in sd.items()}
    a = list(as_model_args(batch))
    a[7] = dict(a[7])
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    ref = R.comp_trans_tts_forward(sdg, mc, pre, *a, step=50001, training=True, train_dropout=False, new_stats={})
    rin = [None, None] + a
    rin[9:11] = ref[-2:]
    rl = RefLoss(pre, mc, tc)(rin, ref[:-2], 50001)
    rl[0].backward()
    e_mel, e_post = maxerr(out[0], ref[0].detach().numpy()), maxerr(out[1], ref[1].detach().numpy())
    assert e_mel <= MEL_TOL and e_post <= MEL_TOL, (e_mel, e_post)
    assert abs(float(losses[0]) - float(rl[0])) <= 1e-4 * abs(float(rl[0]))
    worst, n = ("", 0.0), 0
    gmax = max(float(v.grad.abs().max()) for v in sdg.values() if v.requires_grad and v.grad is not None)
    for k, p in m.named_parameters():
        if not p.requires_grad:
            continue
        g_ref = sdg[k].grad
        if g_ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        # per-tensor relative error; tensors whose true gradient is zero (conv biases in front of a train-mode BatchNorm) hold
        # only cancellation noise, so the scale is floored at 1e-4 of the largest gradient entry of the model
        e = maxerr(p.grad, g_ref.numpy()) / max(float(g_ref.abs().max()), 1e-4 * gmax)
        n += 1
        if e > worst[1]:
            worst = (k, e)
    print(f"canonical batch, train mode: mel {e_mel:.2e} postnet {e_post:.2e} loss {float(losses[0]):.6f} vs {float(rl[0]):.6f}; "
          f"worst per-tensor gradient rel-max err {worst[1]:.2e} ({worst[0]}) over {n} tensors")
    assert n >= 170 and worst[1] <= 2e-3, worst


def _grads_vs_oracle(m, sdg, min_tensors, bar):
    """worst per-tensor relative gradient error of the product's parameters against the oracle's autograd"""
    worst, n = ("", 0.0), 0
    gmax = max(float(v.grad.abs().max()) for v in sdg.values() if v.requires_grad and v.grad is not None)
    for k, p in m.named_parameters():
        if not p.requires_grad:
            continue
        g_ref = sdg[k].grad
        if g_ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        e = maxerr(p.grad, g_ref.numpy()) / max(float(g_ref.abs().max()), 1e-4 * gmax)
        n += 1
        if e > worst[1]:
            worst = (k, e)
    assert n >= min_tensors and worst[1] <= bar, (n, worst)
    return worst, n


def test_canonical_batch_fused_accumulation_through_trainstep_vs_oracle_full_size():
    """The mode bench.py runs: trainer.TrainStep switches gradient-accumulation fusion ON (kernels add straight into the flat arena,
    weight-gradient partials land in param.grad, the persistent stream-K GEMM takes the large reductions).  One eager forward + loss +
    backward through TrainStep's own code path at the canonical size, dropout off, and the arena against the oracle's autograd."""
    from ctts_amd.loss import CompTransTTSLoss, ScheduledOptim
    from ctts_amd.trainer import TrainStep
    from oracle.loss_restate import RefLoss
    torch.manual_seed(1)
    m, (pre, mc, tc) = build()
    m.train()
    no_dropout(m)
    sd = {k: v.detach().cpu().contiguous().clone() for k, v in m.state_dict().items()}
    batch = make_batch()
    args = list(as_model_args(to_device(batch, DEV)))
    loss_fn = CompTransTTSLoss(pre, mc, tc).to(DEV)
    optim = ScheduledOptim(m, tc, mc, 50000, capturable=True)
    step = TrainStep(m, loss_fn, optim, args, use_graph=False, adam_step=optim.current_step)
    assert __import__("ctts_amd").ops.grad_accumulation_fusion()
    for _ in step._stages():            # forward + loss + backward into the arena, exactly as _eager() does before the optimizer
        pass
    torch.cuda.synchronize()
    trainable = {k for k, p in m.named_parameters() if p.requires_grad}
    sdg = {k: (v.requires_grad_(True) if k in trainable else v) for k, v in sd.items()}
    a = list(as_model_args(batch))
    a[7] = dict(a[7])
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    ref = R.comp_trans_tts_forward(sdg, mc, pre, *a, step=50001, training=True, train_dropout=False, new_stats={})
    rin = [None, None] + a
    rin[9:11] = ref[-2:]
    rl = RefLoss(pre, mc, tc)(rin, ref[:-2], 50001)
    rl[0].backward()
    assert abs(float(step.loss_val) - float(rl[0])) <= 1e-4 * abs(float(rl[0]))
    for p in step.params:                # every trainable parameter's gradient IS a view of the arena
        assert p.grad is not None and p.grad.data_ptr() >= step.flat_grad.data_ptr()
    worst, n = _grads_vs_oracle(m, sdg, 170, 2e-3)
    print(f"TrainStep (fusion on), canonical batch: loss {float(step.loss_val):.6f} vs {float(rl[0]):.6f}; worst gradient {worst[1]:.2e} "
          f"({worst[0]}) over {n} tensors")


def test_conformer_b4_t1000_train_mode_gradients_vs_oracle_full_length():
    """BASELINE configs[2] geometry, train mode (BatchNorm batch statistics), dropout off, decoder at the full T = 1000: outputs and every
    parameter gradient against the oracle's autograd - the fused relative-position attention BACKWARD (63-wide Toeplitz band, lane
    rotation, padded dS slab) and the depthwise-conv weight gradient at the length the bench runs."""
    from ctts_amd.loss import CompTransTTSLoss
    from ctts_amd.synthetic import CANONICAL_SRC_LENS
    from oracle.loss_restate import RefLoss
    ops_mod = __import__("ctts_amd").ops
    ops_mod.set_grad_accumulation_fusion(False)
    torch.manual_seed(2)
    m, (pre, mc, tc) = build(block="conformer")
    m.train()
    no_dropout(m)
    sd = {k: v.detach().cpu().contiguous().clone() for k, v in m.state_dict().items()}
    batch = make_batch(CANONICAL_SRC_LENS[:4], max_mel_cap=1000)
    args = list(as_model_args(to_device(batch, DEV)))
    args[7] = dict(args[7])
    out = m(*args, step=50001)
    inputs = [None, None] + args
    inputs[9:11] = out[-2:]
    losses = CompTransTTSLoss(pre, mc, tc).to(DEV)(inputs, out[:-2], 50001)
    losses[0].backward()
    torch.cuda.synchronize()
    trainable = {k for k, p in m.named_parameters() if p.requires_grad}
    sdg = {k: (v.requires_grad_(True) if k in trainable else v) for k, v in sd.items()}
    a = list(as_model_args(batch))
    a[7] = dict(a[7])
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    ref = R.comp_trans_tts_forward_conformer(sdg, mc, pre, *a, step=50001, training=True, train_dropout=False, new_stats={})
    rin = [None, None] + a
    rin[9:11] = ref[-2:]
    rl = RefLoss(pre, mc, tc)(rin, ref[:-2], 50001)
    rl[0].backward()
    e_mel, e_post = maxerr(out[0], ref[0].detach().numpy()), maxerr(out[1], ref[1].detach().numpy())
    assert out[0].shape[1] == 1000 and e_mel <= MEL_TOL and e_post <= MEL_TOL, (e_mel, e_post)
    assert abs(float(losses[0]) - float(rl[0])) <= 1e-4 * abs(float(rl[0]))
    worst, n = _grads_vs_oracle(m, sdg, 300, 2e-3)
    print(f"conformer B=4 T=1000, train mode: mel {e_mel:.2e} postnet {e_post:.2e} loss {float(losses[0]):.6f} vs {float(rl[0]):.6f}; "
          f"worst gradient {worst[1]:.2e} ({worst[0]}) over {n} tensors")


def test_c5_canonical_batch_forward_vs_oracle_full_size():
    """BASELINE configs[4] (liu2021 prosody + learn_alignment) at the canonical batch, train mode (prosody encoders read the mel,
    BatchNorm batch statistics), dropout off: the GRU over ~1,000 mel steps, the Conv2d stack on [16, Tm, 80], the aligner's
    671 MB-class distance map and the device MAS against the CPU oracle."""
    from ctts_amd.synthetic import make_unsup_batch
    torch.manual_seed(3)
    pre, mc, tc = get_configs()
    mc["prosody_modeling"]["model_type"] = "liu2021"
    mc["duration_modeling"]["learn_alignment"] = True
    m = ctts_amd.CompTransTTS(pre, mc, tc).to(DEV)
    m.train()
    no_dropout(m)
    sd = {k: v.detach().cpu().contiguous().clone() for k, v in m.state_dict().items()}
    batch = make_unsup_batch()
    args = list(as_model_args(to_device(batch, DEV)))
    args[7] = dict(args[7])
    with torch.no_grad():
        out = m(*args, step=100001)
    a = list(as_model_args(batch))
    a[7] = dict(a[7])
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    with torch.no_grad():
        ref = R.comp_trans_tts_forward(sd, mc, pre, *a, step=100001, training=True, train_dropout=False, new_stats={})
        # the device MAS against the host MAS on IDENTICAL input (the product's own soft alignment): bit-exact
        hard_on_product_soft = R.binarize_attention(out[10][0].cpu(), batch["src_lens"], batch["mel_lens"])
    assert torch.equal(out[10][1].cpu(), hard_on_product_soft)
    e_soft = maxerr(out[10][0], ref[10][0].numpy())
    e_logp = maxerr(out[10][3], ref[10][3].numpy())
    assert e_soft <= 1e-4 and e_logp <= 2e-3, (e_soft, e_logp)
    same_path = torch.equal(out[5].cpu().to(ref[5].dtype), ref[5])
    e_mel, e_post, e_logd = maxerr(out[0], ref[0].numpy()), maxerr(out[1], ref[1].numpy()), maxerr(out[4], ref[4].numpy())
    print(f"C5 canonical batch: attn_soft {e_soft:.2e} logprob {e_logp:.2e} durations identical {same_path}; "
          f"mel {e_mel:.2e} postnet {e_post:.2e} log_d {e_logd:.2e}")
    assert e_logd <= MEL_TOL                              # duration predictor does not depend on the alignment path
    # monotonic search on a [Tm, Ts] map of near-ties can legitimately pick another path after a 1e-6 change of the scores: the mel
    # comparison is only meaningful when both sides expanded with the same durations (they do on every box measured so far)
    assert same_path, "hard alignments differ: re-check MAS tie handling"
    assert e_mel <= MEL_TOL and e_post <= MEL_TOL
    for n, v, r in zip(("up_vec?", "pp?"), out[11] or (), ref[11] or ()):
        if v is not None and r is not None:
            assert maxerr(v, r.numpy()) <= 1e-3, n


def test_conformer_b4_t1000_forward_vs_oracle_full_length():
    """BASELINE configs[2] geometry at full length: 4 utterances, decoder T = 1000 (the max_seq_len crop), 512 MB-class attention
    maps in the reference - relative-shift index arithmetic and the unmasked softmax at large T against the CPU oracle."""
    from ctts_amd.synthetic import CANONICAL_SRC_LENS
    torch.manual_seed(2)
    m, (pre, mc, tc) = build(block="conformer")
    m.eval()
    batch = make_batch(CANONICAL_SRC_LENS[:4], max_mel_cap=1000)
    with torch.no_grad():
        out = m(*as_model_args(to_device(batch, DEV)))
    sd = {k: v.detach().cpu().contiguous() for k, v in m.state_dict().items()}
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    with torch.no_grad():
        ref = R.comp_trans_tts_forward_conformer(sd, mc, pre, *as_model_args(batch), training=False)
    e_mel, e_post = maxerr(out[0], ref[0].numpy()), maxerr(out[1], ref[1].numpy())
    print(f"conformer B=4 T={out[0].shape[1]}: mel max-abs {e_mel:.2e}, postnet mel {e_post:.2e}")
    assert out[0].shape[1] == 1000 and e_mel <= MEL_TOL and e_post <= MEL_TOL


def test_train_step_with_dropout_runs_and_is_finite():
    torch.manual_seed(2)
    m, _ = build()
    m.train()
    batch = make_batch([40, 33, 21, 12], 8, seed=9)
    out = m(*as_model_args(to_device(batch, DEV)))
    loss = (out[0].abs().mean() + out[1].abs().mean() + out[4].pow(2).mean() + out[3].pow(2).mean() + out[2]["cwt"].abs().mean()
            + out[2]["f0_mean"].abs().mean() + out[2]["f0_std"].abs().mean())
    loss.backward()
    tot = 0.0
    for k, p in m.named_parameters():
        if p.requires_grad:
            assert p.grad is not None, k
            assert torch.isfinite(p.grad).all(), k
            tot += float(p.grad.pow(2).sum())
    assert np.isfinite(tot) and tot > 0
    out2 = m(*as_model_args(to_device(batch, DEV)))
    assert not torch.equal(out[0], out2[0])          # fresh dropout mask per step


def test_product_fails_loudly_on_cpu_tensors():
    pre, mc, tc = get_configs()
    m = ctts_amd.CompTransTTS(pre, mc, tc)           # CPU module: no fallback path exists
    with pytest.raises(Exception):
        m(*as_model_args(make_batch([8, 5], 4)))


def test_g4_conformer_eval_matches_reference():
    g = load_golden("g4_conformer_eval")
    m, _ = build(sd=closed_form_sd("LJSpeech", "conformer"), block="conformer")
    m.eval()
    with torch.no_grad():
        out = m(*args_from(batch_from_golden(g)))
    check_against_golden(out, g)


def test_g4_conformer_train_gradients_match_reference():
    g = load_golden("g4_conformer_train_nodrop")
    m, _ = build(sd=closed_form_sd("LJSpeech", "conformer"), block="conformer")
    m.train()
    no_dropout(m)
    out = m(*args_from(batch_from_golden(g)))
    check_against_golden(out, g)
    sd_after = m.state_dict()
    for k in sd_after:
        if "running_" in k:
            assert maxerr(sd_after[k], g["bn." + k]) < 1e-4, k

    def pseudo(name, shape):
        return torch.from_numpy(_hash_uniform("probe." + name, int(np.prod(shape))).reshape(shape)).float().to(DEV)
    mel, post, p_pred, e_pred, log_d = out[:5]
    loss = ((post * pseudo("post", post.shape)).sum() + (mel * pseudo("mel", mel.shape)).sum()
            + (log_d * pseudo("logd", log_d.shape)).sum() + (e_pred * pseudo("e", e_pred.shape)).sum()
            + (p_pred["cwt"] * pseudo("cwt", p_pred["cwt"].shape)).sum()
            + (p_pred["f0_mean"] * 0.7).sum() + (p_pred["f0_std"] * -0.3).sum())
    loss.backward()
    worst, n = ("", 0.0), 0
    for k, p in m.named_parameters():
        if "grad.stat." + k not in g:
            continue
        gs = g["grad.stat." + k]
        gr = p.grad.flatten() if p.grad is not None else torch.zeros(p.numel(), device=DEV)
        scale = max(1.0, float(gs[1]))
        e = max(maxerr(gr[:64], g["grad.head." + k]) / scale, abs(float(gr.double().pow(2).sum().sqrt()) - gs[1]) / scale)
        if e > worst[1]:
            worst = (k, e)
        n += 1
    print("conformer worst relative gradient error:", worst, "over", n, "parameters")
    assert n > 300 and worst[1] < 2e-3, worst


def test_conformer_train_step_with_dropout_runs():
    torch.manual_seed(3)
    m, _ = build(block="conformer")
    m.train()
    batch = make_batch([40, 33, 21, 12], 8, seed=9)
    out = m(*as_model_args(to_device(batch, DEV)))
    loss = (out[0].abs().mean() + out[1].abs().mean() + out[4].pow(2).mean() + out[3].pow(2).mean() + out[2]["cwt"].abs().mean()
            + out[2]["f0_mean"].abs().mean() + out[2]["f0_std"].abs().mean())
    loss.backward()
    for k, p in m.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), k


def test_grad_accumulation_fusion_matches_plain_autograd():
    """ops.set_grad_accumulation_fusion: kernels add straight into param.grad (flat DP arena) - same gradients."""
    from ctts_amd import ops
    from ctts_amd.dp import FlatGradArena
    torch.manual_seed(5)
    m, _ = build()
    m.train()
    no_dropout(m)
    batch = make_batch([40, 33, 21, 12], 8, seed=9)

    def run(fuse, side_stream=False):
        arena = FlatGradArena(m.parameters())
        ops.set_grad_accumulation_fusion(fuse)
        ops.set_wgrad_stream(torch.cuda.Stream() if side_stream else None)
        try:
            for bn in [x for x in m.modules() if hasattr(x, "running_mean")]:
                bn.running_mean.zero_(); bn.running_var.fill_(1.0)
            out = m(*as_model_args(to_device(batch, DEV)))
            loss = (out[0].abs().mean() + out[1].abs().mean() + out[4].pow(2).mean() + out[3].pow(2).mean()
                    + out[2]["cwt"].abs().mean() + out[2]["f0_mean"].abs().mean() + out[2]["f0_std"].abs().mean())
            loss.backward()
        finally:
            ops.set_grad_accumulation_fusion(False)
            ops.set_wgrad_stream(None)
        return arena.flat.clone()
    g0, g1, g2 = run(False), run(True), run(True, side_stream=True)   # g2: wgrad GEMMs on the side stream, joined by the engine callback
    assert g0.abs().sum() > 0
    for gx in (g1, g2):
        err = (g0 - gx).abs().max().item()
        assert err <= 1e-5 * max(1.0, g0.abs().max().item()), err


def test_conformer_fused_accumulation_in_flat_arenas_matches_plain_autograd():
    """conformer with parameters re-homed into dp.FlatAdam's flat arena (what trainer.TrainStep does): q / k / v weights are then
    neighbours in memory and ops.linear_packed runs the stacked [768,256] projection on VIEWS (no cat), accumulating its weight
    gradient straight into the flat gradient arena; pointwise-conv weights (views of [Cout,Cin,1] leaves), BatchNorm parameters and
    the depthwise-conv weight accumulate in place as well.  Same gradients as plain autograd."""
    from ctts_amd import ops
    from ctts_amd.dp import FlatGradArena, FlatAdam
    torch.manual_seed(6)
    m, _ = build(block="conformer")
    m.train()
    no_dropout(m)
    arena = FlatGradArena(m.named_parameters())
    FlatAdam(arena, 1e-3)                                   # re-homes every parameter into one flat buffer
    at = getattr(getattr(m.decoder.layer_stack, "0").sequential, "1").module.attention
    qkv = [at.query_proj.linear.weight, at.key_proj.linear.weight, at.value_proj.linear.weight]
    assert ops._adjacent([w.detach() for w in qkv]) and ops._adjacent([w.grad for w in qkv])
    batch = make_batch([40, 33, 21, 12], 8, seed=9)

    def run(fuse):
        arena.zero_()
        ops.set_grad_accumulation_fusion(fuse)
        try:
            for bn in [x for x in m.modules() if hasattr(x, "running_mean")]:
                bn.running_mean.zero_(); bn.running_var.fill_(1.0)
            out = m(*as_model_args(to_device(batch, DEV)))
            loss = (out[0].abs().mean() + out[1].abs().mean() + out[4].pow(2).mean() + out[3].pow(2).mean()
                    + out[2]["cwt"].abs().mean() + out[2]["f0_mean"].abs().mean() + out[2]["f0_std"].abs().mean())
            loss.backward()
        finally:
            ops.set_grad_accumulation_fusion(False)
        arena.check_bound()
        return arena.flat.clone()
    g0, g1 = run(False), run(True)
    assert g0.abs().sum() > 0
    err = (g0 - g1).abs().max().item()
    assert err <= 2e-5 * max(1.0, g0.abs().max().item()), err


@pytest.mark.parametrize("gname,step", [("g6_unsup_soft_step100", 100), ("g6_unsup_hard_step60000", 60000)])
def test_g6_unsupervised_alignment_matches_reference(gname, step):
    """learn_alignment=True (the reference's default yaml): AlignmentEncoder + MAS on the device + soft/hard upsampling."""
    g = load_golden(gname)
    pre, mc, tc = get_configs()
    mc["duration_modeling"]["learn_alignment"] = True
    m = ctts_amd.CompTransTTS(pre, mc, tc)
    m.load_state_dict(closed_form_sd(unsup=True))
    m = m.to(DEV)
    m.train()
    no_dropout(m)
    b = to_device(batch_from_golden(g), DEV)
    out = m(b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"], b["mel_lens"], b["max_mel_len"], b["p_targets"],
            b["e_targets"], None, b["attn_priors"], None, step=step)
    a_soft, a_hard, a_dur, a_logp = out[10]
    assert maxerr(a_soft, g["out.attn_soft"]) < 1e-5 and maxerr(a_logp, g["out.attn_logprob"]) < 1e-3
    assert np.array_equal(a_hard.cpu().numpy(), g["out.attn_hard"])          # MAS: bit-exact hard alignment
    assert np.array_equal(a_dur.cpu().numpy(), g["out.attn_hard_dur"])
    assert np.array_equal(out[12]["mel2ph"].cpu().numpy(), g["out.pt_mel2ph"])
    assert maxerr(out[13], g["out.e_targets_out"]) < 1e-5
    for name, i in (("mel", 0), ("postnet_mel", 1), ("log_d", 4), ("e_pred", 3)):
        assert maxerr(out[i], g["out." + name]) <= MEL_TOL, name

    def pseudo(name, shape):
        return torch.from_numpy(_hash_uniform("probe." + name, int(np.prod(shape))).reshape(shape)).float().to(DEV)
    mel, post, p_pred, e_pred, log_d = out[:5]
    loss = ((post * pseudo("post", post.shape)).sum() + (mel * pseudo("mel", mel.shape)).sum()
            + (log_d * pseudo("logd", log_d.shape)).sum() + (e_pred * pseudo("e", e_pred.shape)).sum()
            + (p_pred["cwt"] * pseudo("cwt", p_pred["cwt"].shape)).sum()
            + (p_pred["f0_mean"] * 0.7).sum() + (p_pred["f0_std"] * -0.3).sum()
            + (a_soft * pseudo("asoft", a_soft.shape)).sum() * 10 + (a_logp * pseudo("alogp", a_logp.shape)).sum() * 0.1)
    loss.backward()
    worst, n = ("", 0.0), 0
    for k, p in m.named_parameters():
        if "grad.stat." + k not in g:
            continue
        gs = g["grad.stat." + k]
        gr = p.grad.flatten() if p.grad is not None else torch.zeros(p.numel(), device=DEV)
        scale = max(1.0, float(gs[1]))
        e = max(maxerr(gr[:64], g["grad.head." + k]) / scale, abs(float(gr.double().pow(2).sum().sqrt()) - gs[1]) / scale)
        if e > worst[1]:
            worst = (k, e)
        n += 1
    print(f"unsup step {step}: worst relative gradient error", worst, "over", n)
    assert n > 170 and worst[1] < 2e-3, worst


def test_mas_kernel_vs_oracle_random():
    """MAS on random soft attentions (ragged lengths) against the numpy restatement of mas_width1."""
    from ctts_amd import ops
    g = torch.Generator().manual_seed(21)
    B, Tm, Ts = 5, 97, 23
    attn = torch.softmax(torch.randn(B, 1, Tm, Ts, generator=g) * 3, -1)
    in_lens = torch.tensor([23, 17, 5, 1, 23])
    out_lens = torch.tensor([97, 60, 40, 9, 23])
    hard, dur = ops.mas_binarize(attn.to(DEV), in_lens.to(DEV), out_lens.to(DEV))
    ref = R.binarize_attention(attn, in_lens, out_lens)
    assert np.array_equal(hard.cpu().numpy(), ref.numpy())
    assert np.array_equal(dur.cpu().numpy(), ref.sum(2)[:, 0].numpy())
    assert torch.equal(dur.sum(1).cpu(), out_lens.float())      # every valid frame is assigned to exactly one phoneme


@pytest.mark.parametrize("Tm,Ts", [(1024, 128), (700, 300), (2100, 300)])
def test_mas_kernel_paths_large(Tm, Ts):
    """LDS-bitmask fast path with 1 and 2 columns per thread, and the global-memory fallback (Tq * Tk bits beyond 60 KB of LDS)."""
    from ctts_amd import ops
    g = torch.Generator().manual_seed(Tm + Ts)
    B = 2
    attn = torch.softmax(torch.randn(B, 1, Tm, Ts, generator=g) * 2, -1)
    in_lens = torch.tensor([Ts, Ts - 37])
    out_lens = torch.tensor([Tm, Tm - 211])
    hard, dur = ops.mas_binarize(attn.to(DEV), in_lens.to(DEV), out_lens.to(DEV))
    ref = R.binarize_attention(attn, in_lens, out_lens)
    assert np.array_equal(hard.cpu().numpy(), ref.numpy())
    assert np.array_equal(dur.cpu().numpy(), ref.sum(2)[:, 0].numpy())


PROS_NAMES = ("up_emb", "pp_emb", "up_vec", "pp_vec", "pp_attn")


@pytest.mark.parametrize("gname,unsup,training", [("g10_liu2021_eval", False, False), ("g10_liu2021_train_nodrop", False, True),
                                                  ("g10_liu2021_unsup_step60000", True, True)])
def test_g10_liu2021_prosody_matches_reference(gname, unsup, training):
    """SURVEY a17 (and config C5 = liu2021 + learn_alignment): CoordConv2d stack + BN2d + GRU reference encoders, STL, phoneme-level
    attention and bi-GRU predictors on the HIP kernels against the live-reference goldens, outputs and gradients."""
    g = load_golden(gname)
    pre, mc, tc = get_configs()
    mc["duration_modeling"]["learn_alignment"] = unsup
    mc["prosody_modeling"]["model_type"] = "liu2021"
    m = ctts_amd.CompTransTTS(pre, mc, tc)
    m.load_state_dict(closed_form_sd(unsup=unsup, prosody="liu2021"))
    m = m.to(DEV)
    m.train(training)
    no_dropout(m)
    b = to_device(batch_from_golden(g), DEV)
    out = m(b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"], b["mel_lens"], b["max_mel_len"], b["p_targets"],
            b["e_targets"], b["d_targets"], b["attn_priors"] if unsup else None, None, step=60000 if unsup else None)
    for n, v in zip(PROS_NAMES, out[11]):
        if v is None:
            assert "out.pros." + n not in g, n
        else:
            e = maxerr(v, g["out.pros." + n])
            print("prosody", n, f"{e:.2e}")
            assert e <= 2e-4, (n, e)
    for name, i in (("mel", 0), ("postnet_mel", 1), ("log_d", 4), ("e_pred", 3)):
        assert maxerr(out[i], g["out." + name]) <= MEL_TOL, name
    if not training:
        return
    sd = m.state_dict()
    for k in g:
        if k.startswith("bn.") and "prosody" in k:
            assert maxerr(sd[k[3:]], g[k]) <= 1e-4, k

    def pseudo(name, shape):
        return torch.from_numpy(_hash_uniform("probe." + name, int(np.prod(shape))).reshape(shape)).float().to(DEV)
    mel, post, p_pred, e_pred, log_d = out[:5]
    loss = ((post * pseudo("post", post.shape)).sum() + (mel * pseudo("mel", mel.shape)).sum()
            + (log_d * pseudo("logd", log_d.shape)).sum() + (e_pred * pseudo("e", e_pred.shape)).sum()
            + (p_pred["cwt"] * pseudo("cwt", p_pred["cwt"].shape)).sum()
            + (p_pred["f0_mean"] * 0.7).sum() + (p_pred["f0_std"] * -0.3).sum())
    if unsup:
        a_soft, _, _, a_logp = out[10]
        loss = loss + (a_soft * pseudo("asoft", a_soft.shape)).sum() * 10 + (a_logp * pseudo("alogp", a_logp.shape)).sum() * 0.1
    loss = loss + (out[11][2] * pseudo("upvec", out[11][2].shape)).sum() + (out[11][3] * pseudo("ppvec", out[11][3].shape)).sum()
    loss.backward()
    worst, n = ("", 0.0), 0
    for k, p in m.named_parameters():
        if "grad.stat." + k not in g:
            continue
        gs = g["grad.stat." + k]
        gr = p.grad.flatten() if p.grad is not None else torch.zeros(p.numel(), device=DEV)
        scale = max(1.0, float(gs[1]))
        e = max(maxerr(gr[:64], g["grad.head." + k]) / scale, abs(float(gr.double().pow(2).sum().sqrt()) - gs[1]) / scale)
        # one BN2d output of the unsup fixture sits on the ReLU kink (see tests/test_oracle_golden.py): looser there
        tol = 3e-2 if (unsup and "phoneme_prosody_encoder.encoder." in k) else 2e-3
        assert e < tol, (k, e)
        if e > worst[1]:
            worst = (k, e)
        n += 1
    print(f"{gname}: worst relative gradient error", worst, "over", n)
    assert n > 250, n


def test_packed_batch_upload_and_prefetcher_feed_the_model():
    """one pinned buffer -> one async H2D -> zero-copy views; the Prefetcher yields reference-style batches in order and
    the model trains on them (batch[2:] positional, as train.py:106)."""
    from ctts_amd import data as D
    from tests.util import synthetic_samples
    samples = synthetic_samples(10, 5, False)
    batches = D.collate(samples, 4, sort=True)
    host = [D.PackedBatch.pack(b).host_views() for b in batches]
    got = list(D.Prefetcher(batches, DEV, depth=2))
    assert len(got) == len(batches)
    for h, d in zip(host, got):
        assert h[0] == d[0] and h[5] == d[5] and h[8] == d[8]
        for a, b_ in zip(h, d):
            if torch.is_tensor(a):
                assert b_.is_cuda and b_.dtype == a.dtype and torch.equal(a, b_.cpu())
        for k in h[9]:
            assert torch.equal(h[9][k], d[9][k].cpu())
    m, _ = build()
    m.train()
    batch = got[0]
    out = m(*batch[2:], step=1)                      # speakers, texts, src_lens, max_src_len, mels, mel_lens, max_mel_len, pitch dict, ...
    assert out[0].shape == batch[6].shape and torch.isfinite(out[1]).all()


def test_g12_vctk_multispeaker_unsupervised_matches_reference():
    """reference default VCTK yaml (multi_speaker + learn_alignment): aligner speaker projections, device MAS, gradients."""
    g = load_golden("g12_vctk_unsup_step60000")
    pre, mc, tc = get_configs("VCTK")
    mc["duration_modeling"]["learn_alignment"] = True
    m = ctts_amd.CompTransTTS(pre, mc, tc)
    m.load_state_dict(closed_form_sd("VCTK", unsup=True))
    m = m.to(DEV)
    m.train()
    no_dropout(m)
    b = to_device(batch_from_golden(g), DEV)
    out = m(b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"], b["mel_lens"], b["max_mel_len"], b["p_targets"],
            b["e_targets"], None, b["attn_priors"], b["spker_embeds"], step=60000)
    a_soft, a_hard, a_dur, a_logp = out[10]
    assert maxerr(a_soft, g["out.attn_soft"]) < 1e-5 and maxerr(a_logp, g["out.attn_logprob"]) < 1e-3
    assert np.array_equal(a_hard.cpu().numpy(), g["out.attn_hard"]) and np.array_equal(a_dur.cpu().numpy(), g["out.attn_hard_dur"])
    for name, i in (("mel", 0), ("postnet_mel", 1), ("log_d", 4), ("e_pred", 3)):
        assert maxerr(out[i], g["out." + name]) <= MEL_TOL, name

    def pseudo(name, shape):
        return torch.from_numpy(_hash_uniform("probe." + name, int(np.prod(shape))).reshape(shape)).float().to(DEV)
    mel, post, p_pred, e_pred, log_d = out[:5]
    loss = ((post * pseudo("post", post.shape)).sum() + (mel * pseudo("mel", mel.shape)).sum()
            + (log_d * pseudo("logd", log_d.shape)).sum() + (e_pred * pseudo("e", e_pred.shape)).sum()
            + (p_pred["cwt"] * pseudo("cwt", p_pred["cwt"].shape)).sum()
            + (p_pred["f0_mean"] * 0.7).sum() + (p_pred["f0_std"] * -0.3).sum()
            + (a_soft * pseudo("asoft", a_soft.shape)).sum() * 10 + (a_logp * pseudo("alogp", a_logp.shape)).sum() * 0.1)
    loss.backward()
    n = 0
    for k, p in m.named_parameters():
        if "grad.stat." + k not in g:
            continue
        gs = g["grad.stat." + k]
        gr = p.grad.flatten() if p.grad is not None else torch.zeros(p.numel(), device=DEV)
        scale = max(1.0, float(gs[1]))
        e = max(maxerr(gr[:64], g["grad.head." + k]) / scale, abs(float(gr.double().pow(2).sum().sqrt()) - gs[1]) / scale)
        assert e < 2e-3, (k, e)
        n += 1
    assert n > 170 and any("spk_proj" in k for k in dict(m.named_parameters()) if "grad.stat." + k in g)


@pytest.mark.parametrize("src_lens,fpp", [([1], 1), ([1, 1], 3), ([2, 1, 3], 1), ([128], 8)])
def test_edge_shapes_vs_oracle(src_lens, fpp):
    """degenerate and extreme shapes: single phoneme / single frame / batch of one / one full-length utterance (train-mode BN
    statistics over as few as one row), HIP forward + backward against the CPU oracle."""
    torch.manual_seed(11)
    m, (pre, mc, tc) = build()
    m.train()
    no_dropout(m)
    batch = make_batch(src_lens, fpp, seed=5)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    out = m(*as_model_args(to_device(batch, DEV)))
    ref = R.comp_trans_tts_forward(sd, mc, pre, *as_model_args(batch), training=True)
    rows = batch["mels"].shape[0] * batch["mels"].shape[1]
    tol = MEL_TOL if rows > 4 else 5e-2          # BatchNorm over <= 4 rows divides by a near-zero variance: ill-conditioned in any arithmetic
    assert maxerr(out[0], ref[0].detach().numpy()) <= MEL_TOL
    assert maxerr(out[1], ref[1].detach().numpy()) <= tol
    assert torch.equal(out[9].cpu(), ref[9]) and torch.equal(out[7].cpu(), ref[7])
    (out[0].abs().mean() + out[1].abs().mean() + out[4].pow(2).mean()).backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


def test_conformer_decoder_crops_to_max_seq_len_in_training():
    """conformer.py:148-154: the decoder (and its mask) is cropped to max_seq_len in training; the returned mel is shorter than
    the padded targets - reproduced with max_seq_len shrunk to 40 so that the crop is exercised on a small batch."""
    torch.manual_seed(12)
    pre, mc, tc = get_configs()
    mc["block_type"] = "conformer"
    mc["max_seq_len"] = 40
    m = ctts_amd.CompTransTTS(pre, mc, tc).to(DEV)
    m.train()
    no_dropout(m)
    batch = make_batch([9, 7], 6, seed=6)            # mel lengths 54 / 42 > 40
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    out = m(*as_model_args(to_device(batch, DEV)))
    ref = R.comp_trans_tts_forward_conformer(sd, mc, pre, *as_model_args(batch), training=True)
    assert out[0].shape[1] == 40 == ref[0].shape[1] and out[7].shape[1] == 40
    assert maxerr(out[0], ref[0].detach().numpy()) <= MEL_TOL and maxerr(out[1], ref[1].detach().numpy()) <= MEL_TOL


@pytest.mark.parametrize("gname,lname,unsup,prosody", [
    ("g2_fs2_train_nodrop", "g9_loss", False, "none"),
    ("g6_unsup_hard_step60000", "g6_unsup_loss_step60000", True, "none"),
    ("g10_liu2021_train_nodrop", "g10_liu2021_loss", False, "liu2021"),
    ("g10_liu2021_unsup_step60000", "g10_liu2021_unsup_loss_step100001", True, "liu2021")])
def test_product_loss_matches_reference_goldens(gname, lname, unsup, prosody):
    """The device-only CompTransTTSLoss (fused mel-L1 pair, device ForwardSum, Bin and prosody terms) on the product model's own outputs
    against the 9-tuples captured from the reference's model/loss.py (goldens G9, G6-loss, G10-loss)."""
    from ctts_amd.loss import CompTransTTSLoss
    g, gl = load_golden(gname), load_golden(lname)
    pre, mc, tc = get_configs()
    mc["duration_modeling"]["learn_alignment"] = unsup
    mc["prosody_modeling"]["model_type"] = prosody
    m = ctts_amd.CompTransTTS(pre, mc, tc)
    m.load_state_dict(closed_form_sd(unsup=unsup, prosody=prosody))
    m = m.to(DEV)
    m.train()
    no_dropout(m)
    b = to_device(batch_from_golden(g), DEV)
    args = [b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"], b["mel_lens"], b["max_mel_len"], b["p_targets"],
            b["e_targets"], b["d_targets"], b["attn_priors"] if unsup else None, None]
    step_fwd = 60000 if unsup else None
    out = m(*args, step=step_fwd)
    inputs = [None, None] + list(args)
    inputs[9:11] = out[-2:]
    L = CompTransTTSLoss(pre, mc, tc).to(DEV)
    step = int(gl["step"])
    total, mel, post, pitch, energy, dur, ctc, binl, pros = L(inputs, out[:-2], step)
    got = {"total": total, "mel": mel, "postnet_mel": post, "energy": energy, "pitch.C": pitch["C"], "pitch.uv": pitch["uv"],
           "duration.pdur": dur["pdur"], "duration.wdur": dur["wdur"], "duration.sdur": dur["sdur"], "ctc": ctc, "bin": binl, "prosody": pros}
    for k, v in got.items():
        if "loss." + k not in gl:
            continue
        ref = float(np.asarray(gl["loss." + k]).reshape(-1)[0])
        val = float(v.reshape(-1)[0])
        assert abs(val - ref) <= 5e-4 * max(1.0, abs(ref)), (k, val, ref)
    total.backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    with pytest.raises(Exception):
        L([None, None] + [a.cpu() if torch.is_tensor(a) else a for a in args], tuple(o.cpu() if torch.is_tensor(o) else o for o in out[:-2]),
          step)


@pytest.mark.parametrize("gname,suffix,vp,training", [("g13_swish_left_eval", "_swish_left", dict(ffn_act="swish", ffn_padding="LEFT"), False),
                                                      ("g13_swish_left_train_nodrop", "_swish_left", dict(ffn_act="swish", ffn_padding="LEFT"), True),
                                                      ("g13_relu_train_nodrop", "_relu", dict(ffn_act="relu"), True)])
def test_g13_ffn_act_and_padding_switches_match_reference(gname, suffix, vp, training):
    """VERDICT r03 missing #5: variance_predictor.ffn_act = swish / relu and ffn_padding = LEFT no longer raise - the FFN epilogue takes the
    activation code, every Conv1d of the FFNs and predictors takes the causal padding (ops.conv1d(padding="LEFT"): k-1 zeros in front;
    data gradient with pad 0, weight gradient with pad k-1), and the LEFT model exposes the reference's state-dict keys (ffn_1.1.weight).
    Outputs, BatchNorm statistics and all parameter gradients against the live reference's golden vectors."""
    g = load_golden(gname)
    sd = closed_form_sd(suffix=suffix)
    m, _ = build(sd=sd, vp=vp)
    assert sorted(m.state_dict().keys()) == sorted(sd.keys())
    if not training:
        m.eval()
        with torch.no_grad():
            out = m(*args_from(batch_from_golden(g)))
        check_against_golden(out, g)
        return
    m.train()
    no_dropout(m)
    out = m(*args_from(batch_from_golden(g)))
    check_against_golden(out, g)

    def pseudo(name, shape):
        return torch.from_numpy(_hash_uniform("probe." + name, int(np.prod(shape))).reshape(shape)).float().to(DEV)
    mel, post, p_pred, e_pred, log_d = out[:5]
    loss = ((post * pseudo("post", post.shape)).sum() + (mel * pseudo("mel", mel.shape)).sum()
            + (log_d * pseudo("logd", log_d.shape)).sum() + (e_pred * pseudo("e", e_pred.shape)).sum()
            + (p_pred["cwt"] * pseudo("cwt", p_pred["cwt"].shape)).sum()
            + (p_pred["f0_mean"] * 0.7).sum() + (p_pred["f0_std"] * -0.3).sum())
    assert abs(loss.item() - float(g["grad.loss"])) < 5e-2
    loss.backward()
    worst, n = ("", 0.0), 0
    for k, p in m.named_parameters():
        if "grad.stat." + k not in g:
            continue
        gs = g["grad.stat." + k]
        gr = p.grad.flatten() if p.grad is not None else torch.zeros(p.numel(), device=DEV)
        scale = max(1.0, float(gs[1]))
        e = max(maxerr(gr[:64], g["grad.head." + k]) / scale, abs(float(gr.double().pow(2).sum().sqrt()) - gs[1]) / scale)
        if e > worst[1]:
            worst = (k, e)
        n += 1
    print("worst relative gradient error:", worst, "over", n, "parameters")
    assert n > 150 and worst[1] < 4e-3, worst


@pytest.mark.parametrize("gname,lname,suffix,ve", [("g14_noembed_train_nodrop", "g14_noembed_loss", "_noembed", dict(use_pitch_embed=False, use_energy_embed=False)),
                                                   ("g14_nopitch_train_nodrop", "g14_nopitch_loss", "_nopitch", dict(use_pitch_embed=False)),
                                                   ("g14_noembed_eval", None, "_noembed", dict(use_pitch_embed=False, use_energy_embed=False))])
def test_g14_pitch_and_energy_embedding_switches_match_reference(gname, lname, suffix, ve):
    """VERDICT r03 missing #5: variance_embedding.use_pitch_embed / use_energy_embed = False no longer raise - the branch, its
    parameters (state-dict keys as the reference's) and its prediction (None in the 14-tuple) disappear; CompTransTTSLoss keeps the
    reference's initial zeros for that term (loss.py:328-335).  Forward, BatchNorm statistics, parameter gradients and the loss 9-tuple
    against the live reference's golden vectors."""
    from ctts_amd.loss import CompTransTTSLoss
    g = load_golden(gname)
    sd = closed_form_sd(suffix=suffix)
    pre, mc, tc = get_configs()
    mc["variance_embedding"].update(ve)
    m = ctts_amd.CompTransTTS(pre, mc, tc)
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd)
    m = m.to(DEV)
    training = "train" in gname
    b = batch_from_golden(g)
    args = args_from(b)
    if not training:
        m.eval()
        with torch.no_grad():
            out = m(*args)
    else:
        m.train()
        no_dropout(m)
        out = m(*args)
    mel, post, p_pred, e_pred, log_d = out[:5]
    assert p_pred is None and (e_pred is None) == (not ve.get("use_energy_embed", True))
    for name, a, key in (("mel", mel, "out.mel"), ("postnet_mel", post, "out.postnet_mel"), ("log_d", log_d, "out.log_d")):
        assert maxerr(a, g[key]) <= MEL_TOL, (name, maxerr(a, g[key]))
    if e_pred is not None:
        assert maxerr(e_pred, g["out.e_pred"]) <= MEL_TOL
    assert np.array_equal(out[9].cpu().numpy(), g["out.mel_lens"])
    if not training:
        return

    def pseudo(name, shape):
        return torch.from_numpy(_hash_uniform("probe." + name, int(np.prod(shape))).reshape(shape)).float().to(DEV)
    loss = (post * pseudo("post", post.shape)).sum() + (mel * pseudo("mel", mel.shape)).sum() + (log_d * pseudo("logd", log_d.shape)).sum()
    if e_pred is not None:
        loss = loss + (e_pred * pseudo("e", e_pred.shape)).sum()
    assert abs(loss.item() - float(g["grad.loss"])) < 5e-2
    loss.backward(retain_graph=True)
    worst, n = ("", 0.0), 0
    for k, p in m.named_parameters():
        if "grad.stat." + k not in g:
            continue
        gs = g["grad.stat." + k]
        gr = p.grad.flatten() if p.grad is not None else torch.zeros(p.numel(), device=DEV)
        scale = max(1.0, float(gs[1]))
        e = max(maxerr(gr[:64], g["grad.head." + k]) / scale, abs(float(gr.double().pow(2).sum().sqrt()) - gs[1]) / scale)
        if e > worst[1]:
            worst = (k, e)
        n += 1
    print("worst relative gradient error:", worst, "over", n, "parameters")
    assert n > 90 and worst[1] < 4e-3, worst
    # the loss 9-tuple
    gl = load_golden(lname)
    inputs = [None, None] + list(args)
    inputs[9:11] = out[-2:]
    L = CompTransTTSLoss(pre, mc, tc).to(DEV)
    total, mel_l, post_l, pitch, energy, dur, ctc, binl, pros = L(inputs, out[:-2], int(gl["step"]))
    got = {"total": total, "mel": mel_l, "postnet_mel": post_l, "energy": energy, "pitch.C": pitch["C"], "pitch.uv": pitch["uv"],
           "pitch.f0_mean": pitch["f0_mean"], "pitch.f0_std": pitch["f0_std"],
           "duration.pdur": dur["pdur"], "duration.wdur": dur["wdur"], "duration.sdur": dur["sdur"]}
    for k, v in got.items():
        ref = float(np.asarray(gl["loss." + k]).reshape(-1)[0])
        val = float(v.reshape(-1)[0])
        assert abs(val - ref) <= 5e-4 * max(1.0, abs(ref)), (k, val, ref)
    m.zero_grad()
    total.backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


@pytest.mark.parametrize("case,mode", [("g15_pitch_frame", "eval"), ("g15_pitch_frame", "train_nodrop"), ("g15_pitch_frame", "infer"),
                                       ("g15_pitch_frame_std_nouv", "train_nodrop"), ("g15_pitch_ph", "eval"),
                                       ("g15_pitch_ph", "train_nodrop"), ("g15_pitch_ph", "infer"),
                                       ("g16_energy_frame", "eval"), ("g16_energy_frame", "train_nodrop")])
def test_g15_g16_pitch_type_and_energy_level_switches_match_reference(case, mode):
    """VERDICT r03 missing #5, the last two switch families: preprocessing.pitch.pitch_type "frame" / "ph" (+ pitch_norm standard,
    use_uv False, pitch_loss l2) and preprocessing.energy.feature "frame_level" (modules.py:777-785,892-938,1083-1094; loss.py:173-178,
    202-219,238-242) no longer raise.  Forward, predictions, parameter gradients and the loss 9-tuple against the live reference."""
    from ctts_amd.loss import CompTransTTSLoss
    from tests.util import switch_configs
    g = load_golden(f"{case}_{mode}")
    (pre, mc, tc), sd = switch_configs(case)
    m = ctts_amd.CompTransTTS(pre, mc, tc)
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd)
    m = m.to(DEV)
    b = batch_from_golden(g)
    args = args_from(b)
    if mode == "train_nodrop":
        m.train()
        no_dropout(m)
        out = m(*args)
    else:
        m.eval()
        kw = dict(p_control=1.1, e_control=0.9, d_control=2.0) if mode == "infer" else {}
        with torch.no_grad():
            out = m(*args, **kw)
    mel, post, p_pred, e_pred, log_d = out[:5]
    for name, a, key in (("mel", mel, "out.mel"), ("postnet_mel", post, "out.postnet_mel"), ("log_d", log_d, "out.log_d"),
                         ("e_pred", e_pred, "out.e_pred"), ("f0_denorm", p_pred["f0_denorm"], "out.f0_denorm")):
        tol = MEL_TOL * (40 if name == "f0_denorm" else 1)            # f0_denorm = 2 ** f0 is O(200)
        assert a.shape == g[key].shape and maxerr(a, g[key]) <= tol, (name, maxerr(a, g[key]))
    assert np.array_equal(out[9].cpu().numpy(), g["out.mel_lens"])
    if case.startswith("g15"):
        assert p_pred["cwt"] is None and p_pred["pitch_pred"].shape == g["out.pitch_pred"].shape
        assert maxerr(p_pred["pitch_pred"], g["out.pitch_pred"]) <= MEL_TOL
    if mode != "infer" and "out.pt_f0" in g:
        assert maxerr(out[-2]["f0"], g["out.pt_f0"]) <= 1e-5
    if mode != "train_nodrop":
        return

    def pseudo(name, shape):
        return torch.from_numpy(_hash_uniform("probe." + name, int(np.prod(shape))).reshape(shape)).float().to(DEV)
    loss = ((post * pseudo("post", post.shape)).sum() + (mel * pseudo("mel", mel.shape)).sum() + (log_d * pseudo("logd", log_d.shape)).sum()
            + (e_pred * pseudo("e", e_pred.shape)).sum())
    if p_pred["cwt"] is not None:
        loss = loss + ((p_pred["cwt"] * pseudo("cwt", p_pred["cwt"].shape)).sum() + (p_pred["f0_mean"] * 0.7).sum()
                       + (p_pred["f0_std"] * -0.3).sum())
    else:
        loss = loss + (p_pred["pitch_pred"] * pseudo("ppred", p_pred["pitch_pred"].shape)).sum()
    assert abs(loss.item() - float(g["grad.loss"])) < 5e-2
    loss.backward(retain_graph=True)
    worst, n = ("", 0.0), 0
    for k, p in m.named_parameters():
        if "grad.stat." + k not in g:
            continue
        gs = g["grad.stat." + k]
        gr = p.grad.flatten() if p.grad is not None else torch.zeros(p.numel(), device=DEV)
        scale = max(1.0, float(gs[1]))
        e = max(maxerr(gr[:64], g["grad.head." + k]) / scale, abs(float(gr.double().pow(2).sum().sqrt()) - gs[1]) / scale)
        if e > worst[1]:
            worst = (k, e)
        n += 1
    print("worst relative gradient error:", worst, "over", n, "parameters")
    assert n > 130 and worst[1] < 4e-3, worst
    gl = load_golden(f"{case}_loss")
    inputs = [None, None] + list(args)
    inputs[9:11] = out[-2:]
    L = CompTransTTSLoss(pre, mc, tc).to(DEV)
    losses = L(inputs, out[:-2], int(gl["step"]))
    got = {"total": losses[0], "mel": losses[1], "postnet_mel": losses[2], "energy": losses[4]}
    got.update({"pitch." + k: v for k, v in losses[3].items()})
    got.update({"duration." + k: v for k, v in losses[5].items()})
    assert {k for k in gl if k.startswith("loss.pitch.")} == {"loss.pitch." + k for k in losses[3]}
    for k, v in got.items():
        ref = float(np.asarray(gl["loss." + k]).reshape(-1)[0])
        val = float(v.reshape(-1)[0])
        assert abs(val - ref) <= 5e-4 * max(1.0, abs(ref)), (k, val, ref)
    m.zero_grad()
    losses[0].backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    # the initial zeros before var_start_steps keep the keys of this pitch_type (loss.py:241-264)
    early = L(inputs, [o.detach() if torch.is_tensor(o) else o for o in out[:-2]], 10)
    assert set(early[3]) == set(losses[3]) and all(float(v) == 0 for v in early[3].values())


@pytest.mark.parametrize("case", ["g15_pitch_frame", "g15_pitch_ph", "g16_energy_frame"])
def test_switch_configurations_train_in_a_captured_step_bit_reproducibly(case):
    """The pitch_type / energy-level variants run inside TrainStep's hipGraph (no host sync on their paths: `capture` raises
    otherwise) and two runs from the same state give bit-identical, decreasing losses."""
    from ctts_amd.loss import CompTransTTSLoss, ScheduledOptim
    from ctts_amd.trainer import TrainStep
    from ctts_amd.synthetic import make_batch, as_model_args
    from tests.util import switch_configs
    (pre, mc, tc), _ = switch_configs(case)
    b = make_batch([40, 33, 25, 12], 5, seed=31)
    if case == "g16_energy_frame":
        Tm = b["mels"].shape[1]
        b["e_targets"] = torch.randn(4, Tm, generator=torch.Generator().manual_seed(5)) * (torch.arange(Tm)[None, :] < b["mel_lens"][:, None])
    runs = []
    for _ in range(2):
        torch.manual_seed(1234)
        m = ctts_amd.CompTransTTS(pre, mc, tc).to(DEV)
        m.train()
        loss_fn, optim = CompTransTTSLoss(pre, mc, tc).to(DEV), ScheduledOptim(m, tc, mc, 50000, capturable=True)
        step = TrainStep(m, loss_fn, optim, as_model_args(to_device(b, DEV)), world=1, use_graph=True)
        step.capture(warmup=2)
        vals = []
        for _ in range(4):
            step()
            vals.append(float(step.loss_val))
        assert step.g_all is not None or step.graphs is not None
        runs.append(vals)
    assert runs[0] == runs[1], runs
    assert all(np.isfinite(v) for v in runs[0]) and runs[0][-1] < runs[0][0], runs[0]
