"""CPU: the oracle restatement (oracle/restate.py) reproduces the golden vectors captured
from the live reference (tests/golden/make_goldens.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import restate as R
from tests.util import load_golden, closed_form_sd, batch_from_golden
from ctts_amd.configs import get_configs

TOL = 2e-5


def _run(gname, dataset="LJSpeech", training=False, **kw):
    g = load_golden(gname)
    sd = closed_form_sd(dataset)
    pre, mc, tc = get_configs(dataset)
    b = batch_from_golden(g)
    taps, stats = {}, {}
    out = R.comp_trans_tts_forward(sd, mc, pre, b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"],
                                   b["mel_lens"], b["max_mel_len"], b["p_targets"], b["e_targets"], b["d_targets"],
                                   None, b["spker_embeds"], training=training, taps=taps, new_stats=stats, **kw)
    return g, out, taps, stats, sd


def _close(a, b, tol=TOL, name=""):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    err = np.abs(a.astype(np.float64) - b.astype(np.float64)).max()
    assert err <= tol, f"{name}: max-abs {err}"


def _check_outputs(g, out, taps):
    mel, post, p_pred, e_pred, log_d, d_rounded, src_pad, mel_pad, src_lens, mel_lens = out[:10]
    _close(taps["encoder_out"], g["tap.encoder_out"], name="encoder_out")
    _close(taps["va_out"], g["tap.va_out"], name="va_out")
    _close(taps["decoder_out"], g["tap.decoder_out"], name="decoder_out")
    _close(mel, g["out.mel"], name="mel")
    _close(post, g["out.postnet_mel"], 5e-5, name="postnet_mel")
    _close(log_d, g["out.log_d"], name="log_d")
    _close(e_pred, g["out.e_pred"], name="e_pred")
    _close(p_pred["cwt"], g["out.cwt"], name="cwt")
    _close(p_pred["f0_mean"], g["out.f0_mean"], name="f0_mean")
    _close(p_pred["f0_denorm"], g["out.f0_denorm"], 1e-2, name="f0_denorm")  # values ~O(200) Hz
    assert np.array_equal(src_pad.numpy(), g["out.src_mask"])
    assert np.array_equal(mel_pad.numpy(), g["out.mel_mask"])
    assert np.array_equal(mel_lens.numpy(), g["out.mel_lens"])
    assert np.array_equal(d_rounded.numpy(), g["out.d_rounded"])


def test_g1_fs2_eval():
    g, out, taps, _, _ = _run("g1_fs2_eval")
    _check_outputs(g, out, taps)
    _close(out[12]["f0"], g["out.pt_f0"], 1e-4, name="p_targets.f0")


def test_g2_fs2_train_bn_batch_stats_and_grads():
    g = load_golden("g2_fs2_train_nodrop")
    sd = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in closed_form_sd().items()}
    pre, mc, tc = get_configs()
    b = batch_from_golden(g)
    taps, stats = {}, {}
    out = R.comp_trans_tts_forward(sd, mc, pre, b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"],
                                   b["mel_lens"], b["max_mel_len"], b["p_targets"], b["e_targets"], b["d_targets"],
                                   training=True, taps=taps, new_stats=stats)
    _check_outputs(g, out, taps)
    for k, v in stats.items():
        _close(v, g["bn." + k], name=k)
    from oracle.weights import _hash_uniform

    def pseudo(name, shape):
        return torch.from_numpy(_hash_uniform("probe." + name, int(np.prod(shape))).reshape(shape)).float()
    mel, post, p_pred, e_pred, log_d = out[:5]
    loss = ((post * pseudo("post", post.shape)).sum() + (mel * pseudo("mel", mel.shape)).sum()
            + (log_d * pseudo("logd", log_d.shape)).sum() + (e_pred * pseudo("e", e_pred.shape)).sum()
            + (p_pred["cwt"] * pseudo("cwt", p_pred["cwt"].shape)).sum()
            + (p_pred["f0_mean"] * 0.7).sum() + (p_pred["f0_std"] * -0.3).sum())
    _close(loss, g["grad.loss"], 2e-3, name="probe loss")
    loss.backward()
    n = 0
    for k, v in sd.items():
        if "grad.stat." + k not in g:
            continue
        gs = g["grad.stat." + k]
        gr = v.grad.flatten() if v.grad is not None else torch.zeros(v.numel())
        scale = max(1.0, float(gs[1]))
        _close(gr[:64], g["grad.head." + k], 2e-4 * scale, name="grad " + k)
        assert abs(float(gr.double().pow(2).sum().sqrt()) - gs[1]) <= 2e-4 * scale, k
        n += 1
    assert n > 150


def test_g3_inference_branch():
    g, out, taps, _, _ = _run("g3_fs2_infer", p_control=1.1, e_control=0.9, d_control=2.0)
    _check_outputs(g, out, taps)


def test_g5_vctk_multispeaker():
    g, out, taps, _, _ = _run("g5_vctk_eval", dataset="VCTK")
    _check_outputs(g, out, taps)


def test_g7_integer_vectors_bit_exact():
    g = load_golden("g7_integer")
    d, x = torch.from_numpy(g["lr.dur"]), torch.from_numpy(g["lr.x"])
    for tag, max_len in (("none", None), ("crop", 50), ("pad", 200)):
        idx, mel_len = R.length_regulate_indices(d, max_len)
        out, _ = R.length_regulate(x, d, max_len)
        assert np.array_equal(idx.numpy(), g[f"lr.{tag}.idx"]), tag
        assert np.array_equal(mel_len.numpy(), g[f"lr.{tag}.mel_len"]), tag
        assert np.array_equal(out.numpy(), g[f"lr.{tag}.out"]), tag
    outf, mlf = R.length_regulate(torch.from_numpy(g["lrf.x"]), torch.from_numpy(g["lrf.dur"]), None)
    assert np.array_equal(outf.numpy(), g["lrf.out"]) and np.array_equal(mlf.numpy(), g["lrf.mel_len"])
    pad = torch.from_numpy(g["m2p.pad"])
    assert np.array_equal(R.dur_to_mel2ph(d, pad).numpy(), g["m2p.out"])
    assert np.array_equal(R.dur_to_mel2ph(d, None).numpy(), g["m2p.out_nopad"])
    tok = torch.from_numpy(g["pos.in"])
    assert np.array_equal(R.positions_from_nonpad(tok.ne(0)).numpy(), g["pos.out"])
    assert np.array_equal(R.f0_to_coarse(torch.from_numpy(g["f0c.in"])).numpy(), g["f0c.out"])
    assert np.array_equal(torch.bucketize(torch.from_numpy(g["bkt.in"]), torch.from_numpy(g["bkt.bins"])).numpy(),
                          g["bkt.out"])


def _run_conformer(gname, training):
    g = load_golden(gname)
    sd = closed_form_sd("LJSpeech", "conformer")
    pre, mc, tc = get_configs()
    mc["block_type"] = "conformer"
    b = batch_from_golden(g)
    taps, stats = {}, {}
    out = R.comp_trans_tts_forward_conformer(sd, mc, pre, b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"],
                                             b["mel_lens"], b["max_mel_len"], b["p_targets"], b["e_targets"], b["d_targets"],
                                             training=training, taps=taps, new_stats=stats)
    return g, out, taps, stats


def test_g4_conformer_eval():
    g, out, taps, _ = _run_conformer("g4_conformer_eval", False)
    _check_outputs(g, out, taps)


def test_g4_conformer_train_batch_stats():
    g, out, taps, stats = _run_conformer("g4_conformer_train_nodrop", True)
    _check_outputs(g, out, taps)
    assert len(stats) == 10 * 2 + 5 * 2
    for k, v in stats.items():
        _close(v, g["bn." + k], 5e-5, name=k)


@pytest.mark.parametrize("gname,step", [("g6_unsup_soft_step100", 100), ("g6_unsup_hard_step60000", 60000)])
def test_g6_unsupervised_alignment(gname, step):
    g = load_golden(gname)
    sd = closed_form_sd(unsup=True)
    pre, mc, tc = get_configs()
    mc["duration_modeling"]["learn_alignment"] = True
    b = batch_from_golden(g)
    taps, stats = {}, {}
    out = R.comp_trans_tts_forward(sd, mc, pre, b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"],
                                   b["mel_lens"], b["max_mel_len"], b["p_targets"], b["e_targets"], None, b["attn_priors"], None,
                                   step=step, training=True, taps=taps, new_stats=stats)
    a_soft, a_hard, a_dur, a_logp = out[10]
    _close(a_soft, g["out.attn_soft"], 1e-6, "attn_soft")
    _close(a_logp, g["out.attn_logprob"], 1e-4, "attn_logprob")
    assert np.array_equal(a_hard.numpy(), g["out.attn_hard"])            # MAS path: exact
    assert np.array_equal(a_dur.numpy(), g["out.attn_hard_dur"])
    _close(out[13], g["out.e_targets_out"], 1e-5, "phoneme-level energy targets")
    assert np.array_equal(out[12]["mel2ph"].numpy(), g["out.pt_mel2ph"])
    _close(out[0], g["out.mel"], name="mel")
    _close(out[1], g["out.postnet_mel"], 5e-5, name="postnet_mel")
    _close(out[4], g["out.log_d"], name="log_d")
    _close(out[3], g["out.e_pred"], name="e_pred")


def _pseudo(name, shape):
    from oracle.weights import _hash_uniform
    return torch.from_numpy(_hash_uniform("probe." + name, int(np.prod(shape))).reshape(shape)).float()


PROS_NAMES = ("up_emb", "pp_emb", "up_vec", "pp_vec", "pp_attn")


@pytest.mark.parametrize("gname,unsup,training", [("g10_liu2021_eval", False, False), ("g10_liu2021_train_nodrop", False, True),
                                                  ("g10_liu2021_unsup_step60000", True, True)])
def test_g10_liu2021_prosody(gname, unsup, training):
    """reference encoders (CoordConv2d stack + BN2d + GRU), STL, phoneme-level attention, bi-GRU predictors."""
    g = load_golden(gname)
    sd = closed_form_sd(unsup=unsup, prosody="liu2021")
    if training:
        sd = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    pre, mc, tc = get_configs()
    mc["duration_modeling"]["learn_alignment"] = unsup
    mc["prosody_modeling"]["model_type"] = "liu2021"
    b = batch_from_golden(g)
    taps, stats = {}, {}
    out = R.comp_trans_tts_forward(sd, mc, pre, b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"],
                                   b["mel_lens"], b["max_mel_len"], b["p_targets"], b["e_targets"], b["d_targets"],
                                   b["attn_priors"] if unsup else None, None, step=60000 if unsup else None,
                                   training=training, taps=taps, new_stats=stats)
    for n, v in zip(PROS_NAMES, out[11]):
        if v is None:
            assert "out.pros." + n not in g
        else:
            _close(v, g["out.pros." + n], 2e-5, "prosody " + n)
    _close(out[0], g["out.mel"], 5e-5, name="mel")
    _close(out[1], g["out.postnet_mel"], 1e-4, name="postnet_mel")
    _close(out[4], g["out.log_d"], 5e-5, name="log_d")
    _close(out[3], g["out.e_pred"], 5e-5, name="e_pred")
    if not training:
        return
    for k, v in stats.items():
        _close(v, g["bn." + k], 5e-5, name=k)
    assert sum("prosody_encoder" in k for k in stats) == 24          # 2 encoders x 6 BN2d x (mean, var)
    mel, post, p_pred, e_pred, log_d = out[:5]
    loss = ((post * _pseudo("post", post.shape)).sum() + (mel * _pseudo("mel", mel.shape)).sum()
            + (log_d * _pseudo("logd", log_d.shape)).sum() + (e_pred * _pseudo("e", e_pred.shape)).sum()
            + (p_pred["cwt"] * _pseudo("cwt", p_pred["cwt"].shape)).sum()
            + (p_pred["f0_mean"] * 0.7).sum() + (p_pred["f0_std"] * -0.3).sum())
    if unsup:
        a_soft, _, _, a_logp = out[10]
        loss = loss + (a_soft * _pseudo("asoft", a_soft.shape)).sum() * 10 + (a_logp * _pseudo("alogp", a_logp.shape)).sum() * 0.1
    loss = loss + (out[11][2] * _pseudo("upvec", out[11][2].shape)).sum() + (out[11][3] * _pseudo("ppvec", out[11][3].shape)).sum()
    _close(loss, g["grad.loss"], 5e-3, name="probe loss")
    loss.backward()
    n = 0
    for k, v in sd.items():
        if "grad.stat." + k not in g or "prosody" not in k:
            continue
        gs = g["grad.stat." + k]
        gr = v.grad.flatten() if v.grad is not None else torch.zeros(v.numel())
        scale = max(1.0, float(gs[1]))
        # one BN2d output of the unsup fixture sits within 5e-6 of the ReLU kink: its derivative (0 or 1) is decided by
        # rounding noise and moves the conv-stack gradients below it by ~5e-3 relative (traced against the live reference)
        tol = (1e-2 if (unsup and "phoneme_prosody_encoder.encoder." in k) else 3e-4) * scale
        _close(gr[:64], g["grad.head." + k], tol, name="grad " + k)
        assert abs(float(gr.double().pow(2).sum().sqrt()) - gs[1]) <= tol, k
        n += 1
    assert n > 80, n


def test_g12_vctk_multispeaker_unsupervised():
    """the reference's default VCTK yaml: multi_speaker + learn_alignment=True - aligner with speaker projections (modules.py:1188-1194)."""
    g = load_golden("g12_vctk_unsup_step60000")
    sd = closed_form_sd("VCTK", unsup=True)
    pre, mc, tc = get_configs("VCTK")
    mc["duration_modeling"]["learn_alignment"] = True
    b = batch_from_golden(g)
    out = R.comp_trans_tts_forward(sd, mc, pre, b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"], b["mel_lens"],
                                   b["max_mel_len"], b["p_targets"], b["e_targets"], None, b["attn_priors"], b["spker_embeds"],
                                   step=60000, training=True)
    a_soft, a_hard, a_dur, a_logp = out[10]
    _close(a_soft, g["out.attn_soft"], 1e-6, "attn_soft")
    _close(a_logp, g["out.attn_logprob"], 1e-4, "attn_logprob")
    assert np.array_equal(a_hard.numpy(), g["out.attn_hard"]) and np.array_equal(a_dur.numpy(), g["out.attn_hard_dur"])
    _close(out[0], g["out.mel"], name="mel")
    _close(out[1], g["out.postnet_mel"], 5e-5, name="postnet_mel")


@pytest.mark.parametrize("gname,suffix,vp,training", [("g13_swish_left_eval", "_swish_left", dict(ffn_act="swish", ffn_padding="LEFT"), False),
                                                      ("g13_swish_left_train_nodrop", "_swish_left", dict(ffn_act="swish", ffn_padding="LEFT"), True),
                                                      ("g13_relu_train_nodrop", "_relu", dict(ffn_act="relu"), True)])
def test_g13_ffn_act_and_padding_switches(gname, suffix, vp, training):
    """variance_predictor.ffn_act / ffn_padding away from the shipped values (transformer_fs2.py:203-239; the predictors' ConstantPad1d
    modules.py:1270-1283,1328-1331): swish + LEFT (causal; ffn_1 is nn.Sequential there: state-dict key ffn_1.1.*) and relu + SAME."""
    g = load_golden(gname)
    sd = closed_form_sd(suffix=suffix)
    pre, mc, tc = get_configs()
    mc["variance_predictor"].update(vp)
    b = batch_from_golden(g)
    taps, stats = {}, {}
    out = R.comp_trans_tts_forward(sd, mc, pre, b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"],
                                   b["mel_lens"], b["max_mel_len"], b["p_targets"], b["e_targets"], b["d_targets"],
                                   None, b["spker_embeds"], training=training, taps=taps, new_stats=stats)
    _check_outputs(g, out, taps)
    R._SW.update(ffn_act="gelu", ffn_padding="SAME")


@pytest.mark.parametrize("gname,suffix,ve,training", [("g14_noembed_eval", "_noembed", dict(use_pitch_embed=False, use_energy_embed=False), False),
                                                      ("g14_noembed_train_nodrop", "_noembed", dict(use_pitch_embed=False, use_energy_embed=False), True),
                                                      ("g14_nopitch_train_nodrop", "_nopitch", dict(use_pitch_embed=False), True)])
def test_g14_pitch_and_energy_embedding_switches(gname, suffix, ve, training):
    """variance_embedding.use_pitch_embed / use_energy_embed = False (modules.py:735-736,754-821,1071-1099): the branch, its parameters
    and its prediction (None) disappear."""
    g = load_golden(gname)
    sd = closed_form_sd(suffix=suffix)
    pre, mc, tc = get_configs()
    mc["variance_embedding"].update(ve)
    b = batch_from_golden(g)
    taps, stats = {}, {}
    out = R.comp_trans_tts_forward(sd, mc, pre, b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"],
                                   b["mel_lens"], b["max_mel_len"], b["p_targets"], b["e_targets"], b["d_targets"],
                                   None, b["spker_embeds"], training=training, taps=taps, new_stats=stats)
    mel, post, p_pred, e_pred, log_d = out[:5]
    assert p_pred is None and (e_pred is None) == (not ve.get("use_energy_embed", True))
    assert "out.cwt" not in g and ("out.e_pred" in g) == (e_pred is not None)
    for name, a, key in (("encoder_out", taps["encoder_out"], "tap.encoder_out"), ("va_out", taps["va_out"], "tap.va_out"),
                         ("decoder_out", taps["decoder_out"], "tap.decoder_out"), ("mel", mel, "out.mel"), ("log_d", log_d, "out.log_d")):
        _close(a, g[key], name=name)
    _close(post, g["out.postnet_mel"], 5e-5, name="postnet_mel")
    if e_pred is not None:
        _close(e_pred, g["out.e_pred"], name="e_pred")


@pytest.mark.parametrize("case,mode", [("g15_pitch_frame", "eval"), ("g15_pitch_frame", "train_nodrop"), ("g15_pitch_frame", "infer"),
                                       ("g15_pitch_frame_std_nouv", "train_nodrop"), ("g15_pitch_ph", "eval"),
                                       ("g15_pitch_ph", "train_nodrop"), ("g15_pitch_ph", "infer"),
                                       ("g16_energy_frame", "eval"), ("g16_energy_frame", "train_nodrop")])
def test_g15_g16_pitch_type_and_energy_level_switches(case, mode):
    """preprocessing.pitch.pitch_type "frame" / "ph" (+ pitch_norm standard, use_uv False) and preprocessing.energy.feature
    "frame_level" (modules.py:777-785,892-938,1083-1094): forward values and - for the train cases - the loss 9-tuple
    (loss.py:173-178,202-219,238-242) against the live reference's goldens."""
    from tests.util import switch_configs
    from oracle.loss_restate import RefLoss
    g = load_golden(f"{case}_{mode}")
    (pre, mc, tc), sd = switch_configs(case)
    b = batch_from_golden(g)
    taps, stats = {}, {}
    kw = dict(p_control=1.1, e_control=0.9, d_control=2.0) if mode == "infer" else {}
    args = [b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"], b["mel_lens"], b["max_mel_len"], b["p_targets"],
            b["e_targets"], b["d_targets"], None, b["spker_embeds"]]
    with torch.no_grad():
        out = R.comp_trans_tts_forward(sd, mc, pre, *args, training=(mode == "train_nodrop"), taps=taps, new_stats=stats, **kw)
    mel, post, p_pred, e_pred, log_d = out[:5]
    for name, a, key in (("encoder_out", taps["encoder_out"], "tap.encoder_out"), ("va_out", taps["va_out"], "tap.va_out"),
                         ("decoder_out", taps["decoder_out"], "tap.decoder_out"), ("mel", mel, "out.mel"), ("log_d", log_d, "out.log_d"),
                         ("e_pred", e_pred, "out.e_pred"), ("f0_denorm", p_pred["f0_denorm"], "out.f0_denorm")):
        _close(a, g[key], name=name)
    _close(post, g["out.postnet_mel"], 5e-5, name="postnet_mel")
    if case.startswith("g15"):
        assert "out.cwt" not in g and p_pred["cwt"] is None
        _close(p_pred["pitch_pred"], g["out.pitch_pred"], name="pitch_pred")
        assert p_pred["pitch_pred"].shape[-1] == (1 if "ph" in case else 2)
    if mode != "infer" and "out.pt_f0" in g:
        _close(out[-2]["f0"], g["out.pt_f0"], name="pitch_target f0 (padding-zeroed / phoneme level)")
    if mode == "train_nodrop":
        gl = load_golden(f"{case}_loss")
        inputs = [None, None] + list(args)
        inputs[9:11] = out[-2:]
        losses = RefLoss(pre, mc, tc)(inputs, out[:-2], int(gl["step"]))
        flat = {"total": losses[0], "mel": losses[1], "postnet_mel": losses[2], "energy": losses[4]}
        flat.update({"pitch." + k: v for k, v in losses[3].items()})
        flat.update({"duration." + k: v for k, v in losses[5].items()})
        assert {k for k in gl if k.startswith("loss.pitch.")} == {"loss.pitch." + k for k in losses[3]}
        for k, v in flat.items():
            ref = float(np.asarray(gl["loss." + k]).reshape(-1)[0])
            assert abs(float(v.reshape(-1)[0]) - ref) <= 2e-4 * max(1.0, abs(ref)), (k, float(v.reshape(-1)[0]), ref)
