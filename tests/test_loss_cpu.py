"""CPU: the oracle's loss restatement (oracle/loss_restate.py) reproduces the reference's 9-tuple (goldens G9 / G6-loss / G10-loss,
captured from model/loss.py); the product loss is device-only and is checked against the same goldens in tests/test_model_gpu.py."""
import numpy as np
import torch

from ctts_amd.configs import get_configs
from ctts_amd.loss import ScheduledOptim
from oracle.loss_restate import RefLoss as CompTransTTSLoss
from oracle import restate as R
from tests.util import load_golden, closed_form_sd, batch_from_golden


def test_loss_matches_reference_golden():
    g2, g9 = load_golden("g2_fs2_train_nodrop"), load_golden("g9_loss")
    pre, mc, tc = get_configs()
    b = batch_from_golden(g2)
    args = [b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"], b["mel_lens"], b["max_mel_len"],
            b["p_targets"], b["e_targets"], b["d_targets"], None, None]
    out = R.comp_trans_tts_forward(closed_form_sd(), mc, pre, *args, training=True)
    inputs = [None, None] + list(args)
    inputs[9:11] = out[-2:]
    L = CompTransTTSLoss(pre, mc, tc)
    total, mel, post, pitch, energy, dur, ctc, binl, pros = L(inputs, out[:-2], int(g9["step"]))
    exp = {"total": total, "mel": mel, "postnet_mel": post, "energy": energy, "pitch.C": pitch["C"], "pitch.uv": pitch["uv"],
           "pitch.f0_mean": pitch["f0_mean"], "pitch.f0_std": pitch["f0_std"], "duration.pdur": dur["pdur"],
           "duration.wdur": dur["wdur"], "duration.sdur": dur["sdur"]}
    for k, v in exp.items():
        ref = float(np.asarray(g9["loss." + k]).reshape(-1)[0])
        assert abs(float(v.reshape(-1)[0]) - ref) <= 2e-4 * max(1.0, abs(ref)), (k, float(v.reshape(-1)[0]), ref)
    # before var_start_steps only the two mel terms count (loss.py:329-335)
    t0 = L(inputs, out[:-2], 10)[0]
    assert abs(float(t0) - float(mel + post)) < 1e-6


def test_noam_schedule_values():
    pre, mc, tc = get_configs()
    lin = torch.nn.Linear(4, 4)
    so = ScheduledOptim(lin, tc, mc, 0)
    lrs = [so.update_learning_rate() for _ in range(3)]
    init = 256 ** -0.5
    assert abs(lrs[0] - init * 4000 ** -1.5 * 1) < 1e-12 and abs(lrs[2] - init * 4000 ** -1.5 * 3) < 1e-12
    so.current_step = 299999
    a = so.update_learning_rate()
    b = so.update_learning_rate()
    assert abs(a - init * 300000 ** -0.5) < 1e-12 and abs(b - init * 300001 ** -0.5 * 0.3) < 1e-12


def test_unsupervised_loss_matches_reference_golden():
    """ForwardSumLoss (batched CTC restatement) + BinLoss + the rest at step 60000 (bin-loss weight 1)."""
    g, g9 = load_golden("g6_unsup_hard_step60000"), load_golden("g6_unsup_loss_step60000")
    pre, mc, tc = get_configs()
    mc["duration_modeling"]["learn_alignment"] = True
    b = batch_from_golden(g)
    args = [b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"], b["mel_lens"], b["max_mel_len"],
            b["p_targets"], b["e_targets"], None, b["attn_priors"], None]
    out = R.comp_trans_tts_forward(closed_form_sd(unsup=True), mc, pre, *args, step=60000, training=True)
    inputs = [None, None] + list(args)
    inputs[9:11] = out[-2:]
    L = CompTransTTSLoss(pre, mc, tc)
    total, mel, post, pitch, energy, dur, ctc, binl, pros = L(inputs, out[:-2], 60000)
    exp = {"total": total, "mel": mel, "postnet_mel": post, "energy": energy, "pitch.C": pitch["C"], "pitch.uv": pitch["uv"],
           "duration.pdur": dur["pdur"], "duration.wdur": dur["wdur"], "duration.sdur": dur["sdur"], "ctc": ctc, "bin": binl}
    for k, v in exp.items():
        ref = float(np.asarray(g9["loss." + k]).reshape(-1)[0])
        assert abs(float(v.reshape(-1)[0]) - ref) <= 3e-4 * max(1.0, abs(ref)), (k, float(v.reshape(-1)[0]), ref)


def test_liu2021_prosody_loss_matches_reference_golden():
    """prosody L1 terms (loss.py:319-324, incl. the pad-position selection quirk) after prosody_loss_enable_steps,
    supervised and with learn_alignment=True (config C5)."""
    for gname, lname, unsup in (("g10_liu2021_train_nodrop", "g10_liu2021_loss", False),
                                ("g10_liu2021_unsup_step60000", "g10_liu2021_unsup_loss_step100001", True)):
        g, g9 = load_golden(gname), load_golden(lname)
        pre, mc, tc = get_configs()
        mc["duration_modeling"]["learn_alignment"] = unsup
        mc["prosody_modeling"]["model_type"] = "liu2021"
        b = batch_from_golden(g)
        args = [b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"], b["mel_lens"], b["max_mel_len"],
                b["p_targets"], b["e_targets"], b["d_targets"], b["attn_priors"] if unsup else None, None]
        out = R.comp_trans_tts_forward(closed_form_sd(unsup=unsup, prosody="liu2021"), mc, pre, *args,
                                       step=60000 if unsup else None, training=True)
        inputs = [None, None] + list(args)
        inputs[9:11] = out[-2:]
        L = CompTransTTSLoss(pre, mc, tc)
        step = int(g9["step"])
        total, mel, post, pitch, energy, dur, ctc, binl, pros = L(inputs, out[:-2], step)
        exp = {"total": total, "mel": mel, "postnet_mel": post, "prosody": pros, "ctc": ctc, "bin": binl, "energy": energy}
        for k, v in exp.items():
            ref = float(np.asarray(g9["loss." + k]).reshape(-1)[0])
            assert abs(float(v.reshape(-1)[0]) - ref) <= 3e-4 * max(1.0, abs(ref)), (gname, k, float(v.reshape(-1)[0]), ref)
        assert float(L(inputs, out[:-2], 1000)[8]) == 0.0          # before prosody_loss_enable_steps
