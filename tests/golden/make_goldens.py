"""Golden-vector generator - runs ONLY in the build container (needs /root/reference).

Imports the live reference (oracle/ref_import.py), loads closed-form weights
(oracle/weights.py), runs it on seeded synthetic inputs (ctts_amd.synthetic) and writes
small .npz fixtures next to this file.  Fixtures are data (inputs + expected outputs);
no reference source travels.  Re-run:  python tests/golden/make_goldens.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))

import ctts_amd  # noqa: E402
from ctts_amd.synthetic import make_batch, as_model_args  # noqa: E402
from oracle import ref_import  # noqa: E402
from oracle.weights import closed_form_state_dict, _hash_uniform  # noqa: E402

ref_import.install()
import torch.nn.functional as F  # noqa: E402

_real_dropout = F.dropout


def _np(v):
    if v is None:
        return None
    return v.detach().cpu().clone().numpy()


def pseudo(name, shape):
    return torch.from_numpy(_hash_uniform("probe." + name, int(np.prod(shape))).reshape(shape)).float()


def build(dataset="LJSpeech", block_type="transformer_fs2", learn_alignment=False, prosody="none", vp_overrides=None, tag_suffix="",
          ve_overrides=None, pitch_overrides=None, energy_overrides=None, loss_overrides=None):
    from model import CompTransTTS

    pre, mc, tc = ref_import.load_configs(dataset)
    pre["preprocessing"]["pitch"].update(pitch_overrides or {})
    pre["preprocessing"]["energy"].update(energy_overrides or {})
    tc["loss"].update(loss_overrides or {})
    mc["duration_modeling"]["learn_alignment"] = learn_alignment
    mc["prosody_modeling"]["model_type"] = prosody
    mc["block_type"] = block_type
    mc["variance_predictor"].update(vp_overrides or {})
    mc["variance_embedding"].update(ve_overrides or {})
    model = CompTransTTS(pre, mc, tc)
    sd = closed_form_state_dict(model.state_dict())
    model.load_state_dict(sd)
    import json
    tag = f"{dataset}_{block_type}" + ("_unsup" if learn_alignment else "") + ("" if prosody == "none" else "_" + prosody) + tag_suffix
    with open(os.path.join(OUT, f"state_dict_schema_{tag}.json"), "w") as f:
        json.dump({k: [list(v.shape), str(v.dtype).replace("torch.", ""), bool(k in dict(model.named_parameters()))]
                   for k, v in model.state_dict().items()}, f, indent=0)
    return model, (pre, mc, tc)


def flatten_outputs(out, prefix="out."):
    (mel, post, p_pred, e_pred, log_d, d_rounded, src_mask, mel_mask, src_lens, mel_lens, attn, pros, p_t, e_t) = out
    d = {
        "mel": _np(mel), "postnet_mel": _np(post), "e_pred": _np(e_pred) if e_pred is not None else None, "log_d": _np(log_d),
        "d_rounded": _np(d_rounded), "src_mask": _np(src_mask), "mel_mask": _np(mel_mask),
        "src_lens": _np(src_lens), "mel_lens": _np(mel_lens),
    }
    if p_pred is not None:                   # None with variance_embedding.use_pitch_embed = False (G14)
        d.update({"cwt": _np(p_pred["cwt"]), "f0_denorm": _np(p_pred["f0_denorm"]), "f0_mean": _np(p_pred["f0_mean"]),
                  "f0_std": _np(p_pred["f0_std"]), "pitch_pred": _np(p_pred["pitch_pred"])})      # pitch_pred: pitch_type frame / ph (G15)
    if p_t is not None:
        if "f0" in p_t:
            d["pt_f0"] = _np(p_t["f0"])
        d["pt_mel2ph"] = _np(p_t["mel2ph"])
    if attn is not None and attn[0] is not None:
        d["attn_soft"], d["attn_hard"], d["attn_hard_dur"], d["attn_logprob"] = [_np(a) for a in attn]
        d["e_targets_out"] = _np(e_t)
    if pros is not None:
        for n, v in zip(("up_emb", "pp_emb", "up_vec", "pp_vec", "pp_attn"), pros):
            d["pros." + n] = _np(v)
    return {prefix + k: v for k, v in d.items() if v is not None}


def batch_arrays(batch):
    d = {}
    batch = {k: v for k, v in batch.items() if v is not None}
    for k, v in batch.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                d[f"in.p_targets.{kk}"] = _np(vv)
        elif torch.is_tensor(v):
            d["in." + k] = _np(v)
        elif v is not None:
            d["in." + k] = np.asarray(v)
    return d


def run_case(model, batch, mode, name, with_grads=False, extra_kwargs=None):
    taps = {}
    hooks = [
        model.encoder.register_forward_hook(lambda m, i, o: taps.__setitem__("encoder_out", o[0])),
        model.variance_adaptor.register_forward_hook(lambda m, i, o: taps.__setitem__("va_out", o[0])),
        model.decoder.register_forward_hook(lambda m, i, o: taps.__setitem__("decoder_out", o[0])),
    ]
    if mode == "eval":
        model.eval()
    else:
        model.train()
        F.dropout = lambda x, p=0.5, training=True, inplace=False: x
    args = list(as_model_args(batch))
    if "attn_priors" in batch:
        args[10] = batch["attn_priors"]
    # the reference mutates inputs in place (p_targets dict; e_targets through a shared numpy view, modules.py:882-888): pass copies
    args = [a.clone() if torch.is_tensor(a) else a for a in args]
    if args[7] is not None:
        args[7] = {k: v.clone() for k, v in args[7].items()}
    kw = dict(extra_kwargs or {})
    sd_before = {k: v.clone() for k, v in model.state_dict().items()}
    with torch.set_grad_enabled(with_grads):
        out = model(*args, **kw)
    F.dropout = _real_dropout
    for h in hooks:
        h.remove()
    arrs = batch_arrays(batch)
    arrs.update(flatten_outputs(out))
    for k, v in taps.items():
        arrs["tap." + k] = _np(v)
    if mode == "train":
        for k, v in model.state_dict().items():
            if "running_" in k:
                arrs["bn." + k] = _np(v)
    if with_grads:
        mel, post, p_pred, e_pred, log_d = out[0], out[1], out[2], out[3], out[4]
        loss = ((post * pseudo("post", post.shape)).sum() + (mel * pseudo("mel", mel.shape)).sum()
                + (log_d * pseudo("logd", log_d.shape)).sum())
        if e_pred is not None:
            loss = loss + (e_pred * pseudo("e", e_pred.shape)).sum()
        if p_pred is not None and p_pred["cwt"] is not None:
            loss = loss + ((p_pred["cwt"] * pseudo("cwt", p_pred["cwt"].shape)).sum()
                           + (p_pred["f0_mean"] * 0.7).sum() + (p_pred["f0_std"] * -0.3).sum())
        if p_pred is not None and p_pred["pitch_pred"] is not None:
            loss = loss + (p_pred["pitch_pred"] * pseudo("ppred", p_pred["pitch_pred"].shape)).sum()
        if out[10][0] is not None:      # unsupervised: attention outputs enter the loss too
            a_soft, _, _, a_logp = out[10]
            loss = loss + (a_soft * pseudo("asoft", a_soft.shape)).sum() * 10 + (a_logp * pseudo("alogp", a_logp.shape)).sum() * 0.1
        if out[11] is not None:         # liu2021: the predictors only reach the loss through prosody_info
            loss = loss + (out[11][2] * pseudo("upvec", out[11][2].shape)).sum() + (out[11][3] * pseudo("ppvec", out[11][3].shape)).sum()
        model.zero_grad()
        loss.backward()
        arrs["grad.loss"] = _np(loss)
        for k, p in model.named_parameters():
            if p.grad is None:
                continue
            g = p.grad.detach().flatten()
            arrs["grad.head." + k] = _np(g[:64])
            arrs["grad.stat." + k] = np.array([g.double().sum().item(), g.double().pow(2).sum().sqrt().item()])
    model.load_state_dict(sd_before)  # undo BN running-stat updates
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, f"{os.path.getsize(path)/1024:.0f} KB", "mel absmax", np.abs(arrs['out.mel']).max())
    return out


def golden_loss(model_out, batch, cfgs, name, step=None):
    """G9: CompTransTTSLoss 9-tuple (model/loss.py:266-347) on a train-mode output."""
    from model import CompTransTTSLoss

    pre, mc, tc = cfgs
    L = CompTransTTSLoss(pre, mc, tc)
    L.train()
    b = [None, None] + list(as_model_args(batch))
    out = model_out
    b[9:11], output = out[-2:], out[:-2]
    step = step or tc["step"]["var_start_steps"] + 1
    losses = L(b, output, step=step)
    arrs = {}
    names = ["total", "mel", "postnet_mel", "pitch", "energy", "duration", "ctc", "bin", "prosody"]
    for n, l in zip(names, losses):
        if isinstance(l, dict):
            for k, v in l.items():
                arrs[f"loss.{n}.{k}"] = _np(v)
        else:
            arrs[f"loss.{n}"] = _np(l)
    arrs["step"] = np.asarray(step)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, {k: float(np.asarray(v).reshape(-1)[0]) for k, v in arrs.items()})


def golden_integer_vectors():
    """G7: integer kernels - LengthRegulator indices, dur_to_mel2ph, make_positions,
    f0_to_coarse, bucketize.  Includes zero durations, cropping and padded phonemes."""
    from model.modules import LengthRegulator
    from utils.tools import dur_to_mel2ph, make_positions, get_mask_from_lengths
    from utils.pitch_tools import f0_to_coarse

    g = torch.Generator().manual_seed(7)
    arrs = {}
    B, Ts, H = 5, 23, 8
    d = torch.randint(0, 9, (B, Ts), generator=g)
    d[1, 10:] = 0          # padded phonemes have zero duration
    d[2, :] = 0            # an all-zero row
    d[3, 3] = 40           # one long phoneme
    x = torch.arange(B * Ts * H, dtype=torch.float32).reshape(B, Ts, H) + 1
    lr = LengthRegulator()
    for tag, max_len in (("none", None), ("crop", 50), ("pad", 200)):
        if tag == "none":
            dd = d.clone()
        else:
            dd = d
        out, mel_len = lr(x, dd, max_len)
        # phoneme index per frame recovered from the gathered data (x rows are unique, >0)
        idx = torch.where(out[..., 0] > 0, ((out[..., 0] - 1) / H).long() % Ts, torch.full_like(out[..., 0], -1).long())
        arrs[f"lr.{tag}.idx"] = _np(idx)
        arrs[f"lr.{tag}.mel_len"] = _np(mel_len)
        arrs[f"lr.{tag}.out"] = _np(out)
    arrs["lr.dur"] = _np(d)
    arrs["lr.x"] = _np(x)
    # float durations (inference branch: clamp(round(exp(logd)-1)*ctrl, 0) is float32)
    df = torch.tensor([[0.0, 1.0, 2.0, 3.0, 0.0, 5.0], [2.0, 0.0, 0.0, 1.0, 1.0, 0.0]])
    xf = torch.arange(2 * 6 * 4, dtype=torch.float32).reshape(2, 6, 4) + 1
    outf, mlf = lr(xf, df, None)
    arrs["lrf.dur"], arrs["lrf.x"], arrs["lrf.out"], arrs["lrf.mel_len"] = _np(df), _np(xf), _np(outf), _np(mlf)
    src_lens = torch.tensor([23, 10, 23, 23, 15])
    pad = get_mask_from_lengths(src_lens, Ts)
    arrs["m2p.dur"] = _np(d)
    arrs["m2p.pad"] = _np(pad)
    arrs["m2p.out"] = _np(dur_to_mel2ph(d, pad))
    arrs["m2p.out_nopad"] = _np(dur_to_mel2ph(d, None))
    tok = torch.randint(0, 5, (4, 19), generator=g)
    arrs["pos.in"] = _np(tok)
    arrs["pos.out"] = _np(make_positions(tok, 0))
    f0 = torch.cat([torch.zeros(8), torch.rand(500, generator=g) * 1300, torch.tensor([50.0, 1100.0, 49.9, 1100.1])])
    arrs["f0c.in"] = _np(f0)
    arrs["f0c.out"] = _np(f0_to_coarse(f0.clone()))
    bins = torch.linspace(-1.431044578552246, 8.184337615966797, 255)
    ev = torch.cat([torch.randn(300, generator=g) * 3 + 2, bins[::17], torch.tensor([-5.0, 20.0])])
    arrs["bkt.bins"], arrs["bkt.in"], arrs["bkt.out"] = _np(bins), _np(ev), _np(torch.bucketize(ev, bins))
    path = os.path.join(OUT, "g7_integer.npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path)


def golden_stft():
    """G8: TacotronSTFT.mel_spectrogram (audio/stft.py:166-185) on a seeded waveform."""
    from audio.stft import TacotronSTFT

    st = TacotronSTFT(1024, 256, 1024, 80, 22050, 0, 8000)
    g = torch.Generator().manual_seed(11)
    y = (torch.rand(2, 22050, generator=g) - 0.5)
    t = torch.arange(22050) / 22050.0
    y[1] = 0.4 * torch.sin(2 * np.pi * 440 * t) + 0.1 * torch.sin(2 * np.pi * 3000 * t)
    mag, _ = st.stft_fn.transform(y)
    mel, energy = st.mel_spectrogram(y)
    arrs = {"y": _np(y), "mag": _np(mag), "mel": _np(mel), "energy": _np(energy), "mel_basis": _np(st.mel_basis),
            "window": _np(st.stft_fn.forward_basis[0, 0, :] * 0) }
    path = os.path.join(OUT, "g8_stft.npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, mel.shape, energy.shape)


def make_unsup_batch(src_lens, fpp, seed, **kw):
    """learn_alignment=True inputs: no durations, frame-level energy targets, attention prior [B,Ts,Tm]."""
    b = make_batch(src_lens, fpp, seed=seed, **kw)
    g = torch.Generator().manual_seed(seed + 1)
    B, Ts, Tm = b["texts"].shape[0], b["texts"].shape[1], b["mels"].shape[1]
    prior = torch.zeros(B, Ts, Tm)
    for i in range(B):
        P, M = int(b["src_lens"][i]), int(b["mel_lens"][i])
        # smooth diagonal-ish positive prior (stands in for the beta-binomial prior of preprocessor.py:551-560)
        t = torch.arange(M)[None, :] / M
        s_ = torch.arange(P)[:, None] / P
        prior[i, :P, :M] = torch.exp(-((t - s_) ** 2) / 0.02) + 0.05 * torch.rand(P, M, generator=g)
    b["attn_priors"] = prior
    b["d_targets"] = None
    b["e_targets"] = torch.randn(B, Tm, generator=g) * (torch.arange(Tm)[None, :] < b["mel_lens"][:, None])
    return b


def main():
    torch.manual_seed(0)
    model, cfgs = build("LJSpeech")
    small = make_batch([24, 17], 6, seed=1234)
    run_case(model, small, "eval", "g1_fs2_eval")
    out = run_case(model, small, "train", "g2_fs2_train_nodrop", with_grads=True)
    golden_loss(out, small, cfgs, "g9_loss")
    inf = {k: v for k, v in small.items()}
    inf.update(mels=None, mel_lens=None, max_mel_len=None, p_targets=None, e_targets=None, d_targets=None)
    run_case(model, inf, "eval", "g3_fs2_infer", extra_kwargs=dict(p_control=1.1, e_control=0.9, d_control=2.0))
    model_v, cfgs_v = build("VCTK")
    vb = make_batch([21, 24, 9], 5, seed=77, multi_speaker=True)
    run_case(model_v, vb, "eval", "g5_vctk_eval")
    golden_integer_vectors()
    golden_stft()
    # G6: unsupervised duration modelling (learn_alignment=True, the reference's default yaml): aligner + MAS
    model_u, cfgs_u = build("LJSpeech", "transformer_fs2", learn_alignment=True)
    ub = make_unsup_batch([24, 17], 6, seed=99)
    run_case(model_u, ub, "train", "g6_unsup_soft_step100", with_grads=True, extra_kwargs=dict(step=100))
    out_u = run_case(model_u, ub, "train", "g6_unsup_hard_step60000", with_grads=True, extra_kwargs=dict(step=60000))
    golden_loss(out_u, ub, cfgs_u, "g6_unsup_loss_step60000", step=60000)
    # G4: conformer block_type (unmasked relative attention, GLU / depthwise conv / BatchNorm module)
    model_c, cfgs_c = build("LJSpeech", "conformer")
    cb = make_batch([24, 17], 6, seed=4321)
    run_case(model_c, cb, "eval", "g4_conformer_eval")
    run_case(model_c, cb, "train", "g4_conformer_train_nodrop", with_grads=True)


def main_ffn_switches():
    """G13: variance_predictor.ffn_act / ffn_padding away from the shipped values (transformer_fs2.py:203-239, modules.py:1270-1283,
    1328-1331): swish + LEFT (causal ConstantPad1d((k-1, 0)) in every FFN and predictor; ffn_1 becomes nn.Sequential: key ffn_1.1.*)
    and relu + SAME."""
    torch.manual_seed(0)
    b = make_batch([24, 17], 6, seed=1313)
    m1, _ = build("LJSpeech", vp_overrides=dict(ffn_act="swish", ffn_padding="LEFT"), tag_suffix="_swish_left")
    run_case(m1, b, "eval", "g13_swish_left_eval")
    run_case(m1, b, "train", "g13_swish_left_train_nodrop", with_grads=True)
    m2, _ = build("LJSpeech", vp_overrides=dict(ffn_act="relu"), tag_suffix="_relu")
    run_case(m2, b, "train", "g13_relu_train_nodrop", with_grads=True)


def main_embed_switches():
    """G14: variance_embedding.use_pitch_embed / use_energy_embed = False (modules.py:735-736,754-821,1071-1099; loss.py:331-334): the
    pitch / energy branches and their parameters do not exist, predictions[2] / [3] are None, the loss keeps the initial zeros."""
    torch.manual_seed(0)
    b = make_batch([24, 17], 6, seed=1414)
    m, cfgs = build("LJSpeech", ve_overrides=dict(use_pitch_embed=False, use_energy_embed=False), tag_suffix="_noembed")
    run_case(m, b, "eval", "g14_noembed_eval")
    out = run_case(m, b, "train", "g14_noembed_train_nodrop", with_grads=True)
    golden_loss(out, b, cfgs, "g14_noembed_loss")
    m2, cfgs2 = build("LJSpeech", ve_overrides=dict(use_pitch_embed=False), tag_suffix="_nopitch")
    out2 = run_case(m2, b, "train", "g14_nopitch_train_nodrop", with_grads=True)
    golden_loss(out2, b, cfgs2, "g14_nopitch_loss")


def main_pitch_energy_switches():
    """G15: preprocessing.pitch.pitch_type "frame" / "ph" (modules.py:777-785,892-938,1083-1084; loss.py:173-178,202-219), with
    pitch_norm "standard" + use_uv False + pitch_loss l2 as a second frame case; G16: preprocessing.energy.feature "frame_level"
    (modules.py:1092-1094; loss.py:238-242; stats key energy_sup_frame)."""
    torch.manual_seed(0)
    b = make_batch([24, 17], 6, seed=1515)
    inf = {k: v for k, v in b.items()}
    inf.update(mels=None, mel_lens=None, max_mel_len=None, p_targets=None, e_targets=None, d_targets=None)
    m, cfgs = build("LJSpeech", pitch_overrides=dict(pitch_type="frame"), tag_suffix="_pitchframe")
    run_case(m, b, "eval", "g15_pitch_frame_eval")
    out = run_case(m, b, "train", "g15_pitch_frame_train_nodrop", with_grads=True)
    golden_loss(out, b, cfgs, "g15_pitch_frame_loss")
    run_case(m, inf, "eval", "g15_pitch_frame_infer", extra_kwargs=dict(p_control=1.1, e_control=0.9, d_control=2.0))
    m2, cfgs2 = build("LJSpeech", pitch_overrides=dict(pitch_type="frame", pitch_norm="standard", use_uv=False, f0_mean=7.4, f0_std=0.35),
                      loss_overrides=dict(pitch_loss="l2"), tag_suffix="_pitchframe_nouv")
    out2 = run_case(m2, b, "train", "g15_pitch_frame_std_nouv_train_nodrop", with_grads=True)
    golden_loss(out2, b, cfgs2, "g15_pitch_frame_std_nouv_loss")
    m3, cfgs3 = build("LJSpeech", pitch_overrides=dict(pitch_type="ph"), tag_suffix="_pitchph")
    run_case(m3, b, "eval", "g15_pitch_ph_eval")
    out3 = run_case(m3, b, "train", "g15_pitch_ph_train_nodrop", with_grads=True)
    golden_loss(out3, b, cfgs3, "g15_pitch_ph_loss")
    run_case(m3, inf, "eval", "g15_pitch_ph_infer", extra_kwargs=dict(p_control=1.1, e_control=0.9, d_control=2.0))
    # G16: frame-level energy - the targets are per frame
    g = torch.Generator().manual_seed(1616)
    be = {k: v for k, v in b.items()}
    Tm = b["mels"].shape[1]
    be["e_targets"] = torch.randn(len(b["src_lens"]), Tm, generator=g) * (torch.arange(Tm)[None, :] < b["mel_lens"][:, None])
    m4, cfgs4 = build("LJSpeech", energy_overrides=dict(feature="frame_level"), tag_suffix="_energyframe")
    run_case(m4, be, "eval", "g16_energy_frame_eval")
    out4 = run_case(m4, be, "train", "g16_energy_frame_train_nodrop", with_grads=True)
    golden_loss(out4, be, cfgs4, "g16_energy_frame_loss")


from tests.util import synthetic_samples  # noqa: E402  (shared with tests/test_data_cpu.py)


def golden_collate():
    """G11: Dataset.collate_fn / reprocess (dataset.py:166-248) + to_device (utils/tools.py:69-134) on in-memory samples."""
    from dataset import Dataset
    from utils.tools import to_device

    arrs = {}
    for tag, la in (("sup", False), ("unsup", True)):
        ds = Dataset.__new__(Dataset)
        ds.pitch_type, ds.learn_alignment, ds.load_spker_embed, ds.batch_size, ds.sort, ds.drop_last = "cwt", la, False, 4, True, False
        samples = synthetic_samples(10, 5 + int(la), la)
        batches = ds.collate_fn(samples)
        arrs[f"{tag}.n_batches"] = np.asarray(len(batches))
        for bi, b in enumerate(batches):
            dev = to_device(b, "cpu")
            arrs[f"{tag}.b{bi}.ids"] = np.array(b[0])
            flat = {"speakers": dev[2], "texts": dev[3], "src_lens": dev[4], "max_src_len": torch.tensor(int(dev[5])), "mels": dev[6],
                    "mel_lens": dev[7], "max_mel_len": torch.tensor(int(dev[8])), "energies": dev[10], "durations": dev[11],
                    "attn_priors": dev[12]}
            flat.update({"pitch." + k: v for k, v in dev[9].items()})
            for k, v in flat.items():
                if v is not None:
                    arrs[f"{tag}.b{bi}.{k}"] = v.numpy()
    path = os.path.join(OUT, "g11_collate.npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, len(arrs), "arrays")


def main_vctk_unsup():
    """G12: the reference's DEFAULT VCTK yaml - multi_speaker + learn_alignment=True (aligner with speaker projections)."""
    torch.manual_seed(0)
    model, cfgs = build("VCTK", "transformer_fs2", learn_alignment=True)
    vb = make_unsup_batch([21, 24, 9], 5, seed=78, multi_speaker=True)
    run_case(model, vb, "train", "g12_vctk_unsup_step60000", with_grads=True, extra_kwargs=dict(step=60000))


def main_liu2021():
    """G10: prosody_modeling.model_type = liu2021, supervised and with learn_alignment=True (config C5)."""
    torch.manual_seed(0)
    model, cfgs = build("LJSpeech", "transformer_fs2", prosody="liu2021")
    small = make_batch([24, 17], 6, seed=1234)
    run_case(model, small, "eval", "g10_liu2021_eval")
    out = run_case(model, small, "train", "g10_liu2021_train_nodrop", with_grads=True)
    golden_loss(out, small, cfgs, "g10_liu2021_loss", step=100001)
    model_u, cfgs_u = build("LJSpeech", "transformer_fs2", learn_alignment=True, prosody="liu2021")
    ub = make_unsup_batch([24, 17], 6, seed=99)
    out_u = run_case(model_u, ub, "train", "g10_liu2021_unsup_step60000", with_grads=True, extra_kwargs=dict(step=60000))
    golden_loss(out_u, ub, cfgs_u, "g10_liu2021_unsup_loss_step100001", step=100001)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "liu2021":
        main_liu2021()
    elif len(sys.argv) > 1 and sys.argv[1] == "collate":
        golden_collate()
    elif len(sys.argv) > 1 and sys.argv[1] == "vctk_unsup":
        main_vctk_unsup()
    elif len(sys.argv) > 1 and sys.argv[1] == "ffn_switches":
        main_ffn_switches()
    elif len(sys.argv) > 1 and sys.argv[1] == "embed_switches":
        main_embed_switches()
    elif len(sys.argv) > 1 and sys.argv[1] == "pitch_energy_switches":
        main_pitch_energy_switches()
    else:
        main()
        main_liu2021()
        golden_collate()
        main_vctk_unsup()
        main_ffn_switches()
        main_embed_switches()
        main_pitch_energy_switches()
