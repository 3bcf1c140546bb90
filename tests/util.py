"""Shared helpers for tests: golden loading, closed-form state dicts, batch rebuild."""
import json
import os

import numpy as np
import torch

import ctts_amd  # noqa: F401  (root shim)
from ctts_amd.configs import get_configs
from oracle.weights import fill_tensor, SKIP

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def schema(dataset="LJSpeech", block="transformer_fs2", unsup=False, prosody="none", suffix=""):
    tag = f"{dataset}_{block}{'_unsup' if unsup else ''}{'' if prosody == 'none' else '_' + prosody}{suffix}"
    with open(os.path.join(GOLDEN, f"state_dict_schema_{tag}.json")) as f:
        return json.load(f)


# G15 / G16: preprocessing.pitch / preprocessing.energy / train.loss overrides of the goldens make_goldens.main_pitch_energy_switches wrote
SWITCH_CASES = {
    "g15_pitch_frame": dict(suffix="_pitchframe", pitch=dict(pitch_type="frame")),
    "g15_pitch_frame_std_nouv": dict(suffix="_pitchframe_nouv", loss=dict(pitch_loss="l2"),
                                     pitch=dict(pitch_type="frame", pitch_norm="standard", use_uv=False, f0_mean=7.4, f0_std=0.35)),
    "g15_pitch_ph": dict(suffix="_pitchph", pitch=dict(pitch_type="ph")),
    "g16_energy_frame": dict(suffix="_energyframe", energy=dict(feature="frame_level"), energy_key="energy_sup_frame"),
}


def switch_configs(case):
    """(preprocess, model, train) configs of a SWITCH_CASES entry + its closed-form state dict"""
    c = SWITCH_CASES[case]
    pre, mc, tc = get_configs()
    pre["preprocessing"]["pitch"].update(c.get("pitch", {}))
    pre["preprocessing"]["energy"].update(c.get("energy", {}))
    tc["loss"].update(c.get("loss", {}))
    return (pre, mc, tc), closed_form_sd(suffix=c["suffix"], energy_key=c.get("energy_key"))


def closed_form_sd(dataset="LJSpeech", block="transformer_fs2", unsup=False, prosody="none", suffix="", energy_key=None):
    """Closed-form weights for every schema key (energy_bins from stats.json like modules.py:795-818).  `suffix`: schema of a
    configuration variant (G13: "_swish_left", "_relu")."""
    pre, mc, tc = get_configs(dataset)
    sd = {}
    for k, (shape, dtype, is_param) in schema(dataset, block, unsup, prosody, suffix).items():
        if k.endswith("energy_bins"):
            with open(os.path.join(pre["path"]["preprocessed_path"], "stats.json")) as f:
                emin, emax = json.load(f)[energy_key or ("energy_unsup_frame" if unsup else "energy_sup_phone")][:2]
            sd[k] = torch.linspace(emin, emax, shape[0])
        elif "position_enc" in k or "positional_encoding" in k:
            from oracle.restate import interleaved_sinusoid_table
            sd[k] = interleaved_sinusoid_table(shape[1], shape[2]).unsqueeze(0)
        elif any(s in k for s in SKIP):
            sd[k] = torch.zeros(shape)
        else:
            sd[k] = fill_tensor(k, tuple(shape))
    return sd


def batch_from_golden(g):
    """Rebuild the positional model args from the 'in.*' arrays of a golden."""
    def t(k):
        return torch.from_numpy(g[k]) if k in g else None
    p_targets = {k[len("in.p_targets."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("in.p_targets.")}
    return dict(
        speakers=t("in.speakers"), texts=t("in.texts"), src_lens=t("in.src_lens"),
        max_src_len=int(g["in.max_src_len"]), mels=t("in.mels"), mel_lens=t("in.mel_lens"),
        max_mel_len=int(g["in.max_mel_len"]) if "in.max_mel_len" in g else None,
        p_targets=p_targets or None, e_targets=t("in.e_targets"), d_targets=t("in.d_targets"),
        spker_embeds=t("in.spker_embeds"), attn_priors=t("in.attn_priors"),
    )


def synthetic_samples(n, seed, learn_alignment):
    """in-memory stand-ins for Dataset.__getitem__ outputs (dataset.py:52-146): ragged text / mel lengths, reference dtypes"""
    rng = np.random.RandomState(seed)
    out = []
    for i in range(n):
        P = int(rng.randint(3, 20))
        dur = rng.randint(1, 6, size=P)
        M = int(dur.sum())
        s = {"id": f"utt{i:03d}", "speaker": int(rng.randint(0, 4)), "text": rng.randint(1, 360, size=P), "raw_text": f"raw {i}",
             "mel": rng.randn(M, 80).astype(np.float32), "pitch": rng.randn(M).astype(np.float64), "f0": rng.randn(M),
             "uv": (rng.rand(M) > 0.7).astype(np.float64), "cwt_spec": rng.randn(M, 10).astype(np.float32),
             "f0_mean": float(rng.randn()), "f0_std": float(abs(rng.randn())), "energy": rng.randn(P if not learn_alignment else M).astype(np.float32),
             "duration": None if learn_alignment else dur, "mel2ph": None if learn_alignment else np.repeat(np.arange(1, P + 1), dur),
             "attn_prior": rng.rand(P, M).astype(np.float32) if learn_alignment else None, "spker_embed": None}
        out.append(s)
    return out
