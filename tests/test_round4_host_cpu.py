"""CPU: host-side logic added in round 4 that needs no GPU - configuration switches of the model constructor (state-dict keys of the
variants), the mask cache, argument checks of ops.conv1d, the reference loop's grad_acc_step clip pattern."""
import pytest
import torch

import ctts_amd
from ctts_amd import model as M
from ctts_amd import ops
from ctts_amd.configs import get_configs
from tests.util import schema


def _model(vp=None, ve=None):
    pre, mc, tc = get_configs()
    mc["variance_predictor"].update(vp or {})
    mc["variance_embedding"].update(ve or {})
    return ctts_amd.CompTransTTS(pre, mc, tc)


@pytest.mark.parametrize("suffix,vp,ve", [("_swish_left", dict(ffn_act="swish", ffn_padding="LEFT"), None),
                                          ("_relu", dict(ffn_act="relu"), None),
                                          ("_noembed", None, dict(use_pitch_embed=False, use_energy_embed=False)),
                                          ("_nopitch", None, dict(use_pitch_embed=False))])
def test_config_variants_expose_the_reference_state_dict_keys_in_order(suffix, vp, ve):
    """the schema files were written by tests/golden/make_goldens.py from the live reference model of each variant"""
    m = _model(vp, ve)
    want = schema(suffix=suffix)
    assert list(m.state_dict().keys()) == list(want.keys())
    for k, v in m.state_dict().items():
        assert list(v.shape) == want[k][0], k


def test_ffn_switch_parsing():
    assert M._ffn_switches({"variance_predictor": {"ffn_act": "gelu", "ffn_padding": "SAME"}}) == (ops.ACT_GELU, "SAME")
    assert M._ffn_switches({"variance_predictor": {"ffn_act": "swish", "ffn_padding": "LEFT"}}) == (ops.ACT_SWISH, "LEFT")
    assert M._ffn_switches({"variance_predictor": {"ffn_act": "relu"}}) == (ops.ACT_RELU, "SAME")
    # transformer_fs2.py:228-233: a string that is none of gelu / relu / swish applies no activation at all
    assert M._ffn_switches({"variance_predictor": {"ffn_act": "none"}})[0] == ops.ACT_NONE
    with pytest.raises(NotImplementedError):
        M._ffn_switches({"variance_predictor": {"ffn_padding": "VALID"}})
    with pytest.raises(NotImplementedError):
        _model(ve=None, vp=dict(ffn_padding="RIGHT"))


def test_mask_aux_is_computed_once_per_mask_and_takes_known_lengths():
    lens = torch.tensor([5, 3, 0])
    mask = torch.arange(6)[None, :] >= lens[:, None]
    nonpad, li = M.mask_aux(mask)
    assert nonpad.dtype == torch.float32 and nonpad.shape == (18,) and li.dtype == torch.int32
    assert li.tolist() == [5, 3, 0] and nonpad.view(3, 6).sum(1).tolist() == [5.0, 3.0, 0.0]
    assert M.mask_aux(mask)[0] is nonpad and M.mask_aux(mask)[1] is li            # cached on the mask object
    mask2 = torch.arange(6)[None, :] >= lens[:, None]
    li2 = M.mask_aux(mask2, lens)[1]
    assert li2.tolist() == [5, 3, 0] and M.mask_aux(mask2)[1] is li2


def test_conv1d_rejects_unknown_padding_before_any_launch():
    x, w = torch.zeros(1, 4, 8), torch.zeros(8, 8, 3)
    with pytest.raises(ValueError, match="padding"):
        ops.conv1d(x, w, padding="VALID")


def test_grad_acc_step_clip_pattern_follows_train_py():
    """train.py:118 clips when step % grad_acc_step == 0 (and steps / zeroes every iteration)"""
    from ctts_amd.trainer import TrainStep
    ts = TrainStep.__new__(TrainStep)
    for k, want in ((1, [True] * 4), (2, [False, True, False, True]), (3, [True, False, False, True])):
        ts.grad_acc_step = k
        got = []
        for step in (50001, 50002, 50003, 50004):
            ts.step_no = step
            got.append(ts._clip_now())
        assert got == want, (k, got)


def test_phoneme_level_pitch_matches_the_reference_scatter_formulation():
    """model.phoneme_level_pitch (fp64 one-hot contraction, device-side) against the oracle's per-utterance scatter_add
    (utils/tools.py:47-53 / modules.py:873-880), including 0-frame phonemes and frames beyond mel_len."""
    from oracle import restate as R
    g = torch.Generator().manual_seed(3)
    src_lens, Ts, Tm = torch.tensor([7, 4, 5]), 7, 30
    dur = torch.zeros(3, Ts, dtype=torch.long)
    for b, n in enumerate(src_lens.tolist()):
        dur[b, :n] = torch.randint(0, 6, (n,), generator=g)
        dur[b, 0] = max(int(dur[b, 0]), 1)
    mel_lens = dur.sum(1)
    mel2ph = torch.zeros(3, Tm, dtype=torch.long)
    for b in range(3):
        mel2ph[b, : int(mel_lens[b])] = torch.repeat_interleave(torch.arange(1, Ts + 1), dur[b])
    f0 = torch.randn(3, Tm, generator=g) + 7.0            # NOT zeroed beyond mel_len: those frames must not count
    want = R.phoneme_level_pitch(Ts, src_lens, mel2ph, mel_lens, f0)
    got = M.phoneme_level_pitch(f0, mel2ph, mel_lens, Ts)
    assert got.shape == want.shape == (3, Ts)
    assert (got - want).abs().max().item() <= 1e-6
    assert (got[dur == 0] == 0).all()


@pytest.mark.parametrize("norm,use_uv", [("log", True), ("log", False), ("standard", True)])
def test_denorm_f0_matches_pitch_tools(norm, use_uv):
    """model.denorm_f0 (torch.where masks) == the oracle's in-place restatement of utils/pitch_tools.py:69-82"""
    from oracle import restate as R
    g = torch.Generator().manual_seed(4)
    cfg = {"pitch_norm": norm, "use_uv": use_uv, "f0_mean": 7.4, "f0_std": 0.35}
    f0 = torch.randn(2, 9, generator=g)
    uv = (torch.rand(2, 9, generator=g) < 0.4).float()
    pad = torch.arange(9)[None, :] >= torch.tensor([9, 6])[:, None]
    got = M.denorm_f0(f0, uv, cfg, pad)
    want = R.denorm_f0(f0.clone(), uv, cfg, pad)
    assert torch.equal(got, want)
    assert torch.equal(M.denorm_f0(f0, None, cfg, torch.tensor(False)), R.denorm_f0(f0.clone(), None, cfg, torch.tensor(False)))


def test_pitch_and_energy_switches_that_the_reference_cannot_run_raise():
    pre, mc, tc = get_configs()
    pre["preprocessing"]["pitch"]["pitch_ar"] = True
    with pytest.raises(NotImplementedError, match="pitch_ar"):
        ctts_amd.CompTransTTS(pre, mc, tc)
    pre, mc, tc = get_configs()
    pre["preprocessing"]["pitch"]["pitch_type"] = "dct"
    with pytest.raises(NotImplementedError, match="pitch_type"):
        ctts_amd.CompTransTTS(pre, mc, tc)
    pre, mc, tc = get_configs()
    pre["preprocessing"]["energy"]["feature"] = "word_level"
    with pytest.raises(NotImplementedError, match="energy"):
        ctts_amd.CompTransTTS(pre, mc, tc)
    pre, mc, tc = get_configs()
    pre["preprocessing"]["pitch"]["pitch_type"] = "frame"
    tc["loss"]["pitch_loss"] = "ssim"
    from ctts_amd.loss import CompTransTTSLoss
    with pytest.raises(NotImplementedError, match="pitch_loss"):
        CompTransTTSLoss(pre, mc, tc)


def test_bench_pmc_parser_and_live_traffic_guards(tmp_path, monkeypatch):
    """bench.py's live roofline.traffic: per-launch average of a counter summed over its hardware instances from a rocpd result, and the
    guards that make it fall back to the committed passes (no rocprofv3 / already under a profiler)."""
    import importlib.util
    import os
    import sqlite3
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    dbp = str(tmp_path / "x_results.db")
    db = sqlite3.connect(dbp)
    db.execute("create table counters_collection (dispatch_id int, kernel_name text, counter_name text, value real)")
    rows = []
    for disp in (1, 2):                     # two launches of the kernel, 8 instances each
        rows += [(disp, "void (anonymous namespace)::gemm_sk_kernel<true, true>(x)", "FETCH_SIZE", 100.0 * disp)] * 8
    rows += [(3, "other_kernel", "FETCH_SIZE", 7.0), (1, "gemm_sk_kernel<..>", "WRITE_SIZE", 5.0)]
    db.executemany("insert into counters_collection values (?,?,?,?)", rows)
    db.commit()
    db.close()
    v, n = bench._pmc_per_launch(dbp, "FETCH_SIZE")
    assert n == 2 and v == (800.0 + 1600.0) / 2
    with pytest.raises(RuntimeError, match="no SQ_WAVES rows"):
        bench._pmc_per_launch(dbp, "SQ_WAVES")
    monkeypatch.setenv("ROCPROFILER_REGISTER_FORCE_LOAD", "1")
    monkeypatch.setattr("shutil.which", lambda name: "/opt/rocm/bin/rocprofv3")
    with pytest.raises(RuntimeError, match="under a profiler"):
        bench.measure_traffic_live()
    monkeypatch.delenv("ROCPROFILER_REGISTER_FORCE_LOAD")
    monkeypatch.setattr("shutil.which", lambda name: None)
    with pytest.raises(RuntimeError, match="not on PATH"):
        bench.measure_traffic_live()
