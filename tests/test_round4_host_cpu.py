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
