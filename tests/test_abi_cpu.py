"""CPU: the C-ABI library builds/loads and exports every symbol include/ctts.h declares; the
Python surface keeps the reference contract; the product has no CPU path."""
import ctypes
import inspect
import os
import re

import pytest
import torch

import ctts_amd
from ctts_amd import _lib
from ctts_amd.configs import get_configs
from ctts_amd.synthetic import make_batch, as_model_args
from tests.util import schema

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    hdr = open(os.path.join(ROOT, "include", "ctts.h")).read()
    declared = set(re.findall(r"\b(ctts_[a-z0-9_]+)\s*\(", hdr))
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in ctts.h but not exported"
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    assert ctypes.sizeof(_lib.GemmDesc) > 0
    lib.ctts_version.restype = ctypes.c_int
    assert lib.ctts_version() >= 1


def test_forward_signature_matches_reference_contract():
    sig = inspect.signature(ctts_amd.CompTransTTS.forward)
    names = list(sig.parameters)[1:]
    assert names == ["speakers", "texts", "src_lens", "max_src_len", "mels", "mel_lens", "max_mel_len", "p_targets",
                     "e_targets", "d_targets", "attn_priors", "spker_embeds", "p_control", "e_control", "d_control", "step"]
    assert list(inspect.signature(ctts_amd.CompTransTTS.__init__).parameters)[1:] == ["preprocess_config", "model_config",
                                                                                     "train_config"]


@pytest.mark.parametrize("dataset", ["LJSpeech", "VCTK"])
def test_state_dict_schema_matches_reference(dataset):
    pre, mc, tc = get_configs(dataset)
    m = ctts_amd.CompTransTTS(pre, mc, tc)
    sd = m.state_dict()
    sch = schema(dataset)
    assert set(sd) == set(sch)
    params = dict(m.named_parameters())
    for k, (shape, dtype, is_param) in sch.items():
        assert list(sd[k].shape) == shape, k
        assert str(sd[k].dtype) == "torch." + dtype, k
        assert (k in params) == is_param, k
    assert sum(p.numel() for p in m.parameters()) == (35094225 if dataset == "LJSpeech" else 35225553)
    enc, dec = ctts_amd.TextEncoder(mc), ctts_amd.Decoder(mc)
    assert enc.d_model == 256 and dec.d_model == 256        # block_type plugin surface


def test_conformer_state_dict_schema_matches_reference():
    pre, mc, tc = get_configs()
    mc["block_type"] = "conformer"
    m = ctts_amd.CompTransTTS(pre, mc, tc)
    sd, sch = m.state_dict(), schema("LJSpeech", "conformer")
    assert set(sd) == set(sch), (set(sd) ^ set(sch))
    params = dict(m.named_parameters())
    for k, (shape, dtype, is_param) in sch.items():
        assert list(sd[k].shape) == shape, k
        assert (k in params) == is_param, k
    assert sum(p.numel() for p in m.parameters()) == 22623952        # SURVEY.md headline facts
    # the sinusoid table is ONE Parameter object shared by the stack and every attention module (conformer.py:322)
    assert m.encoder.layer_stack[2].sequential.__getattr__("1").module.positional_encoding is m.encoder.position_enc
    assert m.encoder.d_model == 256 and m.decoder.d_model == 256


@pytest.mark.parametrize("unsup", [False, True])
def test_liu2021_state_dict_schema_matches_reference(unsup):
    """prosody_modeling.model_type = liu2021, alone and as config C5 (with learn_alignment=True)."""
    pre, mc, tc = get_configs()
    mc["prosody_modeling"]["model_type"] = "liu2021"
    mc["duration_modeling"]["learn_alignment"] = unsup
    m = ctts_amd.CompTransTTS(pre, mc, tc)
    sd, sch = m.state_dict(), schema("LJSpeech", "transformer_fs2", unsup, "liu2021")
    assert set(sd) == set(sch), (set(sd) ^ set(sch))
    params = dict(m.named_parameters())
    for k, (shape, dtype, is_param) in sch.items():
        assert list(sd[k].shape) == shape, k
        assert str(sd[k].dtype) == "torch." + dtype, k
        assert (k in params) == is_param, k
        assert torch.isfinite(sd[k].float()).all(), k
    mc["prosody_modeling"]["model_type"] = "du2021"
    with pytest.raises(NotImplementedError):
        ctts_amd.CompTransTTS(pre, mc, tc)


def test_vctk_default_yaml_schema_matches_reference():
    """config/VCTK/model.yaml as shipped: multi_speaker + learn_alignment=True (aligner incl. key/query speaker projections)."""
    pre, mc, tc = get_configs("VCTK")
    mc["duration_modeling"]["learn_alignment"] = True
    m = ctts_amd.CompTransTTS(pre, mc, tc)
    sd, sch = m.state_dict(), schema("VCTK", "transformer_fs2", True)
    assert set(sd) == set(sch), (set(sd) ^ set(sch))
    for k, (shape, dtype, is_param) in sch.items():
        assert list(sd[k].shape) == shape, k
        assert torch.isfinite(sd[k].float()).all(), k


def test_unsupported_block_types_raise():
    pre, mc, tc = get_configs()
    mc["block_type"] = "reformer"
    with pytest.raises(NotImplementedError):
        ctts_amd.CompTransTTS(pre, mc, tc)
    mc["block_type"] = "nope"
    with pytest.raises(NotImplementedError):
        ctts_amd.CompTransTTS(pre, mc, tc)


def test_no_cpu_fallback_in_product():
    pre, mc, tc = get_configs()
    m = ctts_amd.CompTransTTS(pre, mc, tc)
    with pytest.raises(Exception) as ei:
        m(*as_model_args(make_batch([8, 5], 4)))
    assert "CPU" in str(ei.value) or "device" in str(ei.value).lower()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "comprehensive-transformer-tts_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# checker", ""), f


@pytest.mark.parametrize("dataset,block,unsup,prosody", [
    ("LJSpeech", "transformer_fs2", False, "none"), ("VCTK", "transformer_fs2", False, "none"), ("LJSpeech", "conformer", False, "none"),
    ("LJSpeech", "transformer_fs2", True, "none"), ("LJSpeech", "transformer_fs2", False, "liu2021"),
    ("LJSpeech", "transformer_fs2", True, "liu2021"), ("VCTK", "transformer_fs2", True, "none")])
def test_state_dict_and_parameter_ORDER_match_reference(dataset, block, unsup, prosody):
    """The optimizer half of a reference checkpoint is keyed on POSITIONS in `model.parameters()` (utils/model.py:22-26,
    optimizer.py:8-14), so not only the key set but the registration order must be the reference's (captured schemas keep the
    reference's state_dict order)."""
    import os
    from tests.util import GOLDEN
    tag = f"{dataset}_{block}{'_unsup' if unsup else ''}{'' if prosody == 'none' else '_' + prosody}"
    if not os.path.exists(os.path.join(GOLDEN, f"state_dict_schema_{tag}.json")):
        pytest.skip(f"no captured schema for {tag}")
    pre, mc, tc = get_configs(dataset)
    mc["block_type"] = block
    mc["duration_modeling"]["learn_alignment"] = unsup
    mc["prosody_modeling"]["model_type"] = prosody
    m = ctts_amd.CompTransTTS(pre, mc, tc)
    sch = schema(dataset, block, unsup, prosody)
    assert list(m.state_dict().keys()) == list(sch.keys())
    ref_params = [k for k, (_, _, is_param) in sch.items() if is_param]
    # named_parameters() yields a shared Parameter once (first registration); the captured schema marks parameters the same way
    assert [k for k, _ in m.named_parameters()] == ref_params
    # Adam built the reference's way covers ALL parameters (frozen ones included) in that order
    from ctts_amd.loss import ScheduledOptim
    opt = ScheduledOptim(m, tc, mc, 0)
    assert len(opt._optimizer.param_groups[0]["params"]) == len(list(m.parameters()))
