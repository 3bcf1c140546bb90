"""CPU: host data path - ctts_amd.data.collate / reprocess reproduce the reference's Dataset.collate_fn + to_device
(golden G11, captured from dataset.py:166-248 and utils/tools.py:69-134), and PackedBatch round-trips every field through ONE buffer."""
import numpy as np
import pytest
import torch

from ctts_amd import data as D
from tests.util import load_golden, synthetic_samples


def _flat(ref_list):
    flat = {"speakers": ref_list[2], "texts": ref_list[3], "src_lens": ref_list[4], "mels": ref_list[6], "mel_lens": ref_list[7],
            "energies": ref_list[10], "durations": ref_list[11], "attn_priors": ref_list[12]}
    flat.update({"pitch." + k: v for k, v in ref_list[9].items()})
    return {k: v for k, v in flat.items() if v is not None}


@pytest.mark.parametrize("tag,la", [("sup", False), ("unsup", True)])
def test_collate_and_pack_match_reference(tag, la):
    g = load_golden("g11_collate")
    samples = synthetic_samples(10, 5 + int(la), la)
    batches = D.collate(samples, 4, sort=True, drop_last=False, learn_alignment=la)
    assert len(batches) == int(g[f"{tag}.n_batches"]) == 3               # 4 + 4 + tail of 2
    for bi, b in enumerate(batches):
        assert list(b[0]) == list(g[f"{tag}.b{bi}.ids"])
        pb = D.PackedBatch.pack(b, pin=False)
        ref_list = pb.host_views()
        assert ref_list[5] == int(g[f"{tag}.b{bi}.max_src_len"]) and ref_list[8] == int(g[f"{tag}.b{bi}.max_mel_len"])
        flat = _flat(ref_list)
        want = {k[len(f"{tag}.b{bi}."):] for k in g if k.startswith(f"{tag}.b{bi}.")} - {"ids", "max_src_len", "max_mel_len"}
        assert set(flat) == want, (set(flat) ^ want)
        for k, v in flat.items():
            e = g[f"{tag}.b{bi}.{k}"]
            assert str(v.dtype).replace("torch.", "") == str(e.dtype), (k, v.dtype, e.dtype)      # device dtype of to_device
            assert np.array_equal(v.numpy(), e), k
        for o, shape, dt in pb.layout.values():
            assert o % 256 == 0
    assert len(D.collate(samples, 4, sort=True, drop_last=True, learn_alignment=la)) == 2


def test_to_device_has_no_cpu_path():
    b = D.collate(synthetic_samples(4, 1, False), 4)[0]
    with pytest.raises(RuntimeError):
        D.PackedBatch.pack(b, pin=False).to_device("cpu")
