import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _reset_global_op_switches():
    """ops.set_grad_accumulation_fusion is process-global (trainer.TrainStep turns it on): every test starts from plain autograd"""
    yield
    try:
        from ctts_amd import ops
        ops.set_grad_accumulation_fusion(False)
        ops.set_wgrad_stream(None)
        ops.set_fused_attention(None)
    except Exception:      # noqa: BLE001
        pass
