"""Shadow of the reference's `model` package (`/root/reference/model/__init__.py:1-4`): put THIS directory's parent on `sys.path`
in front of the reference checkout and the reference's own, UNMODIFIED entry points resolve to the MI355X-native hot path -

    PYTHONPATH=<repo>/dropin:<repo>  python train.py --dataset LJSpeech      # utils/model.py:8  `from model import CompTransTTS, ScheduledOptim`
                                                                             # train.py:19, evaluate.py:11  `from model import CompTransTTSLoss`

Same constructors, same 16-argument `forward`, same 14-tuple, same `state_dict()` keys and order (a reference checkpoint loads), same
`ScheduledOptim(model, train_config, model_config, current_step)`.  Only the names the train / evaluate / synthesize drivers import are
bound; `PreDefinedEmbedder` (DeepSpeaker speaker embeddings at preprocessing time, SURVEY section 2: out of scope) says so when touched.
"""
import os
import sys

_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _REPO not in sys.path:
    sys.path.insert(0, _REPO)

import ctts_amd  # noqa: E402
from ctts_amd.model import CompTransTTS  # noqa: E402,F401
from ctts_amd.loss import CompTransTTSLoss, ScheduledOptim  # noqa: E402,F401

__all__ = ["CompTransTTS", "CompTransTTSLoss", "ScheduledOptim"]


def __getattr__(name):
    if name == "PreDefinedEmbedder":
        raise ImportError("model.PreDefinedEmbedder (DeepSpeaker embeddings, preprocessing only) is outside the accelerated hot path: "
                          "import it from the reference checkout's model/speaker_embedder.py")
    raise AttributeError(f"module 'model' (ctts_amd drop-in) has no attribute '{name}'")
