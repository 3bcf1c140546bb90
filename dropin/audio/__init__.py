"""Shadow of the reference's `audio` package for the mel front end (`/root/reference/audio/stft.py:137-185`, `audio/tools.py:8-15`):
`import audio as Audio; Audio.stft.TacotronSTFT(...)`, `Audio.tools.get_mel_from_wav(wav, stft)` as `preprocessor/preprocessor.py:38-46,
232` uses them, resolved to the HIP mel kernel.  `inv_mel_spec` (Griffin-Lim, a demo helper) is not on the hot path and is not bound."""
import os
import sys

_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _REPO not in sys.path:
    sys.path.insert(0, _REPO)

from . import stft, tools  # noqa: E402,F401
