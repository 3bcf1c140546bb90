"""`audio.stft.TacotronSTFT` (`/root/reference/audio/stft.py:137-185`) -> ctts_amd.audio.TacotronSTFT (csrc/mel.hip)"""
from ctts_amd.audio import TacotronSTFT  # noqa: F401
