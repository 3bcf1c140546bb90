"""`audio.tools.get_mel_from_wav` (`/root/reference/audio/tools.py:8-15`) -> ctts_amd.audio.get_mel_from_wav"""
from ctts_amd.audio import get_mel_from_wav  # noqa: F401
