"""MI355X-native hot path for keonlee9420/Comprehensive-Transformer-TTS (CompTransTTS
forward / train step + the audio/stft.py mel front-end) behind the reference's own Python
surface.  See DESIGN.md.  Import as `ctts_amd` (root-level shim: the directory name
required by the build contract is not a valid Python identifier)."""
from . import configs, synthetic  # noqa: F401
