"""MI355X-native hot path for keonlee9420/Comprehensive-Transformer-TTS (CompTransTTS
forward / train step + the audio/stft.py mel front-end) behind the reference's own Python
surface.  See DESIGN.md.  Import as `ctts_amd` (root-level shim: the directory name
required by the build contract is not a valid Python identifier).

Importing the package never needs the GPU or the built library; the first kernel call loads
comprehensive-transformer-tts_amd/csrc/libctts_hip.so and raises if it is missing (no fallback)."""
from . import configs, synthetic  # noqa: F401
from . import _lib, kernels, ops, model, audio, loss, dp, conformer  # noqa: F401
from .model import CompTransTTS, TextEncoder, Decoder, PostNet, VarianceAdaptor  # noqa: F401
from .audio import TacotronSTFT  # noqa: F401
