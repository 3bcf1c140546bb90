"""CPU: the Slaney mel filterbank restated in ctts_amd/audio.py (the reference takes it from librosa==0.7.2 `filters.mel`, a dependency
that is absent from /root/reference and from this image).  Known-answer vectors: the worked examples printed in librosa's own API
documentation for `hz_to_mel`, `mel_to_hz` and `mel_frequencies(n_mels=40)` (Slaney scale, htk=False) - these pin the mel scale, the
one non-trivial ingredient; the triangle construction and the area normalisation are checked structurally against their published
definition.  (G8 pins the STFT arithmetic of the reference, not this basis: see the audio.py docstring.)"""
import numpy as np

from ctts_amd.audio import _hz_to_mel, _mel_to_hz, slaney_mel_basis

# librosa.mel_frequencies(n_mels=40) as printed in the librosa documentation (fmin = 0, fmax = 11025, htk = False)
DOC_MEL_FREQUENCIES_40 = [
    0.0, 85.317, 170.635, 255.952, 341.269, 426.586, 511.904, 597.221, 682.538, 767.855, 853.173, 938.49, 1024.856, 1119.114, 1222.042,
    1334.436, 1457.167, 1591.187, 1737.532, 1897.337, 2071.84, 2262.393, 2470.47, 2697.686, 2945.799, 3216.731, 3512.582, 3835.643,
    4188.417, 4573.636, 4994.285, 5453.621, 5955.205, 6502.92, 7101.009, 7754.107, 8467.272, 9246.028, 10096.408, 11025.0]


def test_mel_scale_matches_librosa_documented_examples():
    assert abs(float(_hz_to_mel(60)) - 0.9) < 1e-12                       # librosa.hz_to_mel(60) -> 0.9
    assert np.allclose(_hz_to_mel(np.array([110.0, 220.0, 440.0])), [1.65, 3.3, 6.6], atol=1e-12)
    assert abs(float(_mel_to_hz(3)) - 200.0) < 1e-9                       # librosa.mel_to_hz(3) -> 200.
    assert np.allclose(_mel_to_hz(np.array([1.0, 2.0, 3.0, 4.0, 5.0])), [66.667, 133.333, 200.0, 266.667, 333.333], atol=1e-3)
    f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(11025.0), 40))
    assert np.abs(f - np.array(DOC_MEL_FREQUENCIES_40)).max() < 6e-4     # printed with 3 decimals


def test_filterbank_structure_matches_the_published_definition():
    sr, n_fft, n_mels, fmin, fmax = 22050, 1024, 80, 0.0, 8000.0
    w = slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax).astype(np.float64)
    assert w.shape == (80, 513) and (w >= 0).all()
    freqs = np.linspace(0, sr / 2, 513)
    pts = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    for i in range(n_mels):
        nz = np.nonzero(w[i])[0]
        assert len(nz) > 0 and (np.diff(nz) == 1).all()                   # one contiguous triangle
        assert freqs[nz[0]] > pts[i] - 1e-9 and freqs[nz[-1]] < pts[i + 2] + 1e-9
        # weights of filter i: min(rising, falling) ramp, scaled by 2 / (f[i+2] - f[i]) (Slaney area normalisation)
        up = (freqs - pts[i]) / (pts[i + 1] - pts[i])
        down = (pts[i + 2] - freqs) / (pts[i + 2] - pts[i + 1])
        want = np.maximum(0, np.minimum(up, down)) * 2.0 / (pts[i + 2] - pts[i])
        assert np.abs(w[i] - want).max() < 1e-9
    # bins above fmax carry no weight (the kernel skips them: 372 of 513 bins for fmax 8 kHz)
    assert w[:, freqs > fmax].sum() == 0 and np.nonzero(w.sum(0))[0].max() <= 372
