"""Drop-in `TacotronSTFT` (reference: audio/stft.py:137-185) on the gfx950 kernels.

mel_spectrogram(y[B,N] in [-1,1]) -> (mel[B,80,F], energy[B,F]),  F = 1 + N // hop.
For the reference's filter_length = 1024 the whole front end is ONE launch (csrc/mel.hip): a 1024-point real FFT per frame, hann
window and reflect padding folded into the load, |X| + energy in registers, the mel filterbank as a banded fp32-MFMA GEMM,
log-clamp fused.  Other FFT sizes take the DFT-as-GEMM path (the [2*(n_fft/2+1), n_fft] windowed basis the reference feeds
to F.conv1d, stft.py:32-56: rows = real parts then imaginary parts, periodic hann window).

PARITY UNPINNED for one ingredient: the mel filterbank restates the published algorithm of librosa==0.7.2 `filters.mel` (Slaney scale,
area normalisation, htk=False), which the reference pulls from a third-party dependency (requirements.txt:9) that is absent from
/root/reference and from this image.  The golden `tests/golden/g8_stft.npz` was captured from the reference's own `TacotronSTFT`
running on THIS restated basis (oracle/ref_import.py stubs librosa with it), so G8 pins the STFT / magnitude / log arithmetic against
the reference, but not the basis.  What pins the basis is tests/test_mel_basis_cpu.py: the Slaney mel scale against the worked examples
of librosa's own documentation (hz_to_mel / mel_to_hz / mel_frequencies(n_mels=40), all 40 values), and the triangle construction with
the 2 / (f[i+2] - f[i]) area normalisation against its published definition.  Swap in librosa's array (`stft.mel_basis.copy_(...)`)
where it is available.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import ops


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax):
    if fmax is None:
        fmax = sr / 2.0
    nb = 1 + n_fft // 2
    fftfreqs = np.linspace(0, sr / 2.0, nb)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    w = np.zeros((n_mels, nb))
    for i in range(n_mels):
        w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def padded_window(n_fft, win_length):
    """scipy get_window('hann', fftbins=True) zero-padded (centred) to n_fft (stft.py:46-50), float32"""
    n = np.arange(win_length)
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)
    lpad = (n_fft - win_length) // 2
    return np.pad(win, (lpad, n_fft - win_length - lpad)).astype(np.float32)


def windowed_dft_basis(n_fft, win_length):
    """[2*(n_fft/2+1), n_fft]: rows [0,nb) = cos, [nb,2nb) = -sin, times the periodic hann window
    zero-padded (centred) to n_fft - the matrix of audio/stft.py:32-56."""
    nb = n_fft // 2 + 1
    four = np.fft.fft(np.eye(n_fft))
    basis = np.vstack([np.real(four[:nb, :]), np.imag(four[:nb, :])])
    n = np.arange(win_length)
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)       # scipy get_window('hann', fftbins=True)
    lpad = (n_fft - win_length) // 2
    win = np.pad(win, (lpad, n_fft - win_length - lpad))
    return (basis.astype(np.float32) * win.astype(np.float32)[None, :]).astype(np.float32)


class TacotronSTFT(nn.Module):
    def __init__(self, filter_length, hop_length, win_length, n_mel_channels, sampling_rate, mel_fmin, mel_fmax):
        super().__init__()
        self.n_fft, self.hop, self.n_mel_channels, self.sampling_rate = filter_length, hop_length, n_mel_channels, sampling_rate
        self.nbins = filter_length // 2 + 1
        mel = slaney_mel_basis(sampling_rate, filter_length, n_mel_channels, mel_fmin, mel_fmax)
        self.register_buffer("mel_basis", torch.from_numpy(mel))
        ld = (self.nbins + 3) // 4 * 4
        padded = np.zeros((n_mel_channels, ld), dtype=np.float32)
        padded[:, :self.nbins] = mel
        self.register_buffer("_mel_basis_padded", torch.from_numpy(padded), persistent=False)
        self.register_buffer("_dft_basis", torch.from_numpy(windowed_dft_basis(filter_length, win_length)), persistent=False)
        self.register_buffer("_window", torch.from_numpy(padded_window(filter_length, win_length)), persistent=False)
        self._fft_ws = None          # device workspace of the FFT kernel (twiddles, transposed filterbank), built on first use
        self.use_fft = filter_length == 1024 and n_mel_channels <= 96
        nz = np.nonzero(np.abs(mel).sum(0))[0]
        self._kmax = int(nz.max()) + 1 if len(nz) else 0       # bins above fmax carry no filter weight: not kept on chip
        self._range = None           # range check of the FFT path: (flag in pinned host memory, event of the last launch)
        # strict_range = True (default): the reference's contract - the assertion is raised by the SAME call that was handed the bad
        # waveform (one event wait per call instead of the reference's two reductions + sync).  False: the training / bulk-extraction
        # hot path - the flag is looked at without waiting at the next call, or on demand with check_range() (call it before using the
        # last result).
        self.strict_range = True

    # ---- the reference asserts min(y) >= -1 and max(y) <= 1 on the host before computing (stft.py:177-178): two reductions and a
    # device -> host synchronisation per call, which halves the throughput of a 50 us kernel.  Here the kernel raises a flag while it
    # loads the samples - one word of pinned host memory, written only by an offending input.  By default (strict_range) the call waits
    # for its own launch and raises like the reference; with strict_range = False the flag is looked at, without waiting, at the NEXT
    # call, or on demand with check_range() - same AssertionError, raised one call later at the latest (a single or final
    # call must not return a mel for NaN / out-of-range audio silently, so deferral is opt-in).
    def _range_state(self, dev):
        if self._range is None:
            self._range = (torch.zeros(1, dtype=torch.int32).pin_memory(), torch.cuda.Event())
        return self._range

    def _raise_if_out_of_range(self, wait):
        if self._range is None:
            return
        host, ev = self._range
        if wait:
            ev.synchronize()
        elif not ev.query():
            return
        bits = int(host[0]) & 0xFFFFFFFF
        if bits:
            host.zero_()
            val = np.array([bits], dtype=np.uint32).view(np.float32)[0]
            raise AssertionError(f"TacotronSTFT.mel_spectrogram: waveform outside [-1, 1] (largest |sample| {val})  [audio/stft.py:177-178]")

    def check_range(self):
        """block until the range flag of every mel_spectrogram call issued so far has arrived and raise if a waveform left [-1, 1]"""
        self._raise_if_out_of_range(wait=True)

    def _workspace(self):
        from . import kernels as K
        if self._fft_ws is None or self._fft_ws.device != self.mel_basis.device:
            self._fft_ws = K.mel_prepare(self.mel_basis.contiguous(), self.n_fft)
        return self._fft_ws

    def mel_spectrogram(self, y):
        """y [B,N] float32 on the HIP device, values in [-1, 1] (asserted like stft.py:177-178)."""
        if not y.is_cuda:
            raise RuntimeError("TacotronSTFT (ctts_amd) computes on the MI355X: pass a device tensor")
        if self._dft_basis.device != y.device:
            self.to(y.device)
        if self.use_fft:
            from . import kernels as K
            self._raise_if_out_of_range(wait=False)          # earlier calls: never blocks
            host, ev = self._range_state(y.device)
            mel, energy, _ = K.mel_spectrogram_fft(y.float().contiguous(), self._window, self._workspace(), self.n_fft, self.hop,
                                                   self.n_mel_channels, kmax=self._kmax, range_flag=host)
            ev.record()
            if self.strict_range:
                self._raise_if_out_of_range(wait=True)
            return mel, energy
        lo, hi = torch.aminmax(y.detach())              # DFT-as-GEMM path (other FFT sizes): one reduction + one sync for the two asserts
        assert lo >= -1 and hi <= 1
        mel, energy, _ = ops.mel_spectrogram(y.float(), self._dft_basis, self._mel_basis_padded, self.n_fft, self.hop,
                                             self.n_mel_channels, self.nbins)
        return mel, energy

    def magnitudes(self, y):
        if self._dft_basis.device != y.device:
            self.to(y.device)
        if self.use_fft:
            from . import kernels as K
            _, _, mag = K.mel_spectrogram_fft(y.float().contiguous(), self._window, self._workspace(), self.n_fft, self.hop,
                                              self.n_mel_channels, want_mag=True, kmax=self._kmax)
            return mag.view(y.shape[0], -1, mag.shape[-1])[:, :, :self.nbins].transpose(1, 2)
        _, _, mag = ops.mel_spectrogram(y.float(), self._dft_basis, self._mel_basis_padded, self.n_fft, self.hop,
                                        self.n_mel_channels, self.nbins)
        B = y.shape[0]
        return mag.view(B, -1, mag.shape[-1])[:, :, :self.nbins].transpose(1, 2)

    def mel_spectrograms_ragged(self, wavs):
        """Batched mel extraction for preprocessing (reference: preprocessor.py:387,467 -> audio/tools.py:8-15, one utterance per call,
        B = 1): `wavs` = list of 1-D float arrays / tensors of different lengths.  ONE pinned host buffer, one H2D copy, ONE kernel
        launch with per-utterance lengths (reflection at each utterance's own end), one D2H copy; returns the per-utterance
        (mel [n_mel, F_b] float32 numpy, energy [F_b]) exactly as `get_mel_from_wav` would, values clipped to [-1, 1] like it does."""
        from . import kernels as K
        if not self.use_fft:
            raise NotImplementedError("mel_spectrograms_ragged needs the FFT kernel (filter_length 1024)")
        dev = self.mel_basis.device
        if dev.type != "cuda":
            raise RuntimeError("TacotronSTFT (ctts_amd) computes on the MI355X: move the module to the device first")
        lens = [int(len(w)) for w in wavs]
        if min(lens) <= self.n_fft // 2:
            raise ValueError("reflect padding needs more than n_fft/2 samples per utterance")
        nmax = (max(lens) + 1) // 2 * 2
        host = torch.zeros(len(wavs), nmax, dtype=torch.float32, pin_memory=True)
        for i, w in enumerate(wavs):
            host[i, :lens[i]] = torch.as_tensor(np.asarray(w, dtype=np.float32)).clamp_(-1, 1)
        y = host.to(dev, non_blocking=True)
        lt = torch.tensor(lens, dtype=torch.int32).to(dev, non_blocking=True)
        mel, energy, _ = K.mel_spectrogram_fft(y, self._window, self._workspace(), self.n_fft, self.hop, self.n_mel_channels,
                                               kmax=self._kmax, lens=lt)
        mel, energy = mel.cpu().numpy(), energy.cpu().numpy()
        out = []
        for i, n in enumerate(lens):
            fb = 1 + n // self.hop
            out.append((np.ascontiguousarray(mel[i, :, :fb]), np.ascontiguousarray(energy[i, :fb])))
        return out


def get_mel_from_wav(audio, _stft):
    """audio/tools.py:8-15: one utterance (numpy) -> (mel [n_mel, F], energy [F]) float32 numpy, input clipped to [-1, 1]"""
    return _stft.mel_spectrograms_ragged([audio])[0]
