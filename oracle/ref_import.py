"""TEST INFRASTRUCTURE (container only) - import the live reference from /root/reference.

Only `tests/golden/make_goldens.py` and `oracle/validate_against_reference.py` use this
module; it cannot run on the GPU box (`/root/reference` does not exist there) and the
product package never imports it.

Recipe follows SURVEY.md section 8(c): third-party modules that are absent from the
image and are *not* on the forward path (numba, librosa, parselmouth, ...) are replaced
by inert stubs before `import model`, and the process chdir()s into /root/reference
because the reference opens its configs / stats.json by relative path
(utils/tools.py:20, model/modules.py:795-797).
"""
import importlib.machinery
import os
import sys
import types

REF_ROOT = "/root/reference"


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        sub = _Stub(self.__name__ + "." + name)
        sub.__spec__ = importlib.machinery.ModuleSpec(sub.__name__, None)
        setattr(self, name, sub)
        return sub

    def __call__(self, *a, **k):
        return None


def _mk(name):
    m = _Stub(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    sys.modules[name] = m
    return m


def slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """Restatement of librosa==0.7.2 `filters.mel(sr, n_fft, n_mels, fmin, fmax)`
    (htk=False, norm=1 i.e. Slaney area normalisation) - the call at
    audio/stft.py:152-154.  librosa is not installed, so this is the published
    algorithm restated; the copy that travels to the GPU box is `ctts_amd.audio.slaney_mel_basis` (checked against this
    one through golden G8, which stores the basis the reference run used)."""
    import numpy as np

    def hz_to_mel(f):
        f = np.asanyarray(f, dtype=np.float64)
        f_sp = 200.0 / 3
        mels = f / f_sp
        min_log_hz = 1000.0
        min_log_mel = min_log_hz / f_sp
        logstep = np.log(6.4) / 27.0
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)

    def mel_to_hz(m):
        m = np.asanyarray(m, dtype=np.float64)
        f_sp = 200.0 / 3
        freqs = f_sp * m
        min_log_hz = 1000.0
        min_log_mel = min_log_hz / f_sp
        logstep = np.log(6.4) / 27.0
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)

    if fmax is None:
        fmax = sr / 2.0
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float64)
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2 : n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, None]
    return weights.astype(np.float32)


class _StubFinder:
    """meta-path hook: any submodule of a stubbed top-level package is a stub too."""
    roots = set()

    @classmethod
    def find_spec(cls, name, path=None, target=None):
        if name.split(".")[0] in cls.roots:
            return importlib.machinery.ModuleSpec(name, cls, is_package=True)
        return None

    @staticmethod
    def create_module(spec):
        m = _Stub(spec.name)
        m.__path__ = []
        return m

    @staticmethod
    def exec_module(module):
        pass


_installed = False


def install():
    """Make `import model`, `import audio.stft`, `import utils.tools` work."""
    global _installed
    if _installed:
        return
    import numpy as np
    import torch
    import torch._dynamo  # noqa: F401  (must be imported before the stubs exist)

    numba = _mk("numba")
    numba.jit = lambda *a, **k: (lambda f: f)
    numba.prange = range
    _StubFinder.roots.update(["tensorflow", "python_speech_features", "pyloudnorm", "tgt", "g2p_en",
                              "pypinyin", "parselmouth", "pyworld", "pycwt", "inflect", "tensorboard"])
    sys.meta_path.append(_StubFinder)
    for name in [
        "librosa", "librosa.util", "librosa.filters", "parselmouth", "pyworld", "pycwt",
        "unidecode", "inflect", "tensorflow", "tensorflow.keras", "python_speech_features",
        "pyloudnorm", "tgt", "g2p_en", "pypinyin", "tensorboard",
    ]:
        if name not in sys.modules:
            _mk(name)
    lib = sys.modules["librosa"]
    lib.util = sys.modules["librosa.util"]
    lib.filters = sys.modules["librosa.filters"]

    def pad_center(data, size, axis=-1, **kw):
        n = data.shape[axis]
        lpad = int((size - n) // 2)
        lengths = [(0, 0)] * data.ndim
        lengths[axis] = (lpad, int(size - n - lpad))
        return np.pad(data, lengths, **kw)

    lib.util.pad_center = pad_center
    lib.util.tiny = lambda x: np.finfo(np.float32).tiny
    lib.filters.mel = slaney_mel_filterbank
    sys.modules["unidecode"].unidecode = lambda s: s
    if not hasattr(np, "int"):
        np.int = int  # utils/pitch_tools.py:36 uses the removed alias (numpy path only)

    os.chdir(REF_ROOT)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    # audio/stft.py:75-76 and model/coordconv.py hard-code .cuda(); no GPU here.
    torch.Tensor.cuda = lambda self, *a, **k: self
    _installed = True


def load_configs(dataset="LJSpeech"):
    import yaml

    cfgs = []
    for n in ("preprocess", "model", "train"):
        with open(os.path.join(REF_ROOT, "config", dataset, n + ".yaml")) as f:
            cfgs.append(yaml.safe_load(f))
    pre, model, train = cfgs
    pre["preprocessing"]["pitch"]["cwt_scales"] = list(range(10))  # only len() is used (pitch_tools.py:260)
    pre["path"]["preprocessed_path"] = os.path.join(REF_ROOT, "preprocessed_data", dataset)
    return pre, model, train
