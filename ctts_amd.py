"""Import shim: `import ctts_amd` == the package in ./comprehensive-transformer-tts_amd/
(whose directory name is fixed by the build contract and is not a Python identifier)."""
import importlib
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
_REAL = "comprehensive-transformer-tts_amd"
_pkg = importlib.import_module(_REAL)


def _alias():
    for name, mod in list(sys.modules.items()):
        if name == _REAL or name.startswith(_REAL + "."):
            sys.modules["ctts_amd" + name[len(_REAL):]] = mod


_alias()
_pkg._alias_submodules = _alias
sys.modules[__name__] = _pkg
