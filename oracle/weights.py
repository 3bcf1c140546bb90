"""TEST INFRASTRUCTURE - closed-form deterministic weight fill (no RNG, no checkpoint in git).

Every state-dict entry is filled from a splitmix64 hash of (crc32(key), element index), so
the golden generator (reference side, build container) and the tests (GPU box) regenerate
bit-identical parameters from the key names alone.  Scales keep activations O(1) and make
every "default-initialised-to-zero/one" tensor (LayerNorm/BatchNorm affine, biases,
pos_embed_alpha) non-trivial so that parity tests see their effect.
"""
import zlib

import numpy as np
import torch


def _hash_uniform(key, n):
    """-> float64 array of n values in [-1, 1)"""
    kh = np.uint64(zlib.crc32(key.encode()))
    with np.errstate(over="ignore"):
        z = (np.arange(n, dtype=np.uint64) + kh * np.uint64(0x100000001B3)) * np.uint64(0x9E3779B97F4A7C15)
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    return (z >> np.uint64(11)).astype(np.float64) / float(1 << 52) - 1.0


def fill_tensor(key, shape, dtype=torch.float32):
    n = int(np.prod(shape)) if len(shape) else 1
    u = _hash_uniform(key, n).reshape(shape)
    leaf = key.split(".")[-1]
    if key.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "running_var":
        v = 0.75 + 0.5 * (u + 1.0) / 2
    elif leaf == "running_mean":
        v = 0.2 * u
    elif leaf in ("pos_embed_alpha",):
        v = 1.0 + 0.2 * u
    elif leaf == "bias":
        v = 0.1 * u
    elif leaf == "weight" and len(shape) == 1:          # LayerNorm / BatchNorm gamma
        v = 1.0 + 0.2 * u
    elif "embed" in key or "emb." in key and len(shape) == 2 and "speaker" not in key:
        v = u * (3.0 / shape[1]) ** 0.5                  # ~ N(0, d^-0.5) scale
        v[0] = 0.0                                       # padding_idx row (blocks.py:13-14)
    else:
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        v = u * (3.0 / fan_in) ** 0.5
    return torch.from_numpy(np.ascontiguousarray(v)).to(dtype)


SKIP = ("_float_tensor", "energy_bins", "position_enc", "positional_encoding")


def closed_form_state_dict(template):
    """template: a state dict (tensors or shapes) with reference key names -> filled copy."""
    out = {}
    for k, v in template.items():
        if any(s in k for s in SKIP):
            out[k] = v.clone() if torch.is_tensor(v) else v
        else:
            out[k] = fill_tensor(k, tuple(v.shape))
    return out
