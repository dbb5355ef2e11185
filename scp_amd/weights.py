"""Deterministic synthetic weights (no trained checkpoint is available offline).

`fill_weights(module, seed)` overwrites every floating-point entry of `module.state_dict()`
from a numpy PCG64 stream seeded by (seed, crc32(key)), so the reference model (in the build
container) and this repo's modules (anywhere) get bit-identical parameters from one seed,
independent of the order in which the two implementations register their parameters.
Scales keep activations O(1) so that the 1e-3 logit tolerance is a real test:
  * matrices / conv kernels : N(0, 1/fan_in)
  * embeddings              : N(0, 1)
  * relative-position bias  : N(0, 0.5^2)
  * norm weights            : 1 + 0.1 N(0,1);  biases 0.1 N(0,1)
  * BN running_mean         : 0.1 N(0,1);  running_var : 1 + |0.1 N(0,1)|
Integer buffers (relative_position_index, num_batches_tracked) and the fixed
`mask` / `position_enc.pe` buffers are left untouched.
"""
import zlib

import numpy as np
import torch

_SKIP_SUFFIX = ("relative_position_index", "num_batches_tracked")
_SKIP_EXACT = ("mask",)


def _is_norm_weight(key):
    if not key.endswith(".weight"):
        return False
    stem = key[: -len(".weight")]
    last = stem.split(".")[-1]
    if last in ("layernorm_before", "layernorm_after", "norm", "norm1", "norm2"):
        return True
    # BatchNorm inside nn.Sequential(conv, bn, act): "...convN.1.weight"
    parts = stem.split(".")
    return len(parts) >= 2 and parts[-1] == "1" and parts[-2] in ("conv1", "conv2", "conv3")


def _is_embedding(key):
    last = key.split(".")[-2] if key.endswith(".weight") else ""
    return last in ("occ_enc", "level_enc", "octant_enc")


@torch.no_grad()
def fill_weights(module, seed=0):
    """Overwrite the floating-point entries of a module's state_dict in place."""
    fill_state_dict(module.state_dict(), seed)
    return module


@torch.no_grad()
def fill_state_dict(sd, seed=0):
    """Same, for a plain {key: tensor} mapping (tensors are modified in place)."""
    for key in sd.keys():
        t = sd[key]
        rng = np.random.default_rng([int(seed), zlib.crc32(key.encode())])
        if key in _SKIP_EXACT or key.endswith(_SKIP_SUFFIX) or key.endswith("position_enc.pe"):
            continue
        if not torch.is_floating_point(t):
            continue
        shape = tuple(t.shape)
        g = rng.standard_normal(shape, dtype=np.float32)
        if key.endswith("running_var"):
            v = 1.0 + np.abs(0.1 * g)
        elif key.endswith("running_mean"):
            v = 0.1 * g
        elif key.endswith("relative_position_bias_table"):
            v = 0.5 * g
        elif _is_norm_weight(key):
            v = 1.0 + 0.1 * g
        elif _is_embedding(key):
            v = g
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            v = g / np.sqrt(np.float32(fan_in))
        else:
            v = 0.1 * g
        t.copy_(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(t.device))
    return sd
