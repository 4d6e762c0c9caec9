"""Deterministic, framework-independent parameter filling.

Golden fixtures store only a *seed* for the network weights; both the fixture
generator (which fills the reference's modules) and the tests (which fill the
oracle and the HIP host modules) regenerate the identical arrays from numpy's
PCG64 stream, so the fixtures stay small.  Parameter names/shapes are the
reference's, so one dict serves every consumer.
"""
import zlib
import numpy as np


def _kind(name, shape):
    last = name.rsplit(".", 1)[-1]
    if "running_var" in name:
        return "var"
    if "running_mean" in name:
        return "norm_b"
    if "norm" in name or ("input_proj" in name and (name.endswith(".1.weight") or name.endswith(".1.bias"))):
        # LayerNorm / GroupNorm / FrozenBN affine
        return "norm_w" if last == "weight" else "norm_b"
    if "running_var" in name:
        return "var"
    if "running_mean" in name:
        return "norm_b"
    if "sampling_offsets.bias" in name:
        return "offs_b"
    if last in ("bias", "in_proj_bias"):
        return "bias"
    if "sampling_offsets.weight" in name:
        return "small_w"
    if "embed" in name and len(shape) == 2 and "mask_embed" not in name and "class_embed" not in name:
        return "embed"
    if "query_feat" in name:
        return "embed"
    return "weight"


def seeded_array(name, shape, seed):
    """One array, a function of (name, shape, seed) only."""
    h = zlib.crc32(name.encode()) & 0xFFFFFFFF
    rng = np.random.Generator(np.random.PCG64([seed, h]))
    shape = tuple(int(s) for s in shape)
    x = rng.standard_normal(shape)
    kind = _kind(name, shape)
    if kind == "weight":
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        x *= 1.0 / np.sqrt(max(fan_in, 1))
    elif kind == "small_w":
        x *= 0.3 / np.sqrt(shape[1])
    elif kind == "offs_b":
        x = rng.uniform(-2.5, 2.5, shape)
    elif kind == "bias":
        x *= 0.05
    elif kind == "norm_w":
        x = 1.0 + 0.1 * x
    elif kind == "norm_b":
        x *= 0.1
    elif kind == "var":
        x = 0.5 + rng.random(shape)
    elif kind == "embed":
        x *= 1.0
    return x.astype(np.float32)


def seeded_state(named_shapes, seed, keep=()):
    """named_shapes: iterable of (name, shape). Names in `keep` are skipped
    (left at the module's own init, e.g. sampling_offsets.bias compass grid)."""
    out = {}
    for name, shape in named_shapes:
        if any(k in name for k in keep):
            continue
        out[name] = seeded_array(name, shape, seed)
    return out
