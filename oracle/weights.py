"""Portable-formula tensors: deterministic, platform-independent weights and inputs for parity tests.

TEST INFRASTRUCTURE (oracle/): imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.

Every value is a pure function of (tensor name, element index): a 32-bit integer hash mapped to a
uniform in [-1, 1) and scaled by a per-tensor rule, then rounded to a bf16-representable fp32.  The same
numbers can therefore be regenerated in the build container (to fill the imported reference model when
minting tests/golden/*), on the GPU box (to fill the oracle and the HIP model) and in C++, so the 150 M
weights of an end-to-end fixture never have to be stored.
"""
import zlib

import numpy as np
import torch


def _mix(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(13); x = (x * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x


def uniform_pm1(name, numel, salt=0):
    """[-1, 1) uniform from hash(crc32(name) + salt, index); float64 array of length numel."""
    seed = np.uint64((zlib.crc32(name.encode()) + 0x9E3779B9 * (salt + 1)) & 0xFFFFFFFF)
    idx = np.arange(numel, dtype=np.uint64)
    h = _mix((idx * np.uint64(0x9E3779B1) & np.uint64(0xFFFFFFFF)) ^ seed)
    return (h >> np.uint64(8)).astype(np.float64) / float(1 << 23) - 1.0


def to_bf16_grid(t):
    return t.to(torch.bfloat16).to(torch.float32)


def formula_tensor(name, shape, scale=1.0, offset=0.0, bf16=True, salt=0):
    numel = int(np.prod(shape)) if len(shape) else 1
    v = uniform_pm1(name, numel, salt) * scale + offset
    t = torch.from_numpy(v.astype(np.float32)).reshape(tuple(shape))
    return to_bf16_grid(t) if bf16 else t


def _rule(key, shape):
    """(scale, offset) for a state_dict entry, by its reference key name (SURVEY.md §8b key contract)."""
    leaf = key.rsplit(".", 1)[-1]
    if "running_var" in key:
        return 0.4, 1.0                      # var in [0.6, 1.4]
    if "running_mean" in key:
        return 0.2, 0.0
    is_bn = (".bn" in key or "downsample.1" in key) and "img_backbone" in key
    if is_bn and leaf == "weight":
        return (0.1, 0.35) if ".bn3." in key else (0.25, 1.0)   # damp the residual branch
    if is_bn and leaf == "bias":
        return 0.1, 0.0
    if "LayerNorm" in key or ".norm" in key or key.endswith("norm.weight") or key.endswith("norm.bias") \
            or (".gn" in key) or _is_norm_in_seq(key, shape):
        return (0.2, 1.0) if leaf == "weight" else (0.1, 0.0)
    if leaf == "bias" or leaf == "in_proj_bias":
        return 0.05, 0.0
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        if "embed" in key and "bbox_embed" not in key:   # embedding tables: [num, dim]
            return 0.5, 0.0
        relu_fed = "img_backbone" in key
        gain = 6.0 if relu_fed else 3.0
        return (gain / fan_in) ** 0.5, 0.0
    return 0.5, 0.0


def _is_norm_in_seq(key, shape):
    # mlp_mapping / context_out / input_proj hold LayerNorm / GroupNorm at Sequential indices 1 and 5
    if len(shape) != 1:
        return False
    parts = key.split(".")
    return len(parts) >= 2 and parts[-2] in ("1", "5") and (
        "map_sentence" in key or "map_phrase" in key or "fuse_encoder_query" in key
        or "context_out" in key or "input_proj" in key)


def fill_state_dict(sd, skip=()):
    """In-place: overwrite every floating tensor of a state_dict with its formula value."""
    with torch.no_grad():
        for k, v in sd.items():
            if not torch.is_floating_point(v) or any(s in k for s in skip):
                continue
            scale, offset = _rule(k, tuple(v.shape))
            v.copy_(formula_tensor(k, tuple(v.shape), scale, offset).to(v.dtype))
    return sd


def formula_state(shapes):
    """dict name -> tensor for a {name: shape} table (used to build oracle / HIP params from scratch)."""
    out = {}
    for k, shp in shapes.items():
        scale, offset = _rule(k, tuple(shp))
        out[k] = formula_tensor(k, tuple(shp), scale, offset)
    return out
