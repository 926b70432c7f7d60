"""Deterministic, name-keyed parameter fill shared by make_golden.py (reference side) and the tests
(oracle / HIP side), so fixtures only need to hold inputs and expected outputs, not weights.

Values come from numpy's PCG64 bit generator seeded by crc32(param name): full-rank, O(1)
activations, identical wherever numpy is.
"""
import zlib

import numpy as np
import torch


def tensor_for(name: str, shape, salt: int = 0) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    rng = np.random.Generator(np.random.PCG64((zlib.crc32(name.encode()) + 7919 * salt) & 0xFFFFFFFF))
    u = (rng.random(n) - 0.5) * np.sqrt(12.0)  # unit variance
    leaf = name.rsplit(".", 1)[-1]
    if len(shape) == 0:
        v = np.array(2.0 + 0.1 * u[0])  # logit scales
    elif len(shape) == 1:
        if leaf in ("weight", "gamma") or name.endswith("LayerNorm.weight"):
            v = 1.0 + 0.1 * u  # norm scales
        elif leaf == "class_embedding":
            v = 0.5 * u
        else:
            v = 0.1 * u  # biases
    else:
        fan_in = int(np.prod(shape[1:]))
        if "embedding" in name or "embed" in name and len(shape) == 2:
            v = 0.5 * u
        elif leaf in ("proj", "text_projection", "img_proj"):
            v = u * (shape[0] ** -0.5)
        else:
            v = u * (fan_in ** -0.5)
    return torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shape).copy())


def fill_module_(module: torch.nn.Module, salt: int = 0, skip=()):
    """In-place fill of every parameter / float buffer of `module`, keyed by its canonical (first
    registered, de-duplicated) name from named_parameters()/named_buffers() -- shared tensors that
    appear under several state_dict names are filled once, under the first name."""
    with torch.no_grad():
        for name, t in list(module.named_parameters()) + list(module.named_buffers()):
            if not t.is_floating_point() or any(s in name for s in skip):
                continue
            t.copy_(tensor_for(name, t.shape, salt).to(t.dtype))
    return module


def fill_dict(shapes: dict, salt: int = 0) -> dict:
    return {k: tensor_for(k, s, salt) for k, s in shapes.items()}


def data_tensor(tag: str, shape, scale: float = 1.0) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    rng = np.random.Generator(np.random.PCG64(zlib.crc32(("data:" + tag).encode())))
    u = (rng.random(int(np.prod(shape))) - 0.5) * np.sqrt(12.0) * scale
    return torch.from_numpy(u.astype(np.float32).reshape(shape).copy())


def data_ints(tag: str, shape, low: int, high: int) -> torch.Tensor:
    rng = np.random.Generator(np.random.PCG64(zlib.crc32(("ints:" + tag).encode())))
    return torch.from_numpy(rng.integers(low, high, size=tuple(shape)).astype(np.int64))
