"""
tests/fake_model.py -- deterministic, platform-independent stand-ins used by the golden generator
and by the parity tests (test infrastructure).

* FakeNet: logits/value are a pure INTEGER-hash function of the input planes, so that they are
  bit-identical on any host, for any batch size and any row order (BLAS-free).
* hash_init_: fills a PolicyValueNet's parameters and BN buffers from an integer hash of
  (state_dict key, flat index), so that "random-init" weights need no torch RNG and no files.
"""
from __future__ import annotations

import hashlib
import zlib

import numpy as np

NUM_ACTIONS = 4672
PLANES = 120 * 64

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


_W = None


def planes_hash(planes: np.ndarray) -> np.ndarray:
    """uint64 hash per row of planes[B, 120, 8, 8] (entries are small non-negative integers)."""
    global _W
    if _W is None:
        _W = _splitmix64(np.arange(PLANES, dtype=np.uint64))
    x = np.ascontiguousarray(planes, dtype=np.float32).reshape(-1, PLANES)
    xi = x.astype(np.int64).astype(np.uint64)
    with np.errstate(over="ignore"):
        return (xi * _W[None, :]).sum(axis=1, dtype=np.uint64)


def fake_logits_values(planes: np.ndarray, scale: float = 6.0, salt: int = 0):
    """planes[B,120,8,8] -> (logits f32[B,4672], values f32[B])."""
    h = planes_hash(planes) ^ np.uint64(salt)
    a = np.arange(NUM_ACTIONS, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = _splitmix64(h[:, None] + a[None, :] * np.uint64(0xD1B54A32D192ED03))
    u = (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    logits = ((u - 0.5) * scale).astype(np.float32)
    zv = _splitmix64(h ^ np.uint64(0xABCDEF0123456789))
    v = ((zv >> np.uint64(40)).astype(np.float64) / float(1 << 24) * 2.0 - 1.0) * 0.9
    return logits, v.astype(np.float32)


def planes_key(planes_row: np.ndarray) -> str:
    """stable text key of one encoded position (sha1 of the float32 bytes)."""
    return hashlib.sha1(np.ascontiguousarray(planes_row, dtype=np.float32).tobytes()).hexdigest()


class FakeNet:
    """torch-module-like callable: model(x) -> (logits[B,4672], value[B,1]) as torch tensors."""

    def __init__(self, scale: float = 6.0, salt: int = 0):
        self.scale, self.salt = scale, salt
        self.calls = []

    def __call__(self, x):
        import torch

        planes = x.detach().cpu().numpy()
        logits, v = fake_logits_values(planes, self.scale, self.salt)
        self.calls.append(planes.shape[0])
        return torch.from_numpy(logits), torch.from_numpy(v).unsqueeze(1)

    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self


def hash_init_(model, gain: float = 0.5):
    """Deterministically fill every parameter/buffer of a torch module from an integer hash."""
    import torch

    sd = model.state_dict()
    new = {}
    for name, t in sd.items():
        n = t.numel()
        if name.endswith("num_batches_tracked"):
            new[name] = torch.zeros_like(t)
            continue
        seed = np.uint64(zlib.crc32(name.encode()))
        with np.errstate(over="ignore"):
            z = _splitmix64(seed * np.uint64(0x100000001B3) + np.arange(n, dtype=np.uint64))
        u = (z >> np.uint64(40)).astype(np.float64) / float(1 << 24) * 2.0 - 1.0  # [-1, 1)
        if name.endswith("running_var"):
            vals = 1.0 + 0.5 * np.abs(u)
        elif name.endswith("running_mean"):
            vals = 0.1 * u
        elif "bn" in name.split(".")[-2] and name.endswith("weight"):
            vals = 1.0 + 0.1 * u
        elif name.endswith("bias"):
            vals = 0.1 * u
        else:
            fan_in = int(np.prod(t.shape[1:])) if t.dim() > 1 else int(t.shape[0])
            vals = u * gain * np.sqrt(3.0 / max(1, fan_in))
        new[name] = torch.from_numpy(vals.astype(np.float32)).reshape(t.shape).to(t.dtype)
    model.load_state_dict(new)
    return model
