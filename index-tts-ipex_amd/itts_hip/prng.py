"""Builder-owned counter-based PRNG for synthetic weights and inputs.

No checkpoint exists offline, so parity fixtures, tests and the benchmark all
run on config-shaped synthetic tensors.  They must regenerate *bit-identically* on the GPU box, in
this container and inside the fixture generator, independent of torch/numpy generator versions, so
the stream is defined here in integer arithmetic only:

    value(name, seed, i) = uniform(-1, 1) from splitmix64(fnv1a64(name) ^ seed*0x9E3779B97F4A7C15 + i)

(top 24 bits -> exactly representable float32), then scaled.  No transcendental functions are used,
so results are identical on every platform.
"""
from __future__ import annotations

import numpy as np

_M64 = (1 << 64) - 1


def fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & _M64
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(name: str, seed: int, n: int, chunk: int = 1 << 24) -> np.ndarray:
    """n float32 values in (-1, 1), a pure function of (name, seed, index)."""
    base = (fnv1a64(name) ^ ((seed * 0x9E3779B97F4A7C15) & _M64)) & _M64
    out = np.empty(n, dtype=np.float32)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        with np.errstate(over="ignore"):
            idx = np.arange(s, e, dtype=np.uint64) + np.uint64(base)
        z = _splitmix64(idx)
        u24 = (z >> np.uint64(40)).astype(np.int64)  # 24 bits
        out[s:e] = ((u24 * 2 + 1).astype(np.float32) * np.float32(1.0 / (1 << 24))) - np.float32(1.0)
    return out


def tensor(name: str, seed: int, shape, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    """Uniform tensor with the requested standard deviation (uniform(-a,a) has std a/sqrt(3))."""
    n = int(np.prod(shape)) if len(shape) else 1
    a = np.float32(std * 1.7320508075688772)
    x = uniform(name, seed, n) * a + np.float32(mean)
    return x.reshape(shape).astype(np.float32)


def randint(name: str, seed: int, n: int, lo: int, hi: int) -> np.ndarray:
    """n integers in [lo, hi) (int64)."""
    base = (fnv1a64(name) ^ ((seed * 0x9E3779B97F4A7C15) & _M64)) & _M64
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + np.uint64(base)
    z = _splitmix64(idx)
    return (lo + (z >> np.uint64(11)).astype(np.int64) % (hi - lo)).astype(np.int64)
