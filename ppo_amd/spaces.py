"""Minimal observation/action space descriptors.

The reference's Agent is duck-typed on `.shape`, `.n`, `.low`, `.high` and on
isinstance(..., gym.spaces.Discrete / Box) (reference src/ppo.py:116-120,184-191,258).
gymnasium is optional here: these classes carry the same attributes, and
`is_discrete` / `is_box` also accept real gymnasium spaces when that package exists.
"""
from __future__ import annotations

import numpy as np


class Space:
    def __init__(self, shape=(), dtype=np.float32):
        self._shape = tuple(shape)
        self.dtype = dtype

    @property
    def shape(self):
        return self._shape


class Discrete(Space):
    def __init__(self, n: int):
        super().__init__((), np.int64)
        self.n = int(n)

    def sample(self):
        return int(np.random.randint(self.n))


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.shape(low)
        super().__init__(shape, dtype)
        self.low = low if np.isscalar(low) else np.asarray(low, dtype=np.float32)
        self.high = high if np.isscalar(high) else np.asarray(high, dtype=np.float32)

    def sample(self):
        lo = np.broadcast_to(np.asarray(self.low, dtype=np.float64), self.shape)
        hi = np.broadcast_to(np.asarray(self.high, dtype=np.float64), self.shape)
        lo = np.where(np.isfinite(lo), lo, -1.0)
        hi = np.where(np.isfinite(hi), hi, 1.0)
        return np.random.uniform(lo, hi).astype(self.dtype)


def is_discrete(space) -> bool:
    return isinstance(space, Discrete) or (hasattr(space, "n") and not hasattr(space, "low"))


def is_box(space) -> bool:
    return isinstance(space, Box) or (hasattr(space, "low") and hasattr(space, "high"))
