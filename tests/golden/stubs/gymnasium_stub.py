"""Minimal stand-in for the `gymnasium` package -- FIXTURE GENERATION ONLY (build container, tests/golden/make_golden.py).

`gymnasium` is a third-party dependency of the reference that is not installed here.  The reference's tomato_env.py /
base_env.py / observations.py only use: gym.Env (reset(seed) seeding self._np_random), gymnasium.spaces.Box (low / high /
shape / dtype) and gymnasium.utils.seeding.np_random.  This stub provides exactly that, with gymnasium's published
semantics: seeding.np_random(seed) = numpy Generator(PCG64(SeedSequence(seed))) (gymnasium/utils/seeding.py), and
Env.reset(seed=...) re-seeds self._np_random only when a seed is given (gymnasium/core.py).
It is injected into sys.modules before the reference modules are imported; nothing on the GPU box ever loads it.
"""
from __future__ import annotations

import sys
import types

import numpy as np


def _np_random(seed=None):
    ss = np.random.SeedSequence(seed)
    return np.random.Generator(np.random.PCG64(ss)), ss.entropy


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.shape(low)
        self.shape = tuple(shape)
        self.low = np.full(self.shape, low, dtype=self.dtype) if np.isscalar(low) else np.asarray(low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype) if np.isscalar(high) else np.asarray(high, dtype=self.dtype)
        self._rng, _ = _np_random(seed)

    def seed(self, seed=None):
        self._rng, s = _np_random(seed)
        return [s]

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


class Dict(dict):
    pass


class Env:
    metadata: dict = {}
    _np_random = None

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self._np_random, _ = _np_random(seed)

    @property
    def np_random(self):
        if self._np_random is None:
            self._np_random, _ = _np_random()
        return self._np_random

    @property
    def unwrapped(self):
        return self

    def close(self):
        pass


def install():
    """Register the stub as `gymnasium`, `gymnasium.spaces`, `gymnasium.utils`, `gymnasium.utils.seeding`."""
    gym = types.ModuleType("gymnasium")
    spaces = types.ModuleType("gymnasium.spaces")
    utils = types.ModuleType("gymnasium.utils")
    seeding = types.ModuleType("gymnasium.utils.seeding")
    spaces.Box, spaces.Dict = Box, Dict
    seeding.np_random = _np_random
    utils.seeding = seeding
    gym.Env, gym.spaces, gym.utils = Env, spaces, utils
    sys.modules.update({"gymnasium": gym, "gymnasium.spaces": spaces, "gymnasium.utils": utils,
                        "gymnasium.utils.seeding": seeding})
    return gym
