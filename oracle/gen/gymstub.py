"""Minimal stand-in for the `gym` package so that the *reference* can be imported
in the build container (gym/gymnasium/SB3 are not installed and there is no network).

TEST INFRASTRUCTURE ONLY.  Used exclusively by oracle/gen/gen_golden_*.py, which run in
the build container where /root/reference exists.  Nothing here is shipped in the
product path and nothing here runs on the GPU box.

It provides exactly the names the reference touches at import / construction time:
gym.Env, gym.spaces.Box, gym.utils.seeding.np_random.
"""
import sys
import types

import numpy as np


def install():
    if "gym" in sys.modules:
        return
    gym = types.ModuleType("gym")
    spaces = types.ModuleType("gym.spaces")
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")

    class Env(object):
        def __init__(self, *a, **k):
            pass

    class Box(object):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=dtype), shape).copy()
            self.shape = tuple(shape)
            self.dtype = np.dtype(dtype)

    def np_random(seed=None):
        return np.random.RandomState(seed), seed

    gym.Env = Env
    gym.spaces = spaces
    gym.utils = utils
    spaces.Box = Box
    utils.seeding = seeding
    seeding.np_random = np_random
    sys.modules["gym"] = gym
    sys.modules["gym.spaces"] = spaces
    sys.modules["gym.utils"] = utils
    sys.modules["gym.utils.seeding"] = seeding
