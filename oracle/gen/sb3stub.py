"""Stand-ins for `stable_baselines3`, `sb3_contrib` and `gymnasium` so that the reference's
tag_00_Dec2023_simpleControlTurbulence/main_02_sbl_contrib_customBuffer.py can be IMPORTED in the build container (none of
the three is installed and there is no network) and its `CustomReplayBuffer.add` - the code under test, run unmodified -
can be executed to write fixtures.

TEST INFRASTRUCTURE ONLY.  Used exclusively by oracle/gen/gen_golden_replay.py; nothing here ships in the product path or
runs on the GPU box.

What is stubbed is only what the module touches at import time plus the base-class constructor: `ReplayBuffer.__init__`
allocates the arrays the way SB3 1.8 documents them (observations / next_observations [buffer_size, n_envs, *obs_shape] in
the observation space's dtype, actions [buffer_size, n_envs, action_dim], rewards / dones / timeouts [buffer_size, n_envs]
float32, with buffer_size = max(buffer_size // n_envs, 1), pos = 0, full = False).  Every line of `add` that runs -
the five sign masks, the `nRollovers > 2` rule, the slot and roll-over bookkeeping, the timeouts read from `infos` - is
the reference's own.
"""
import sys
import types

import numpy as np


def _mod(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def install():
    if "stable_baselines3" in sys.modules:
        return
    # ---- gymnasium.spaces ------------------------------------------------------------------------------------------
    gymn = _mod("gymnasium")
    spaces = _mod("gymnasium.spaces")

    class Space(object):
        pass

    class Discrete(Space):
        def __init__(self, n):
            self.n, self.shape, self.dtype = int(n), (), np.dtype(np.int64)

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=dtype), shape).copy()
            self.shape, self.dtype = tuple(shape), np.dtype(dtype)

    spaces.Space, spaces.Discrete, spaces.Box = Space, Discrete, Box
    gymn.spaces = spaces

    # ---- stable_baselines3 ------------------------------------------------------------------------------------------
    sb3 = _mod("stable_baselines3")
    common = _mod("stable_baselines3.common")
    vec_env = _mod("stable_baselines3.common.vec_env")
    noise = _mod("stable_baselines3.common.noise")
    buffers = _mod("stable_baselines3.common.buffers")
    prep = _mod("stable_baselines3.common.preprocessing")
    aliases = _mod("stable_baselines3.common.type_aliases")
    utils = _mod("stable_baselines3.common.utils")
    sb3.common = common
    common.vec_env, common.noise, common.buffers = vec_env, noise, buffers
    common.preprocessing, common.type_aliases, common.utils = prep, aliases, utils
    for name in ("VecMonitor", "SubprocVecEnv", "VecNormalize"):
        setattr(vec_env, name, type(name, (object,), {}))
    for name in ("NormalActionNoise", "VectorizedActionNoise"):
        setattr(noise, name, type(name, (object,), {}))
    for name in ("DictReplayBufferSamples", "DictRolloutBufferSamples", "ReplayBufferSamples", "RolloutBufferSamples"):
        setattr(aliases, name, type(name, (object,), {}))
    utils.get_device = lambda device="auto": "cpu"

    def get_obs_shape(space):
        return tuple(space.shape)

    def get_action_dim(space):
        return int(np.prod(space.shape))

    prep.get_obs_shape, prep.get_action_dim = get_obs_shape, get_action_dim

    class BaseBuffer(object):
        def __init__(self, buffer_size, observation_space, action_space, device="auto", n_envs=1):
            self.buffer_size = buffer_size
            self.observation_space, self.action_space = observation_space, action_space
            self.obs_shape = get_obs_shape(observation_space)
            self.action_dim = get_action_dim(action_space)
            self.pos, self.full, self.device, self.n_envs = 0, False, "cpu", n_envs

    class ReplayBuffer(BaseBuffer):
        def __init__(self, buffer_size, observation_space, action_space, device="auto", n_envs=1, optimize_memory_usage=False,
                     handle_timeout_termination=True):
            super().__init__(buffer_size, observation_space, action_space, device, n_envs=n_envs)
            self.buffer_size = max(buffer_size // n_envs, 1)
            self.optimize_memory_usage = optimize_memory_usage
            self.observations = np.zeros((self.buffer_size, self.n_envs, *self.obs_shape), dtype=observation_space.dtype)
            self.next_observations = np.zeros((self.buffer_size, self.n_envs, *self.obs_shape), dtype=observation_space.dtype)
            self.actions = np.zeros((self.buffer_size, self.n_envs, self.action_dim), dtype=action_space.dtype)
            self.rewards = np.zeros((self.buffer_size, self.n_envs), dtype=np.float32)
            self.dones = np.zeros((self.buffer_size, self.n_envs), dtype=np.float32)
            self.handle_timeout_termination = handle_timeout_termination
            self.timeouts = np.zeros((self.buffer_size, self.n_envs), dtype=np.float32)

    buffers.BaseBuffer, buffers.ReplayBuffer = BaseBuffer, ReplayBuffer
    _mod("sb3_contrib")
