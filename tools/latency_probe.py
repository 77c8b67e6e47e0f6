#!/usr/bin/env python3
"""Small-batch step latency of the drop-in paths (what an SB3 user with nProc = 16..64 envs sees):
raw C-ABI mvrl_step, MarineVecEnv.step (SB3 semantics, infos list), and the single-env Gym facade."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marinevehiclereinforcementlearning_amd import _lib, params as P
from marinevehiclereinforcementlearning_amd.vec_env import MarineVecEnv
from marinevehiclereinforcementlearning_amd.envs import BlueROV2Heavy6DoFEnv


def timeit(fn, k=2000):
    for _ in range(50):
        fn()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    return (time.perf_counter() - t0) / k


for n in (1, 16, 64, 1024, 16384):
    h = _lib.Handle(P.make_config("rov6", n, seed=1, use_flow=False))
    h.reset()
    a = np.random.default_rng(0).uniform(-1, 1, size=(n, 6)).astype(np.float32)
    t_abi = timeit(lambda: h.step(a, copy=False))
    h.close()
    env = MarineVecEnv("rov6", n, seed=1)
    env.reset()
    t_vec = timeit(lambda: env.step(a))
    env.close()
    print(f"n={n:6d}: mvrl_step {t_abi*1e6:7.1f} us  ({n/t_abi:.3e} env-steps/s)   MarineVecEnv.step {t_vec*1e6:7.1f} us ({n/t_vec:.3e})")
e = BlueROV2Heavy6DoFEnv()
e.reset()
a1 = np.zeros(6, np.float32)
def one():
    if e.step(a1)[2]:
        e.reset()
t1 = timeit(one, 1000)
print(f"BlueROV2Heavy6DoFEnv.step (single-env Gym facade): {t1*1e6:.1f} us/step -> {1/t1:.0f} steps/s (reference: ~20 steps/s)")
