#!/usr/bin/env python3
"""Does a high-priority HIP stream shorten the dependent-launch gap?  One launch per step of C2 (65 536 envs: the gap is 3.3 of 11.7 us)
and C4 on torch's default stream, a normal side stream and a high-priority side stream.  python tools/prio_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marinevehiclereinforcementlearning_amd.vec_env import MarineVecEnv
from marinevehiclereinforcementlearning_amd.flow import ReconstructedFlow

def run(model, n, flow, steps):
    env = MarineVecEnv(model, n, seed=1, flow=flow, device=0, infos="lean")
    act = torch.rand((8, n, env.action_space.shape[0]), device="cuda") * 2 - 1
    env.reset_tensors()
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    streams = {"default": torch.cuda.current_stream(), "side": torch.cuda.Stream(), "side, high priority": torch.cuda.Stream(priority=-1)}
    for rnd in range(2):
        for name, s in streams.items():
            with torch.cuda.stream(s):
                for k in range(200):
                    env.step_tensors(act[k % 8])
                s.synchronize()
                t0 = time.perf_counter()
                for k in range(steps):
                    env.step_tensors(act[k % 8])
                s.synchronize()
                dt = (time.perf_counter() - t0) / steps * 1e6
            print(f"{model} n={n} {name:22s} {dt:8.2f} us per step", flush=True)

flow = ReconstructedFlow.synthetic(n_modes=8, n_time=2000, device=0)
flow.scale(11., 1., 2., translate=(-1.65, -1.1))
run("rov3", 65536, None, 6000)
run("rov6", 1048576, flow, 1500)
