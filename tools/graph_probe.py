#!/usr/bin/env python3
"""Does replaying the step launches from a HIP graph shorten the gap between consecutive launches?  (experiment)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marinevehiclereinforcementlearning_amd.flow import ReconstructedFlow
from marinevehiclereinforcementlearning_amd.vec_env import MarineVecEnv

for model, n, use_flow in (("rov6", 1048576, True), ("rov3", 65536, False), ("rov6", 16384, False)):
    flow = None
    if use_flow:
        flow = ReconstructedFlow.synthetic(n_modes=8, n_time=2000, device=0)
        flow.scale(11., 1., 2., translate=(-1.65, -1.1))
    env = MarineVecEnv(model, n, seed=1, flow=flow, infos="lean")
    env.reset_tensors()
    R = 8
    act = torch.rand((R, n, env.action_space.shape[0]), device="cuda") * 2 - 1
    K = 4000

    def plain(k):
        for i in range(k):
            env.step_tensors(act[i % R])
    plain(500); torch.cuda.synchronize()
    t0 = time.perf_counter(); plain(K); torch.cuda.synchronize(); t_plain = (time.perf_counter() - t0) / K

    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        plain(R)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g, stream=s):
        plain(R)
    for _ in range(60):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K // R):
        g.replay()
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / K
    # a graph of ONE step, replayed per step: what a per-call graph inside mvrl_step_dev could buy VecEnv.step
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1, stream=s):
        plain(1)
    for _ in range(200):
        g1.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        g1.replay()
    torch.cuda.synchronize()
    t_g1 = (time.perf_counter() - t0) / K
    print(f"{model} n={n}: plain {t_plain*1e6:.2f} us/step, graph of {R} steps {t_graph*1e6:.2f} us/step ({t_plain/t_graph:.3f}x), "
          f"graph of 1 step replayed per step {t_g1*1e6:.2f} us/step")
