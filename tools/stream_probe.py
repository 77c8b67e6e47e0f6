#!/usr/bin/env python3
"""Does the one-launch-per-step plan pay for running on the legacy NULL stream?  The same K dependent step launches timed on torch's default
(null) stream, on a torch side stream, and on a non-blocking HIP stream.   python tools/stream_probe.py [workload-envs] [K]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from marinevehiclereinforcementlearning_amd.flow import ReconstructedFlow  # noqa: E402
from marinevehiclereinforcementlearning_amd.vec_env import MarineVecEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
flow = ReconstructedFlow.synthetic(n_modes=8, n_time=2000, device=0)
flow.scale(11., 1., 2., translate=(-1.65, -1.1))
env = MarineVecEnv("rov6", n, seed=12345, flow=flow, infos="lean")
ring = torch.empty((8, n, 6), device="cuda")
for r in range(8):
    env.handle.fill_uniform_dev(ring[r].data_ptr(), n * 6, 12345, r, -1.0, 1.0, torch.cuda.current_stream().cuda_stream)
env.reset_tensors()


def run(label, stream_ctx):
    with stream_ctx:
        for k in range(300):
            env.step_tensors(ring[k & 7])
        torch.cuda.synchronize()
        res = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for k in range(K):
                env.step_tensors(ring[k & 7])
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1e3 / K)
    print(f"{label:34s} us/step: " + " ".join(f"{x:.1f}" for x in res) + f"   median {sorted(res)[2]:.1f}", flush=True)


import contextlib
run("torch default (null) stream", contextlib.nullcontext())
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
run("torch side stream", torch.cuda.stream(side))
torch.cuda.synchronize()
run("default stream again", contextlib.nullcontext())
