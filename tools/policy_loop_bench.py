#!/usr/bin/env python3
"""Closed-loop roll-outs with a device-resident policy IN the loop (GPU box): every env step's actions are computed from the previous
step's observations, as in any RL roll-out.  Compares, for the BASELINE workloads,

    joined   : a = policy(obs); obs = env.step_tensors(a)          - one stream, the SB3-shaped loop
    chains   : ChainStepper.closed_loop(policy, K)                  - one policy -> step loop per lane range, on its own stream
    joined-g : the joined loop captured into a HIP graph of 8 steps (ChainStepper.capture_closed_loop(joined=True)), replayed
    chains-g : the per-chain loops captured into ONE HIP graph with a branch per chain, replayed - no host work in the loop

with two policies: "elementwise" (4 small torch kernels on the observation) and "mlp" (obs -> 64 tanh -> act, torch matmuls).
Results are identical between the two loops (tests/test_gpu_chains.py); this prints microseconds per env step of the whole batch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marinevehiclereinforcementlearning_amd.chains import ChainStepper  # noqa: E402
from marinevehiclereinforcementlearning_amd.flow import ReconstructedFlow  # noqa: E402
from marinevehiclereinforcementlearning_amd.vec_env import MarineVecEnv  # noqa: E402


def main():
    K = 1504
    for name, model, n, use_flow in (("c4", "rov6", 1048576, True), ("c3", "rov6", 262144, False), ("auv", "auv", 1048576, True)):
        flow = None
        if use_flow:
            flow = ReconstructedFlow.synthetic(n_modes=8, n_time=2000, device=0)
            flow.scale(11., 1., 2., translate=(-1.65, -1.1))
        env = MarineVecEnv(model, n, seed=12345, flow=flow, infos="lean")
        ad, od = env.action_space.shape[0], env.observation_space.shape[0]
        g = torch.Generator(device="cuda").manual_seed(1)
        w = torch.rand((1, ad), device="cuda", generator=g) * 4 - 2
        w1 = (torch.rand((od, 64), device="cuda", generator=g) - 0.5)
        w2 = (torch.rand((64, ad), device="cuda", generator=g) - 0.5) * 0.3
        policies = {"elementwise": lambda o: torch.tanh(o[:, :ad] * w + o[:, od - ad:] * 0.5),
                    "mlp 64": lambda o: torch.tanh(torch.tanh(o @ w1) @ w2)}
        for pname, policy in policies.items():
            res = {}
            G = 8
            for plan in ("joined", "chains", "joined-g", "chains-g", "joined", "chains", "joined-g", "chains-g"):
                obs = env.reset_tensors()
                st = ChainStepper(env, n_chains=2)
                graph = None
                if plan.endswith("-g"):
                    graph, _ = st.capture_closed_loop(policy, G, joined=plan == "joined-g")

                def run(k):
                    nonlocal obs
                    if graph is not None:
                        for _ in range(k // G):
                            graph.replay()
                    elif plan == "joined":
                        for _ in range(k):
                            obs, _, _ = env.step_tensors(policy(obs).contiguous())
                    else:
                        st.closed_loop(policy, k)
                run(300)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(K)
                torch.cuda.synchronize()
                res.setdefault(plan, []).append((time.perf_counter() - t0) / K * 1e6)
            # the policy alone, for scale
            obs = env.reset_tensors()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(K):
                a = policy(obs)
            torch.cuda.synchronize()
            tp = (time.perf_counter() - t0) / K * 1e6
            j, c, jg, cg = min(res["joined"]), min(res["chains"]), min(res["joined-g"]), min(res["chains-g"])
            print(f"{name:4s} {n:8d} envs, policy {pname:12s} ({tp:5.1f} us alone): eager joined {j:7.1f} / per-chain {c:7.1f} us/step | "
                  f"HIP graph of {G} steps: joined {jg:7.1f} / per-chain branches {cg:7.1f} us/step ({100 * (1 - cg / jg):.0f} % less than the joined graph, "
                  f"{100 * (1 - cg / j):.0f} % less than the eager joined loop) = {n / min(c, cg, j, jg) * 1e6:.3e} env-steps/s closed loop", flush=True)
        env.close()


if __name__ == "__main__":
    main()
