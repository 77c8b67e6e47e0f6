#!/usr/bin/env python3
"""Calibration of tests/parity_util.F32_BOUNDS on the GPU: for the six parametrisations of
tests/test_gpu_parity.py::test_random_batch_vs_fp64_oracle (4096 envs x 25 steps) print every env beyond 1e-5 with its
smallest distance to each discontinuity up to the step it left the fp64 trajectory, and the distribution of those
distances over ALL envs."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from marinevehiclereinforcementlearning_amd import _lib, params as P  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from parity_util import NAMES, OutlierAudit  # noqa: E402

orc.build()


def batch(dof, n, steps, seed):
    rng = np.random.default_rng(seed)
    npos = 3 if dof == 6 else 2
    init = np.concatenate([(rng.random((n, 2 * npos)) - 0.5) * 10, rng.random((n, dof - npos)) * 2 * np.pi], axis=1).astype(np.float32)
    return init, rng.uniform(-1, 1, size=(steps, n, dof)).astype(np.float32)


def circ(a, b, cols):
    d = np.abs(np.asarray(a, np.float64) - b)
    d[:, cols] = np.minimum(d[:, cols], np.abs(d[:, cols] - 2 * np.pi))
    return (d / np.maximum(1.0, np.abs(b))).max(axis=1)


allm = []
for dof, mode, n_sub in [(6, 0, 4), (6, 1, 4), (3, 0, 4), (3, 1, 4), (6, 0, 2), (6, 0, 8)]:
    n, steps = int(os.environ.get("N", 4096)), int(os.environ.get("STEPS", 25))
    init, actions = batch(dof, n, steps, 77 + dof)
    h = _lib.Handle(P.make_config("rov6" if dof == 6 else "rov3", n, n_substeps=n_sub, control_mode=mode, auto_reset=False,
                                  max_steps=10 ** 9, use_flow=False))
    env = orc.OracleRovEnv(dof, n, "f64", n_substeps=n_sub, control_mode=mode, max_steps=10 ** 9)
    env.reset(init.astype(np.float64))
    h.reset(init=init)
    audit = OutlierAudit(n, 1e-5)
    for s in range(steps):
        env.step(actions[s].astype(np.float64))
        h.step(actions[s])
        audit.update(circ(h.get_state()[: 2 * dof].T, env.y, [3, 4, 5] if dof == 6 else [2]), env.margins)
    print(f"==== dof {dof} mode {mode} n_sub {n_sub}")
    print(audit.report())
    print("  per-step distances of ALL envs, percentiles 0.01 0.05 0.1 0.5 1 5 50 % (last step):")
    for k in range(5):
        print(f"    {NAMES[k]:22s}", " ".join(f"{v:9.2e}" for v in np.percentile(env.margins[:, k], [0.01, 0.05, 0.1, 0.5, 1, 5, 50])))
    h.close()
