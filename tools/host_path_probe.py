#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point mvrl_step (numpy in / numpy out) - for DESIGN.md."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marinevehiclereinforcementlearning_amd import _lib, params as P
for model, n in [("rov6", 262144), ("rov6", 1048576), ("rov3", 65536)]:
    h = _lib.Handle(P.make_config(model, n, seed=1, use_flow=False))
    h.reset()
    a = np.random.default_rng(0).uniform(-1, 1, size=(n, h.act_dim)).astype(np.float32)
    for _ in range(3):
        h.step(a, copy=False)
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        h.step(a, copy=False)
    dt = (time.perf_counter() - t0) / K
    print(f"host-buffer mvrl_step {model} n={n}: {dt*1e3:.3f} ms/step -> {n/dt:.3e} env-steps/s "
          f"({(h.act_dim*4 + h.obs_dim*4 + 5) * n / dt / 1e9:.1f} GB/s over PCIe incl. staging memcpy)")
    h.close()
