#!/usr/bin/env python3
"""Development probe (run on the GPU box): parity of the HIP path vs goldens + quick timing. Not a test."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from marinevehiclereinforcementlearning_amd import _lib, params as P  # noqa: E402

G = os.path.join(REPO, "tests", "golden")


def serr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def traj(name, dof, n_sub, mode=P.CTRL_FAITHFUL, **over):
    g = np.load(os.path.join(G, name))
    n_env, n_steps = g["actions"].shape[:2]
    model = P.MODEL_ROV6 if dof == 6 else P.MODEL_ROV3
    kw = {}
    if over:
        kw["rov6" if dof == 6 else "rov3"] = (P.rov6_params if dof == 6 else P.rov3_params)(**over)
    cfg = P.make_config(model, n_env, n_substeps=n_sub, control_mode=mode, fixed_setpoint=bool(g["fixedSp"]),
                        auto_reset=False, max_steps=10 ** 9, use_flow=False, **kw)
    h = _lib.Handle(cfg)
    npos = 3 if dof == 6 else 2
    init = np.concatenate([g["path"].reshape(n_env, 2 * npos), g["sp0"][:, npos:]], axis=1)
    obs0 = h.reset(init=init).copy()
    e_obs, e_st, worst = serr(obs0, g["obs"][:, 0]), 0.0, []
    for s in range(n_steps):
        obs, rew, done = h.step(g["actions"][:, s])
        st = h.get_state()[: 2 * dof].T
        e = serr(st, g["states"][:, s + 1])
        worst.append(e)
        e_st = max(e_st, e)
        e_obs = max(e_obs, serr(obs, g["obs"][:, s + 1]))
    print(f"{name:42s} {h.variant:28s} state {e_st:.2e} obs {e_obs:.2e}  (steps>1e-5: {int(np.sum(np.array(worst) > 1e-5))}/{n_steps})")
    h.close()
    return worst


def timing(model, n, steps=50, use_flow=False, n_sub=4, mode=P.CTRL_FAITHFUL):
    cfg = P.make_config(model, n, n_substeps=n_sub, control_mode=mode, auto_reset=True, use_flow=use_flow, seed=1)
    h = _lib.Handle(cfg)
    if use_flow:
        nt = 2000
        tab = (np.random.default_rng(0).standard_normal((nt, 41, 61, 2)) * 0.1).astype(np.float32)
        tab[..., 0] += 1
        h.set_flow(tab, 0.022, 0.055, 0.055)
    act_dim, obs_dim = h.act_dim, h.obs_dim
    a = h.dev_alloc(n * act_dim * 4)
    o = h.dev_alloc(n * obs_dim * 4)
    r = h.dev_alloc(n * 4)
    d = h.dev_alloc(n)
    h.fill_uniform_dev(a, n * act_dim, 12345, 0)
    h.reset_dev(None, None, o)
    for _ in range(5):
        h.step_dev(a, o, r, d)
    h.synchronize()
    h.timing_begin()
    t0 = time.time()
    for _ in range(steps):
        h.step_dev(a, o, r, d)
    ms, nl = h.timing_end()
    wall = time.time() - t0
    per = ms / nl
    print(f"timing {h.variant:30s} n={n:8d}: {per*1e3:9.1f} us/step  {n/per*1e3:.3e} env-steps/s  (wall {wall/steps*1e6:.0f} us/step)")
    obs = np.zeros((n, obs_dim), np.float32)
    h.dev_download(o, obs)
    assert np.isfinite(obs).all()
    h.close()


if __name__ == "__main__":
    print("devices:", _lib.device_count())
    traj("g09_rk4_6dof_faithful_nsub4.npz", 6, 4)
    traj("g09_rk4_6dof_faithful_nsub8.npz", 6, 8)
    traj("g09_rk4_6dof_faithful_nsub2.npz", 6, 2)
    traj("g09_rk4_6dof_zoh_nsub4.npz", 6, 4, P.CTRL_ZOH)
    traj("g09_rk4_6dof_fixedsp_nsub4.npz", 6, 4)
    traj("g09_rk4_3dof_faithful_nsub4.npz", 3, 4)
    traj("g09_rk4_3dof_faithful_nsub8.npz", 3, 8)
    traj("g09_rk4_3dof_fixedsp_nsub4.npz", 3, 4)
    for n in [65536, 262144, 1048576]:
        timing(P.MODEL_ROV6, n)
    timing(P.MODEL_ROV6, 1048576, use_flow=True)
    timing(P.MODEL_ROV6, 262144, mode=P.CTRL_ZOH)
    timing(P.MODEL_ROV3, 65536)
    timing(P.MODEL_ROV3, 1048576)
    timing(P.MODEL_AUV, 1048576, use_flow=True)
