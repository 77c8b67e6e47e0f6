#!/usr/bin/env python3
"""Distribution of the fp32-vs-fp64 trajectory error of the 6-DoF step kernel (GPU box): per-env maximum scaled error over a
seeded random-action run, as quantiles, plus the drift / jump split of tests/parity_util.OutlierAudit.  Used to compare
kernel revisions (MVRL_LIB selects the library): a change that only re-orders roundings must leave the quantiles where
they were.   python tests/audit/err_quantiles.py [n] [steps] [n_sub] [mode] [dof]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from marinevehiclereinforcementlearning_amd import _lib, params as P  # noqa: E402
from oracle import oracle as oracle_mod  # noqa: E402   (a measuring tool, like the tests: not a product path)
from tests.parity_util import OutlierAudit, SMOOTH_TOL  # noqa: E402
from tests.test_gpu_parity import circ_err  # noqa: E402
from tests.parity_util import random_rov_batch  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    n_sub = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    mode = int(sys.argv[4]) if len(sys.argv) > 4 else P.CTRL_FAITHFUL
    dof = int(sys.argv[5]) if len(sys.argv) > 5 else 6
    init, actions = random_rov_batch(dof, n, steps, 77 + dof)
    h = _lib.Handle(P.make_config("rov6" if dof == 6 else "rov3", n, n_substeps=n_sub, control_mode=mode, auto_reset=False, max_steps=10 ** 9, use_flow=False))
    env = oracle_mod.OracleRovEnv(dof, n, "f64", n_substeps=n_sub, control_mode=mode, max_steps=10 ** 9)
    env.reset(init.astype(np.float64))
    h.reset(init=init)
    audit = OutlierAudit(n, 1e-5, dof=dof)
    for s in range(steps):
        env.step(actions[s].astype(np.float64))
        h.step(actions[s])
        audit.update(circ_err(h.get_state()[:2 * dof].T, env.y, [3, 4, 5] if dof == 6 else [2]).max(axis=1), env.margins)
    e = audit.max_err
    calm = e[~audit.jumped]
    q = np.quantile(calm, [0.5, 0.9, 0.99, 0.999, 0.9999])
    print(f"lib={os.environ.get('MVRL_LIB', 'default')} {h.variant} n={n} steps={steps} n_sub={n_sub} mode={mode}")
    print("  max-error quantiles of the envs that never jumped (50 / 90 / 99 / 99.9 / 99.99 %): " + " ".join(f"{x:.2e}" for x in q))
    print(f"  beyond 1e-5: {int(audit.bad.sum())} ({100 * audit.bad.mean():.3f} %), drifted (<= {SMOOTH_TOL:g}): {int(audit.smooth().sum())} "
          f"({100 * audit.smooth().mean():.4f} %), jumped: {int(audit.jumped.sum())}, unexplained: {int(audit.unexplained().sum())}")
    # how close to a discontinuity the jumps really were: distance / bound of the NEAREST one, over all envs that jumped - the bounds are sharp
    # when the bulk sits well below 1 and only the last per cent near or beyond it
    jm = np.nonzero(audit.jumped)[0]
    if len(jm):
        ratio = audit.margin_at_jump[jm] / audit.bounds
        near = ratio.min(axis=1)
        which = ratio.argmin(axis=1)
        print("  jumps: nearest discontinuity at distance / bound  50 %% %.3f  90 %% %.3f  99 %% %.3f  max %.2f;  by kind: " % tuple(np.quantile(near, [0.5, 0.9, 0.99, 1.0]))
              + ", ".join(f"{nm} {int((which == k).sum())} (90 pct at {np.quantile(near[which == k], 0.9):.2f})" for k, nm in enumerate(
                  ["pid-increment sign", "thruster dead-band", "wind-up limit", "yaw-error branch", "cos(theta)"]) if (which == k).any()))
    from tests.parity_util import NAMES, ensemble_sensitive, ENSEMBLE_NOISE, ENSEMBLE_NOISE_3DOF
    ENSEMBLE_NOISE = ENSEMBLE_NOISE if dof == 6 else ENSEMBLE_NOISE_3DOF
    un = np.nonzero(audit.unexplained())[0]
    kw = dict(n_substeps=n_sub, control_mode=mode)
    # calibration of the perturbation ensemble: its median deviation on ordinary envs, for three noise levels
    calm_lanes = np.nonzero(~audit.bad)[0][:256]
    for nz in (5e-9, 1e-8, 1.5e-8, 2e-8, 3e-8, 5e-8, 1e-7):
        _, med = ensemble_sensitive(oracle_mod, dof, init, actions, calm_lanes, np.full(len(calm_lanes), steps - 1), kw, members=8, noise=nz,
                                    return_median=True)
        print(f"  ensemble noise {nz:.0e}: median deviation of perturbed fp64 runs on 256 ordinary envs {med:.2e} (GPU median {q[0]:.2e})")
    sens = ensemble_sensitive(oracle_mod, dof, init, actions, un, audit.first_jump[un], kw)
    print(f"  of the {len(un)} unexplained envs, {int(sens.sum())} are left by perturbed fp64 runs too (noise {ENSEMBLE_NOISE:g})")
    ex = np.nonzero(audit.explained())[0][:200]
    sens_ex = ensemble_sensitive(oracle_mod, dof, init, actions, ex, audit.first_jump[ex], kw)
    print(f"  control: of {len(ex)} envs explained by the distance bounds, {int(sens_ex.sum())} are sensitive in the ensemble; "
          f"of 256 ordinary envs {int(ensemble_sensitive(oracle_mod, dof, init, actions, calm_lanes, np.full(len(calm_lanes), steps - 1), kw).sum())}")
    for i, sv in zip(un, sens):
        ratios = audit.margin_at_jump[i] / audit.bounds
        print(f"   {'sensitive  ' if sv else 'UNEXPLAINED'} env {i}: jumped at step {audit.first_jump[i]} to {audit.err_at_jump[i]:.1e} (max {audit.max_err[i]:.1e}); "
              f"distance / bound: " + ", ".join(f"{nm} {r:.2f}" for nm, r in zip(NAMES, ratios)))
    h.close()


if __name__ == "__main__":
    main()
