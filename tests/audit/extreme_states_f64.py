#!/usr/bin/env python3
"""precision = f64 against the fp64 oracle from ARBITRARY states - not the states a roll-out reaches from a reset: body rates up to
+-6 rad/s, velocities up to +-2 m/s, a fifth of the vehicles within |cos(theta)| < 0.1 of gimbal lock, controller memory and integrals
filled with random numbers, errors beyond the wind-up limits, set-points metres and radians away.  Both sides are SET to the same state
(mvrl_set_state; the oracle's arrays) and stepped three times.  python tests/audit/extreme_states_f64.py [n_seeds [dof]]
Prints, per (seed, mode): the share of envs within 1e-8 and, of those beyond, how many came within |cos(theta)| < 0.05 during the steps
(J2 ~ 1 / cos(theta): no two fp64 evaluations agree there) - an env beyond 1e-8 that stayed away from gimbal lock is a FINDING."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from marinevehiclereinforcementlearning_amd import _lib, params as P            # noqa: E402
from marinevehiclereinforcementlearning_amd.synthetic import BASE_DT, synthetic_spod   # noqa: E402
from oracle import flow_ref, oracle as orc                                        # noqa: E402


def sweep(n_seeds, first_seed=0, n=1000, steps=3, dof=6):
    orc.build()
    golden = os.path.join(REPO, "tests", "golden")
    modes, coeffs = synthetic_spod(4, 64)
    base = flow_ref.reconstruct(modes, coeffs, np.load(os.path.join(golden, "ltm.npy")))
    bdx, bdy = flow_ref.grid_spacing(np.load(os.path.join(golden, "turbulence_coords.npy")))
    fd, dx, dy, dt = flow_ref.scale(base, bdx, bdy, BASE_DT, 11., 1., 2.)
    uv = np.ascontiguousarray(fd[..., :2]).astype(np.float32).astype(np.float64)
    findings = 0
    pl = P.STATE_PLANES[P.MODEL_ROV6 if dof == 6 else P.MODEL_ROV3]
    npos, nang = (3, 3) if dof == 6 else (2, 1)
    ang = list(range(npos, dof))
    for seed in range(first_seed, first_seed + n_seeds):
        rng = np.random.default_rng(31000 + seed)
        mode = int(rng.integers(0, 2))
        fixed = bool(rng.integers(0, 2))
        use_flow = bool(rng.integers(0, 2))
        n_sub = int(rng.choice([2, 3, 4, 5, 8]))
        y = np.zeros((n, 2 * dof))
        y[:, :npos] = (rng.random((n, npos)) - 0.5) * (0.6 if use_flow else 10.0)
        y[:, npos:dof] = rng.random((n, nang)) * 2 * np.pi
        if dof == 6:
            near = rng.random(n) < 0.2                              # a fifth next to theta = +-90 deg
            y[near, 4] = (np.pi / 2 + (rng.random(near.sum()) - 0.5) * 0.2 + np.pi * rng.integers(0, 2, near.sum())) % (2 * np.pi)
        y[:, dof:dof + npos] = (rng.random((n, npos)) - 0.5) * 4.0
        y[:, dof + npos:] = (rng.random((n, nang)) - 0.5) * (12.0 if dof == 6 else 8.0)
        sp = np.concatenate([y[:, :npos] + (rng.random((n, npos)) - 0.5) * 6.0, rng.random((n, nang)) * 2 * np.pi], axis=1)
        path = (rng.random((n, 2 * npos)) - 0.5) * 10.0
        eold = np.concatenate([(rng.random((n, npos)) - 0.5) * 5.0, (rng.random((n, nang)) - 0.5) * 2 * np.pi], axis=1)
        eint = (rng.random((n, dof)) - 0.5) * 2.0
        toff = rng.random(n) * 0.3
        actions = rng.uniform(-1, 1, size=(steps, n, dof))
        h = _lib.Handle(P.make_config("rov6" if dof == 6 else "rov3", n, n_substeps=n_sub, control_mode=mode, fixed_setpoint=fixed, auto_reset=False, max_steps=10 ** 9,
                                      use_flow=use_flow, precision="f64"))
        if use_flow:
            h.set_flow(uv, dt, dx, dy)
        h.reset(init=np.concatenate([path, sp[:, npos:]], axis=1))
        st = h.get_state()
        st[pl["y"]] = y.T
        st[pl["eold"]] = eold.T
        st[pl["eint"]] = eint.T
        st[pl["setpoint"]] = sp.T
        st[pl["path"]] = path.T
        st[pl["toffset"]] = toff
        ist = st[pl["istep"]].copy()
        ist.view(np.int64)[:] = 7                                  # not the first step of an episode: the controller has a memory
        st[pl["istep"]] = ist
        h.set_state(st)
        env = orc.OracleRovEnv(dof, n, "f64", n_substeps=n_sub, control_mode=mode, fixed_setpoint=fixed, max_steps=10 ** 9,
                               flow=orc.FlowTable(uv, dt, dx, dy) if use_flow else None)
        env.reset(np.concatenate([path, sp[:, npos:]], axis=1), toffset=toff)
        env.y[:] = y; env.sp[:] = sp; env.eold[:] = eold; env.eint[:] = eint
        env.has_old[:] = 1; env.istep[:] = 7; env.time[:] = 7 * 0.2
        # the controller's previous call: FAITHFUL - the fourth stage of the previous step's last sub-step, AT the step boundary; ZOH - the
        # start of that sub-step, h earlier (the RK4 kernels do not store tOld: under the harness t - tOld is that pattern)
        env.told[:] = 7 * 0.2 - (0.2 / n_sub if mode == 1 else 0.0)
        worst = np.zeros(n)
        blown = np.zeros(n, bool)       # the ORACLE's own trajectory left the representable / sensible range: the integration diverged (h too large for these rates)
        mincos = np.abs(np.cos(y[:, 4])) if dof == 6 else np.ones(n)
        for k in range(steps):
            env.step(actions[k])
            h.step(None if fixed else actions[k])
            yg = h.get_state()[: 2 * dof].T
            blown |= ~np.isfinite(env.y).all(axis=1) | (np.abs(np.nan_to_num(env.y, nan=0.0, posinf=0.0, neginf=0.0)).max(axis=1) > 1e3)
            d = np.abs(yg - env.y)
            d[:, ang] = np.minimum(d[:, ang], np.abs(d[:, ang] - 2 * np.pi))
            worst = np.maximum(worst, np.nan_to_num((d / np.maximum(1.0, np.abs(env.y))).max(axis=1), nan=np.inf))
            if dof == 6:
                mincos = np.minimum(mincos, np.minimum(env.margins[:, 4], np.abs(np.cos(env.y[:, 4]))))
        out = worst >= 1e-8
        away = out & (mincos >= 0.05) & ~blown
        findings += int(away.sum())
        print(f"seed {seed:3d} mode {mode} fixed {int(fixed)} flow {int(use_flow)} n_sub {n_sub} {h.variant:28s}: {100 * (1 - out.mean()):6.2f} % of {n} envs within 1e-8 "
              f"(median {np.median(worst):.1e}); beyond: {int(out.sum())}: {int((out & (mincos < 0.05)).sum())} came within |cos theta| < 0.05, {int((out & blown).sum())} in runs the oracle itself diverges on (|y| > 1e3 or not finite)"
              + (f"   <-- {int(away.sum())} AWAY FROM GIMBAL LOCK: envs {np.nonzero(away)[0][:6]}, worst {worst[away].max():.1e}, their min |cos theta| {mincos[away].min():.2f}" if away.any() else ""),
              flush=True)
        h.close()
    return findings


if __name__ == "__main__":
    f = sweep(int(sys.argv[1]) if len(sys.argv) > 1 else 12, dof=int(sys.argv[2]) if len(sys.argv) > 2 else 6)
    print("envs beyond 1e-8 away from gimbal lock:", f)
    sys.exit(1 if f else 0)
