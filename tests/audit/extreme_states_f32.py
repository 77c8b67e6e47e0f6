#!/usr/bin/env python3
"""The fp32 kernels against the fp64 oracle from ARBITRARY states (see extreme_states_f64.py), ONE env step: what the fp32-only code -
binary angles, the shorter stage-rotation polynomials, the wave votes - does in corners a roll-out from a reset rarely visits.  Both sides
start from the handle's own state (angles decoded from their binary words), so the inputs are identical.  Per seed: share of envs within 1e-5,
and of those beyond how many sat within the fp32 bounds of a discontinuity of the reference's right-hand side in that step
(tests/parity_util.py) - an env beyond 1e-5 that did not is listed.  python tests/audit/extreme_states_f32.py [n_seeds [dof]]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from marinevehiclereinforcementlearning_amd import _lib, params as P            # noqa: E402
from marinevehiclereinforcementlearning_amd.synthetic import BASE_DT, synthetic_spod   # noqa: E402
from oracle import flow_ref, oracle as orc                                        # noqa: E402
from tests.parity_util import F32_BOUNDS, NAMES                                   # noqa: E402


def sweep(n_seeds, first_seed=0, n=4096, dof=6):
    orc.build()
    golden = os.path.join(REPO, "tests", "golden")
    modes, coeffs = synthetic_spod(4, 64)
    base = flow_ref.reconstruct(modes, coeffs, np.load(os.path.join(golden, "ltm.npy")))
    bdx, bdy = flow_ref.grid_spacing(np.load(os.path.join(golden, "turbulence_coords.npy")))
    fd, dx, dy, dt = flow_ref.scale(base, bdx, bdy, BASE_DT, 11., 1., 2.)
    uv = np.ascontiguousarray(fd[..., :2]).astype(np.float32).astype(np.float64)
    pl = P.STATE_PLANES[P.MODEL_ROV6 if dof == 6 else P.MODEL_ROV3]
    npos, ang = (3, (3, 4, 5)) if dof == 6 else (2, (2,))
    bounds = np.array(F32_BOUNDS, float)
    findings = 0
    for seed in range(first_seed, first_seed + n_seeds):
        rng = np.random.default_rng(41000 + seed)
        mode = int(rng.integers(0, 2))
        fixed = bool(rng.integers(0, 2))
        use_flow = bool(rng.integers(0, 2))
        n_sub = int(rng.choice([3, 4, 5, 8]))
        nang = dof - npos
        y = np.zeros((n, 2 * dof))
        y[:, :npos] = (rng.random((n, npos)) - 0.5) * (0.6 if use_flow else 10.0)
        y[:, npos:dof] = rng.random((n, nang)) * 2 * np.pi
        edge = rng.random(n) < 0.1                                  # a tenth with an angle within 1e-3 of the 0 / 2 pi wrap
        y[edge, npos + rng.integers(0, nang)] = (rng.random(edge.sum()) - 0.5) * 2e-3 % (2 * np.pi)
        y[:, dof:dof + npos] = (rng.random((n, npos)) - 0.5) * 3.0
        y[:, dof + npos:] = (rng.random((n, nang)) - 0.5) * 8.0
        sp = np.concatenate([y[:, :npos] + (rng.random((n, npos)) - 0.5) * 6.0, rng.random((n, nang)) * 2 * np.pi], axis=1)
        path = (rng.random((n, 2 * npos)) - 0.5) * 10.0
        eold = np.concatenate([(rng.random((n, npos)) - 0.5) * 5.0, (rng.random((n, nang)) - 0.5) * 2 * np.pi], axis=1)
        eint = (rng.random((n, dof)) - 0.5) * 2.0
        toff = rng.random(n) * 0.3
        action = rng.uniform(-1, 1, size=(n, dof)).astype(np.float32)
        h = _lib.Handle(P.make_config("rov6" if dof == 6 else "rov3", n, n_substeps=n_sub, control_mode=mode, fixed_setpoint=fixed, auto_reset=False, max_steps=10 ** 9,
                                      use_flow=use_flow))
        if use_flow:
            h.set_flow(uv.astype(np.float32), dt, dx, dy)
        h.reset(init=np.concatenate([path, sp[:, npos:]], axis=1).astype(np.float32))
        st = h.get_state()
        st[pl["y"]] = y.T; st[pl["eold"]] = eold.T; st[pl["eint"]] = eint.T; st[pl["setpoint"]] = sp.T; st[pl["path"]] = path.T
        st[pl["toffset"]] = toff
        ist = st[pl["istep"]].copy(); ist.view(np.int32)[:] = 7; st[pl["istep"]] = ist
        h.set_state(st)
        raw = h.get_state(raw=True)                                # what the kernel will start from, word for word
        y0 = raw[pl["y"]].T.astype(np.float64)
        for q in ang:
            y0[:, q] = raw[q].view(np.uint32).astype(np.float64) * (2 * np.pi / 2 ** 32)
        env = orc.OracleRovEnv(dof, n, "f64", n_substeps=n_sub, control_mode=mode, fixed_setpoint=fixed, max_steps=10 ** 9,
                               flow=orc.FlowTable(uv.astype(np.float32).astype(np.float64), dt, dx, dy) if use_flow else None)
        env.reset(np.concatenate([raw[pl["path"]].T, raw[pl["setpoint"]].T[:, npos:]], axis=1).astype(np.float64), toffset=raw[pl["toffset"]].astype(np.float64))
        env.y[:] = y0; env.sp[:] = raw[pl["setpoint"]].T; env.eold[:] = raw[pl["eold"]].T; env.eint[:] = raw[pl["eint"]].T
        env.has_old[:] = 1; env.istep[:] = 7; env.time[:] = 7 * 0.2
        env.told[:] = 7 * 0.2 - (0.2 / n_sub if mode == 1 else 0.0)
        env.step(action.astype(np.float64))
        h.step(None if fixed else action)
        rg = h.get_state(raw=True)
        yg = rg[pl["y"]].T.astype(np.float64)
        for q in ang:
            yg[:, q] = rg[q].view(np.uint32).astype(np.float64) * (2 * np.pi / 2 ** 32)
        d = np.abs(yg - env.y)
        d[:, list(ang)] = np.minimum(d[:, list(ang)], np.abs(d[:, list(ang)] - 2 * np.pi))
        e = np.nan_to_num((d / np.maximum(1.0, np.abs(env.y))).max(axis=1), nan=np.inf)
        out = e > 1e-5
        blown = ~np.isfinite(env.y).all(axis=1) | (np.abs(np.nan_to_num(env.y, nan=0.0, posinf=0.0, neginf=0.0)).max(axis=1) > 1e3)   # the oracle's own step diverged
        near = (env.margins < bounds).any(axis=1)
        away = out & ~near & ~blown
        findings += int(away.sum())
        line = (f"seed {seed:3d} mode {mode} fixed {int(fixed)} flow {int(use_flow)} n_sub {n_sub} {h.variant:24s}: {100 * (1 - out.mean()):6.2f} % of {n} envs within 1e-5 after one "
                f"step (median {np.median(e):.1e}); beyond: {int(out.sum())}, {int((out & near).sum())} of them next to a discontinuity, {int((out & blown).sum())} where the oracle itself diverges")
        if away.any():
            i = int(np.nonzero(away)[0][np.argmax(e[away])])
            k = int(np.argmin(env.margins[i] / bounds))
            line += (f"   <-- {int(away.sum())} NOT: worst env {i} err {e[i]:.1e}, nearest = {NAMES[k]} at {env.margins[i, k] / bounds[k]:.1f} x bound, "
                     + (f"|cos theta| {abs(np.cos(y0[i, 4])):.2f}, " if dof == 6 else "") + f"rates {np.abs(y0[i, dof + npos:]).max():.1f} rad/s")
        print(line, flush=True)
        h.close()
    return findings


if __name__ == "__main__":
    f = sweep(int(sys.argv[1]) if len(sys.argv) > 1 else 12, dof=int(sys.argv[2]) if len(sys.argv) > 2 else 6)
    print("envs beyond 1e-5 away from every recorded discontinuity:", f)
    sys.exit(1 if f else 0)
