#!/usr/bin/env python3
"""Where does a case of the fp64 configuration sweep leave the fp64 oracle?  python tests/audit/f64_sweep_probe.py <seed> <case> [...]
Prints, per case: the first step at which an env is beyond 1e-8, that env's state / |cos(theta)| / errors per word, and the oracle's own
sensitivity to a 1e-15 perturbation on that env (8-member ensemble).  MVRL_LIB selects the library (A/B against a variant)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from marinevehiclereinforcementlearning_amd import _lib, params as P            # noqa: E402
from marinevehiclereinforcementlearning_amd.synthetic import BASE_DT, synthetic_spod   # noqa: E402
from oracle import flow_ref, oracle as orc                                        # noqa: E402
from tests.parity_util import ensemble_sensitive, fuzz_cases                      # noqa: E402

orc.build()
GOLDEN = os.path.join(REPO, "tests", "golden")
modes, coeffs = synthetic_spod(4, 64)
base = flow_ref.reconstruct(modes, coeffs, np.load(os.path.join(GOLDEN, "ltm.npy")))
bdx, bdy = flow_ref.grid_spacing(np.load(os.path.join(GOLDEN, "turbulence_coords.npy")))
fd, fdx, fdy, fdt = flow_ref.scale(base, bdx, bdy, BASE_DT, 11., 1., 2.)
uv = np.ascontiguousarray(fd[..., :2]).astype(np.float32).astype(np.float64)
args = [int(a) for a in sys.argv[1:]]
for seed, case in zip(args[0::2], args[1::2]):
    c = [c for c in fuzz_cases(seed) if c["case"] == case][0]
    dof, n, fixed, use_flow, kw = c["dof"], c["n"], c["fixed"], c["use_flow"], c["kw"]
    h = _lib.Handle(P.make_config("rov6" if dof == 6 else "rov3", n, dt=c["dt"], n_substeps=c["n_sub"], control_mode=c["mode"],
                                  fixed_setpoint=fixed, auto_reset=False, max_steps=10 ** 9, use_flow=use_flow, precision="f64", **kw))
    if use_flow:
        h.set_flow(uv, fdt, fdx, fdy)
    h.reset(init=c["init"].astype(np.float64))
    st = h.get_state()
    st[-2] = c["toff"]
    h.set_state(st)
    env_kw = dict(dt=c["dt"], n_substeps=c["n_sub"], control_mode=c["mode"], fixed_setpoint=fixed,
                  flow=orc.FlowTable(uv, fdt, fdx, fdy) if use_flow else None, **kw)
    env = orc.OracleRovEnv(dof, n, "f64", max_steps=10 ** 9, **env_kw)
    env.reset(c["init"].astype(np.float64), toffset=c["toff"])
    ang = [3, 4, 5] if dof == 6 else [2]
    print(f"== seed {seed} case {case}: {h.variant} n {n} fixed {fixed} flow {use_flow} mode {c['mode']} n_sub {c['n_sub']} dt {c['dt']} lib {os.environ.get('MVRL_LIB', 'default')}")
    reported = False
    for k in range(c["steps"]):
        a = c["actions"][k].astype(np.float64)
        y_before = env.y.copy()
        env.step(a)
        h.step(None if fixed else a)
        yg = h.get_state()[: 2 * dof].T
        d = np.abs(yg - env.y)
        d[:, ang] = np.minimum(d[:, ang], np.abs(d[:, ang] - 2 * np.pi))
        e = (d / np.maximum(1.0, np.abs(env.y))).max(axis=1)
        bad = np.nonzero(e > 1e-8)[0]
        print(f"  step {k}: worst {e.max():.2e}, {len(bad)} envs beyond 1e-8" + (f", min |cos theta| {np.abs(np.cos(env.y[:, 4])).min():.2e}" if dof == 6 else ""))
        if len(bad) and not reported:
            reported = True
            i = int(bad[np.argmax(e[bad])])
            np.set_printoptions(precision=6, linewidth=200)
            print(f"    env {i}: state before the step {y_before[i]}")
            print(f"    oracle after {env.y[i]}\n    kernel after {yg[i]}\n    |diff|       {d[i]}")
            if dof == 6:
                print(f"    cos(theta) before / after: {np.cos(y_before[i, 4]):.3e} / {np.cos(env.y[i, 4]):.3e}")
            sens = ensemble_sensitive(orc, dof, c["init"], [c["actions"][q] for q in range(c["steps"])], np.array([i]), np.array([k]), env_kw,
                                      c["toff"], members=16, noise=1e-15)
            print(f"    the oracle perturbed at 1e-15 leaves its own trajectory on this env by step {k}: {bool(sens[0])}")
    h.close()
