"""CPU oracle vs golden TRAJECTORIES produced by the imported reference:
  G9  - reference derivs under the fixed-step RK4 harness (faithful / ZOH / fixed set-point)
  G10 - the reference's real env.step (scipy RK45), incl. BASELINE config 1 (3-DoF, 1 env, 1000 random steps)
"""
import numpy as np
import pytest

from .conftest import golden, max_scaled_err
from marinevehiclereinforcementlearning_amd import params as P


def run_against(oracle_mod, g, dof, integrator, n_sub=4, mode=P.CTRL_FAITHFUL, max_len=None, horizon=None):
    n_env, n_steps = g["actions"].shape[:2]
    if max_len:
        n_steps = min(n_steps, max_len)
    env = oracle_mod.OracleRovEnv(dof, n_env, "f64", dt=float(g["dt"]), n_substeps=n_sub, integrator=integrator,
                                  control_mode=mode, fixed_setpoint=bool(g["fixedSp"]), max_steps=10 ** 9)
    npos = 3 if dof == 6 else 2
    init = np.concatenate([g["path"].reshape(n_env, 2 * npos), g["sp0"][:, npos:]], axis=1)
    obs0 = env.reset(init)
    err = dict(state=0.0, obs=max_scaled_err(obs0, g["obs"][:, 0]), pid=0.0, rpm=0.0, per_step=[])
    nfev = []
    for s in range(n_steps):
        obs, rew, done = env.step(g["actions"][:, s])
        err["per_step"].append(max_scaled_err(env.y, g["states"][:, s + 1]))
        if horizon is not None and s >= horizon:
            nfev.append(env.nfev.copy())
            continue
        err["state"] = max(err["state"], max_scaled_err(env.y, g["states"][:, s + 1]))
        err["obs"] = max(err["obs"], max_scaled_err(obs, g["obs"][:, s + 1]))
        err["pid"] = max(err["pid"], max_scaled_err(env.eold, g["eOld"][:, s]), max_scaled_err(env.eint, g["eInt"][:, s]))
        err["rpm"] = max(err["rpm"], max_scaled_err(env.rpm / 3500., g["rpm"][:, s] / 3500.))
        assert np.all(rew == 0) and not done.any()
        nfev.append(env.nfev.copy())
    return err, np.array(nfev).T


@pytest.mark.parametrize("name,dof,n_sub,mode", [
    ("g09_rk4_6dof_faithful_nsub4.npz", 6, 4, P.CTRL_FAITHFUL),
    ("g09_rk4_6dof_faithful_nsub4_x64.npz", 6, 4, P.CTRL_FAITHFUL),      # 64 envs x 100 steps (SURVEY 8(c) G9's sample size)
    ("g09_rk4_6dof_zoh_nsub4_x32.npz", 6, 4, P.CTRL_ZOH),                # 32 envs x 60 steps
    ("g09_rk4_6dof_faithful_nsub8.npz", 6, 8, P.CTRL_FAITHFUL),
    ("g09_rk4_6dof_faithful_nsub2.npz", 6, 2, P.CTRL_FAITHFUL),
    ("g09_rk4_6dof_zoh_nsub4.npz", 6, 4, P.CTRL_ZOH),
    ("g09_rk4_6dof_fixedsp_nsub4.npz", 6, 4, P.CTRL_FAITHFUL),
    ("g09_rk4_3dof_faithful_nsub4.npz", 3, 4, P.CTRL_FAITHFUL),
    ("g09_rk4_3dof_faithful_nsub8.npz", 3, 8, P.CTRL_FAITHFUL),
    ("g09_rk4_3dof_fixedsp_nsub4.npz", 3, 4, P.CTRL_FAITHFUL),
])
def test_rk4_harness_trajectories(oracle_mod, name, dof, n_sub, mode):
    g = golden(name)
    assert int(g["n_sub"]) == n_sub
    err, _ = run_against(oracle_mod, g, dof, "rk4", n_sub, mode)
    # fp64 restatement vs fp64 reference: the (e-eOld)/1e-9 derivative amplifies last-bit differences by 1e9,
    # but the saturating clamps swallow almost all of it.
    assert err["state"] < 1e-9, err
    assert err["obs"] < 1e-9, err
    assert err["pid"] < 1e-9, err


@pytest.mark.parametrize("dof", [3, 6])
def test_rk4_harness_trajectories_with_current(oracle_mod, dof):
    """G23: the reference's derivs EXECUTED with a water current that changes every env step (hooks of G21 / G22) under the RK4
    harness - the 3/6-DoF + turbulence composition at trajectory level.  The oracle is served the same sequence through its flow
    lookup (a spatially uniform table whose slices are the per-step currents), so this pins sampling time, hold-over-the-step and the
    right-hand side together."""
    from .parity_util import uniform_current_table
    g = golden(f"g23_rk4_{dof}dof_current.npz")
    n_env, n_steps = g["actions"].shape[:2]
    dt = float(g["dt"])
    ft = oracle_mod.FlowTable(uniform_current_table(g["cur_seq"]), dt, 1.0, 1.0)
    env = oracle_mod.OracleRovEnv(dof, n_env, "f64", dt=dt, n_substeps=int(g["n_sub"]), max_steps=10 ** 9, flow=ft)
    npos = 3 if dof == 6 else 2
    obs0 = env.reset(np.concatenate([g["path"].reshape(n_env, 2 * npos), g["sp0"][:, npos:]], axis=1), toffset=np.zeros(n_env))
    worst = max_scaled_err(obs0, g["obs"][:, 0])
    for s in range(n_steps):
        obs, _, _ = env.step(g["actions"][:, s])
        worst = max(worst, max_scaled_err(env.y, g["states"][:, s + 1]), max_scaled_err(obs, g["obs"][:, s + 1]),
                    max_scaled_err(env.eold, g["eOld"][:, s]), max_scaled_err(env.eint, g["eInt"][:, s]))
    assert worst < 1e-9, worst


def composed_flow_table(g):
    """The table golden G24 was generated on: synthetic SPOD data (K, nT) + the shipped long-time mean, AuvEnv's scaling - through the
    numpy restatement of ReconstructedFlow.__init__ / scale (oracle/flow_ref.py, itself pinned by G12)."""
    import os
    from oracle import flow_ref
    from marinevehiclereinforcementlearning_amd.synthetic import BASE_DT, synthetic_spod
    from .conftest import GOLDEN
    modes, coeffs = synthetic_spod(int(g["K"]), int(g["nT"]))
    base = flow_ref.reconstruct(modes, coeffs, np.load(os.path.join(GOLDEN, "ltm.npy")))
    bdx, bdy = flow_ref.grid_spacing(np.load(os.path.join(GOLDEN, "turbulence_coords.npy")))
    fd, dx, dy, dt = flow_ref.scale(base, bdx, bdy, BASE_DT, *[float(v) for v in g["scale"]])
    assert np.allclose([dx, dy, dt], g["flow_dxdydt"], rtol=1e-12)
    return np.ascontiguousarray(fd[..., :2]), dt, dx, dy


@pytest.mark.parametrize("dof", [3, 6])
def test_fully_reference_composed_trajectories(oracle_mod, dof):
    """G24: the 3/6-DoF + turbulence composition made ENTIRELY of executed reference code - ReconstructedFlow.interp (the reference
    class, AuvEnv's scaling and sampling rule) feeding BlueROV2Heavy{3,6}DoF.derivs through their own velCurrent lines (6-DoF
    hook-assisted), RK4 harness, fixed set-points inside the table: 8 envs x 36 steps.  The oracle's composition (table lookup at
    time + offset and the pre-step position, held over the step; current resolved per RK stage; relative velocity in Ca / D) must
    reproduce states, observations and controller memory to 1e-9."""
    g = golden(f"g24_composed_{dof}dof.npz")
    n_env, n_steps = g["states"].shape[0], g["states"].shape[1] - 1
    uv, fdt, fdx, fdy = composed_flow_table(g)
    env = oracle_mod.OracleRovEnv(dof, n_env, "f64", dt=float(g["dt"]), n_substeps=int(g["n_sub"]), fixed_setpoint=True, max_steps=10 ** 9,
                                  flow=oracle_mod.FlowTable(uv, fdt, fdx, fdy))
    npos = 3 if dof == 6 else 2
    sp = g["sp"]
    env.reset(np.concatenate([sp[:, :npos], sp[:, :npos], sp[:, npos:]], axis=1), toffset=g["toff"])
    env.y[:] = g["start"]
    worst = 0.0
    for s in range(n_steps):
        obs, _, _ = env.step(np.zeros((n_env, dof)))
        worst = max(worst, max_scaled_err(env.y, g["states"][:, s + 1]), max_scaled_err(obs, g["obs"][:, s + 1]),
                    max_scaled_err(env.eold, g["eOld"][:, s]), max_scaled_err(env.eint, g["eInt"][:, s]))
    assert worst < 1e-9, worst


@pytest.mark.parametrize("name,dof", [("g10_envstep_6dof_random.npz", 6), ("g10_envstep_6dof_fixedsp.npz", 6),
                                      ("g10_envstep_3dof_fixedsp.npz", 3)])
def test_envstep_rk45_trajectories(oracle_mod, name, dof):
    """The oracle's scipy-RK45-faithful driver reproduces the reference's real env.step."""
    g = golden(name)
    horizon = None
    if "states_twin" in g.files:
        # The reference's own reproducibility horizon: the same reference run with the set-point moved by one
        # ulp.  Parked at the set-point, the PID derivative (e-eOld)/1e-9 amplifies round-off chaotically (3-DoF:
        # the twin separates after ~20 steps); parity is asserted up to that horizon, boundedness after it.
        twin = np.abs(g["states"] - g["states_twin"]).max(axis=(0, 2))
        if (twin > 1e-10).any():
            horizon = max(1, int(np.argmax(twin > 1e-10)) - 4)
    err, nfev = run_against(oracle_mod, g, dof, "rk45", horizon=horizon)
    assert err["state"] < 1e-8, (err["state"], horizon)
    assert err["obs"] < 1e-8, err["obs"]
    h = nfev.shape[1] if horizon is None else horizon
    # identical accept/reject sequence <=> identical RHS call count per step
    assert np.array_equal(nfev[:, :h], g["ncalls"][:, :h]), (nfev[:, :5], g["ncalls"][:, :5])
    if horizon is not None:
        assert horizon >= 12
        twin_max = float(np.abs(g["states"] - g["states_twin"]).max())
        assert max(err["per_step"]) < 10 * twin_max + 1e-6, (max(err["per_step"]), twin_max)


def test_config1_3dof_1000_random_steps(oracle_mod):
    """BASELINE.json configs[0]: 3-DoF, 1 env, random actions, 1000-step rollout on the reference CPU path."""
    g = golden("g10_envstep_3dof_random1000.npz")
    err, nfev = run_against(oracle_mod, g, 3, "rk45")
    same_calls = float(np.mean(nfev == g["ncalls"]))
    assert err["state"] < 1e-7, (err, same_calls)
    assert same_calls > 0.999, same_calls


def test_perturbation_ensemble_facility(oracle_mod):
    """The oracle's test-only noise switch (orc_set_noise, tests/parity_util.ensemble_sensitive): off by default and after
    every step (two identical runs stay bit-identical, so every golden check above runs on the unperturbed algorithm); on,
    members differ from the plain run by a smooth deviation of the size of the noise times the loop's amplification, the same
    seed reproduces the same member, and envs the distance bounds flag are the ones the ensemble finds sensitive."""
    from .parity_util import ensemble_sensitive, ENSEMBLE_NOISE, F32_BOUNDS
    rng = np.random.default_rng(5)
    n, steps = 256, 12
    init = np.concatenate([(rng.random((n, 6)) - 0.5) * 10, rng.random((n, 3)) * 2 * np.pi], axis=1)
    acts = rng.uniform(-1, 1, size=(steps, n, 6))

    def run(noise=0.0, seed=0):
        env = oracle_mod.OracleRovEnv(6, n, "f64", max_steps=10 ** 9)
        env.reset(init)
        env.noise, env.noise_seed = noise, seed
        for s in range(steps):
            env.step(acts[s])
        return env.y.copy()
    y0, y0b = run(), run()
    assert np.array_equal(y0, y0b)
    y1, y1b, y2 = run(ENSEMBLE_NOISE, 3), run(ENSEMBLE_NOISE, 3), run(ENSEMBLE_NOISE, 4)
    assert np.array_equal(y1, y1b) and not np.array_equal(y1, y2)
    assert np.array_equal(run(), y0)                                  # the switch does not stick
    dev = np.abs(y1 - y0).max(axis=1)
    assert 1e-8 < np.median(dev) < 1e-5, np.median(dev)               # smooth and small for the typical env
    # envs that stay far from every discontinuity are not sensitive; the ensemble flags a small minority
    lanes = np.arange(n)
    sens, med = ensemble_sensitive(oracle_mod, 6, init, acts, lanes, np.full(n, steps - 1), members=8, return_median=True)
    assert med < 1e-5 and sens.mean() < 0.2, (med, sens.mean())
    env = oracle_mod.OracleRovEnv(6, n, "f64", max_steps=10 ** 9)
    env.reset(init)
    near = np.zeros(n, bool)
    for s in range(steps):
        env.step(acts[s])
        near |= (env.margins < 30 * np.asarray(F32_BOUNDS)).any(axis=1)
    assert not (sens & ~near).any()                                    # sensitive envs all passed near a discontinuity
