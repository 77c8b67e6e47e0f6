"""The exactness path on the GPU: precision = f64 (the same kernel text widened to double) and
integrator = rk45 (the reference's scipy solve_ivp restated per lane).

  * fp64 RK4 kernels vs the reference-derived goldens and vs the fp64 oracle: 1e-9, NO outlier lanes
    (the 0.1-0.7 % discontinuity lanes of the fp32 path are an fp32 effect);
  * fp64 RK45 kernels vs the reference's REAL env.step trajectories (goldens g10), including BASELINE
    configs[0] (3-DoF, 1 env, 1000 random-action steps) with the identical RHS-call count per step.
"""
import os

import numpy as np
import pytest

from .conftest import GOLDEN, golden, max_scaled_err
from .parity_util import ensemble_deviation, random_rov_batch
from .test_gpu_parity import circ_err, rov_init
from marinevehiclereinforcementlearning_amd import _lib, params as P
from marinevehiclereinforcementlearning_amd.synthetic import BASE_DT, synthetic_spod

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,dof,n_sub,mode", [
    ("g09_rk4_6dof_faithful_nsub4.npz", 6, 4, P.CTRL_FAITHFUL),
    ("g09_rk4_6dof_faithful_nsub4_x64.npz", 6, 4, P.CTRL_FAITHFUL),      # 64 envs x 100 steps (SURVEY 8(c) G9's sample size)
    ("g09_rk4_6dof_zoh_nsub4_x32.npz", 6, 4, P.CTRL_ZOH),                # 32 envs x 60 steps
    ("g09_rk4_6dof_faithful_nsub8.npz", 6, 8, P.CTRL_FAITHFUL),
    ("g09_rk4_6dof_zoh_nsub4.npz", 6, 4, P.CTRL_ZOH),
    ("g09_rk4_6dof_fixedsp_nsub4.npz", 6, 4, P.CTRL_FAITHFUL),
    ("g09_rk4_3dof_faithful_nsub4.npz", 3, 4, P.CTRL_FAITHFUL),
    ("g09_rk4_3dof_fixedsp_nsub4.npz", 3, 4, P.CTRL_FAITHFUL),
])
def test_f64_rk4_goldens(name, dof, n_sub, mode):
    g = golden(name)
    n_env, n_steps = g["actions"].shape[:2]
    h = _lib.Handle(P.make_config("rov6" if dof == 6 else "rov3", n_env, n_substeps=n_sub, control_mode=mode,
                                  fixed_setpoint=bool(g["fixedSp"]), auto_reset=False, max_steps=10 ** 9, use_flow=False,
                                  precision="f64"))
    assert h.variant.endswith("/f64") and h.dtype == np.float64
    obs0 = h.reset(init=rov_init(g, dof)).copy()
    assert max_scaled_err(obs0, g["obs"][:, 0]) < 1e-12
    worst = 0.0
    for s in range(n_steps):
        obs, rew, done = h.step(g["actions"][:, s])
        st = h.get_state()
        worst = max(worst, max_scaled_err(st[: 2 * dof].T, g["states"][:, s + 1]), max_scaled_err(obs, g["obs"][:, s + 1]))
        assert max_scaled_err(st[2 * dof:3 * dof].T, g["eOld"][:, s]) < 1e-9
        assert max_scaled_err(st[3 * dof:4 * dof].T, g["eInt"][:, s]) < 1e-9
    assert worst < 1e-9, worst
    assert np.all(h.step_counter() == n_steps)
    h.close()


@pytest.mark.parametrize("dof,mode,n_sub", [(6, P.CTRL_FAITHFUL, 4), (6, P.CTRL_ZOH, 4), (3, P.CTRL_FAITHFUL, 4),
                                            (6, P.CTRL_FAITHFUL, 8)])
def test_f64_random_batch_has_no_outlier_lanes(oracle_mod, dof, mode, n_sub):
    n, steps = 2048, 25
    init, actions = random_rov_batch(dof, n, steps, 77 + dof)
    h = _lib.Handle(P.make_config("rov6" if dof == 6 else "rov3", n, n_substeps=n_sub, control_mode=mode, auto_reset=False,
                                  max_steps=10 ** 9, use_flow=False, precision="f64"))
    env = oracle_mod.OracleRovEnv(dof, n, "f64", n_substeps=n_sub, control_mode=mode, max_steps=10 ** 9)
    env.reset(init.astype(np.float64))
    h.reset(init=init.astype(np.float64))
    worst = 0.0
    for s in range(steps):
        env.step(actions[s].astype(np.float64))
        h.step(actions[s].astype(np.float64))
        worst = max(worst, circ_err(h.get_state()[: 2 * dof].T, env.y, [3, 4, 5] if dof == 6 else [2]).max())
    print(f"f64 dof={dof} mode={mode} n_sub={n_sub}: worst scaled error over {n} lanes x {steps} steps: {worst:.1e}")
    assert worst < 1e-8, worst
    h.close()


def run_rk45(g, dof, horizon=None):
    n_env, n_steps = g["actions"].shape[:2]
    h = _lib.Handle(P.make_config("rov6" if dof == 6 else "rov3", n_env, fixed_setpoint=bool(g["fixedSp"]), auto_reset=False,
                                  max_steps=10 ** 9, use_flow=False, precision="f64", integrator="rk45"))
    assert "rk45" in h.variant
    h.reset(init=rov_init(g, dof))
    errs, nfev = [], []
    for s in range(n_steps):
        obs, _, _ = h.step(g["actions"][:, s])
        errs.append(max(max_scaled_err(h.get_state()[: 2 * dof].T, g["states"][:, s + 1]),
                        max_scaled_err(obs, g["obs"][:, s + 1])))
        nfev.append(h.get_nfev().copy())
    h.close()
    return np.array(errs), np.array(nfev).T


@pytest.mark.parametrize("name,dof", [("g10_envstep_6dof_random.npz", 6), ("g10_envstep_6dof_fixedsp.npz", 6),
                                      ("g10_envstep_3dof_fixedsp.npz", 3)])
def test_rk45_reproduces_reference_env_step(name, dof):
    """The GPU reproduces the reference's real env.step (adaptive scipy RK45 + stateful PID in the RHS)."""
    g = golden(name)
    horizon = None
    if "states_twin" in g.files:   # the reference's own reproducibility horizon (see tests/test_oracle_traj.py)
        twin = np.abs(g["states"] - g["states_twin"]).max(axis=(0, 2))
        if (twin > 1e-10).any():
            horizon = max(1, int(np.argmax(twin > 1e-10)) - 4)
    errs, nfev = run_rk45(g, dof)
    hz = len(errs) if horizon is None else horizon
    # fixed set-point runs sit ON the set-point, where the PID derivative (e-eOld)/1e-9 amplifies last-bit differences
    # (fma vs separate multiply-add) by 1e9: same bound as the oracle's own test, one decade looser there
    assert errs[:hz].max() < (1e-6 if bool(g["fixedSp"]) else 1e-8), (errs[:hz].max(), hz)
    assert np.array_equal(nfev[:, :hz], g["ncalls"][:, :hz])    # identical accept/reject sequence


def test_rk45_config1_3dof_1000_random_steps_on_gpu():
    """BASELINE.json configs[0] - 3-DoF, 1 env, random actions, 1000-step rollout of the reference's CPU path -
    reproduced step for step by the fp64 RK45 kernel (about 1.05 M RHS evaluations in one lane)."""
    g = golden("g10_envstep_3dof_random1000.npz")
    errs, nfev = run_rk45(g, 3)
    assert errs.max() < 1e-7, errs.max()
    assert np.mean(nfev == g["ncalls"]) > 0.999


@pytest.mark.timeout(180)
def test_rk45_lane_with_a_non_finite_state_ends_its_step():
    """scipy's RK45 loop leaves a NaN error norm behind by shrinking the step (fmax(0.2, NaN) = 0.2) until TOO_SMALL_STEP; the fp64 twins
    are compiled with -fno-honor-nans, under which the compiler owes a NaN nothing, so the kernel carries an exit of its own
    (mvrl_rk45.hpp).  A lane whose state is not finite must end its step - a wave that never ends takes the GPU with it - and must not
    disturb the lanes beside it."""
    n = 64
    rng = np.random.default_rng(3)
    init = np.concatenate([(rng.random((n, 6)) - 0.5) * 2.0, rng.random((n, 3)) * 2 * np.pi], axis=1)
    act = rng.uniform(-1, 1, size=(n, 6))
    out = []
    for poison in (False, True):
        h = _lib.Handle(P.make_config("rov6", n, auto_reset=False, max_steps=10 ** 9, use_flow=False, precision="f64", integrator="rk45"))
        h.reset(init=init)
        if poison:
            st = h.get_state()
            st[6, 5] = np.nan          # plane 6 = surge velocity u (include/mvrl.h), env 5
            h.set_state(st)
        h.step(act)
        out.append((h.get_state().copy(), h.get_nfev().copy()))
        h.close()
    (clean, nf0), (pois, nf1) = out
    others = np.arange(n) != 5
    assert np.array_equal(clean[:, others], pois[:, others])       # lanes are independent: bit for bit
    assert np.array_equal(nf0[others], nf1[others])
    assert np.isfinite(clean).all() and nf1[5] <= 2000007          # the poisoned lane ended (at the latest by the kernel's own exit)


def test_rk45_random_batches_follow_the_oracles_scipy_driver():
    """integrator = RK45 on seeded random batches (3- and 6-DoF, random / fixed set-points, turbulence on / off, ragged sizes) against the
    oracle's scipy-faithful driver: states to 1e-7 and the identical number of right-hand-side calls in every env step.  Twelve seeds by hand
    in round 5 (tests/audit/rk45_sweep.py, profiles/r05_rk45_sweep.txt: worst 2.3e-8, 10 128 of 10 128 env steps with identical counts)."""
    from .audit.rk45_sweep import sweep
    assert sweep(2) == 0


def test_f64_auv_ragged_batches_follow_the_oracle():
    """AuvEnv / AuvEnvCyl in precision = f64 on seeded ragged batches (1 ... 1000 envs, with and without turbulence, bounds stop on / off,
    way-point switching, time limits): every lane within 1e-9 of the fp64 oracle in pose, observation and reward.  24 seeds (240 batches)
    were run by hand in round 5 (tests/audit/auv_f64_sweep.py: worst 1.5e-12); three of them here."""
    from .audit.auv_f64_sweep import sweep
    assert sweep(3) == 0


def test_f64_auvenv_golden():
    from oracle import flow_ref
    g = golden("g13_auvenv.npz")
    modes, coeffs = synthetic_spod(int(g["K"]), int(g["nT"]))
    ltm = np.load(os.path.join(GOLDEN, "ltm.npy"))
    coords = np.load(os.path.join(GOLDEN, "turbulence_coords.npy"))
    base = flow_ref.reconstruct(modes, coeffs, ltm)
    bdx, bdy = flow_ref.grid_spacing(coords)
    for e in (1, 3):
        vs, ts = g["flow_scale"][e]
        fd, dx, dy, dt = flow_ref.scale(base, bdx, bdy, BASE_DT, 11., vs, ts)
        h = _lib.Handle(P.make_config("auv", 1, dt=float(g["dt"]), auto_reset=False, use_flow=True, precision="f64",
                                      auv=P.auv_params(stopOnBoundsExceeded=bool(g["stop_on_bounds"][e]))))
        h.set_flow(np.ascontiguousarray(fd[..., :2]), dt, dx, dy)
        h.enable_aux(True)
        init = np.concatenate([g["init"][e], [g["t_offset"][e]], g["mult"][e]])[None]
        obs = h.reset(init=init)
        assert np.max(np.abs(obs[0] - g["obs"][e, 0])) < 1e-12
        for s in range(int(g["n_steps"][e])):
            obs, rew, done = h.step(g["actions"][e, s][None])
            assert max_scaled_err(h.get_state()[:6, 0], g["pose"][e, s + 1]) < 1e-9, s
            assert np.max(np.abs(obs[0] - g["obs"][e, s + 1])) < 1e-9, s
            assert abs(rew[0] - g["reward"][e, s]) < 1e-8 * max(1.0, abs(g["reward"][e, s])), s
            assert max_scaled_err(h.get_aux()[0, 6:11], g["terms"][e, s]) < 1e-9
            assert bool(done[0]) == bool(g["done"][e, s])
        h.close()


def test_precision_entry_points_are_checked():
    h32 = _lib.Handle(P.make_config("rov3", 4, use_flow=False))
    h64 = _lib.Handle(P.make_config("rov3", 4, use_flow=False, precision="f64"))
    buf = np.zeros((h64.state_words, 4))
    assert h32.lib.mvrl_get_state_f64(h32.h, buf.ctypes.data, buf.size) == -1   # fp32 handle refuses the f64 call
    assert h64.lib.mvrl_get_state(h64.h, buf.ctypes.data, buf.size) == -1
    with pytest.raises(_lib.MvrlError, match="F64"):
        _lib.Handle(P.make_config("rov6", 4, use_flow=False, integrator="rk45"))    # RK45 needs fp64
    h32.close(); h64.close()


@pytest.mark.parametrize("which", ["c4", "c3", "c2"])
def test_f64_whole_episode_follows_the_reference(oracle_mod, which):
    """The mode that follows the reference for a WHOLE 250-step episode (6DoF.py:569-571) on the benched populations (VERDICT r4
    "next 1"): precision = f64, 4096 envs, Philox resets, uniform random actions.  Under random actions the closed loop is chaotic
    (every fp32 build loses 40 % of the C4 envs by step 250, tests/audit/attribution_cpu.py says why); in fp64 NO env may leave 1e-5
    and the worst one stays below 1e-6."""
    from marinevehiclereinforcementlearning_amd.flow import ReconstructedFlow
    n, steps = 4096, 250
    dof = 3 if which == "c2" else 6
    npos = 3 if dof == 6 else 2
    h = _lib.Handle(P.make_config("rov6" if dof == 6 else "rov3", n, auto_reset=False, max_steps=10 ** 9, use_flow=which == "c4",
                                  seed=12345, precision="f64"))
    ft = None
    if which == "c4":
        flow = ReconstructedFlow.synthetic(n_modes=8, n_time=2000)
        flow.scale(11., 1., 2., translate=(-1.65, -1.1))
        uv = flow.table_uv()
        h.set_flow(uv, flow.dt, flow.dx, flow.dy)
        ft = oracle_mod.FlowTable(uv.astype(np.float64), flow.dt, flow.dx, flow.dy)
    h.reset()
    st = h.get_state()
    init = np.concatenate([st[5 * dof:5 * dof + 2 * npos].T, st[4 * dof:5 * dof].T[:, npos:]], axis=1)
    ref = oracle_mod.OracleRovEnv(dof, n, "f64", max_steps=10 ** 9, flow=ft)
    ref.reset(init, toffset=st[-2].copy())
    ang = [3, 4, 5] if dof == 6 else [2]
    rng = np.random.default_rng(2024)
    worst = np.zeros(n)
    for s in range(steps):
        a = rng.uniform(-1, 1, (n, dof)).astype(np.float32).astype(np.float64)
        ref.step(a)
        h.step(a)
        worst = np.maximum(worst, circ_err(h.get_state()[:2 * dof].T, ref.y, ang).max(axis=1))
    print(f"f64 {which}: worst env {worst.max():.1e}, median {np.median(worst):.1e}, beyond 1e-5: {int((worst > 1e-5).sum())} of {n} after {steps} steps")
    assert (worst > 1e-5).sum() == 0 and worst.max() < 1e-6, worst.max()
    h.close()


# 2024 ... 19: the first five; 4 ... 114: seeds on which the kernels of rounds 3-5 left the fp64 oracle by O(1) in ONE step (a vehicle turning through
# theta = +-90 deg: the carried yaw error was one turn short when the heading moved by more than a circle within an RK stage) and seeds on
# which the reference itself is unstable at 1e-15 (29, 114): found by running seeds 1 ... 128 in round 5's second sitting (all green now)
F64_SWEEP_SEEDS = [2024, 3, 7, 11, 19, 4, 29, 30, 36, 82, 94, 102, 114]
if os.environ.get("MVRL_FUZZ_SEEDS"):      # exploratory, like the fp32 sweep: MVRL_FUZZ_SEEDS=79,102,104 python -m pytest tests/test_gpu_f64.py -m gpu -k config_sweep
    F64_SWEEP_SEEDS = [int(x) for x in os.environ["MVRL_FUZZ_SEEDS"].split(",")]


@pytest.mark.parametrize("seed", F64_SWEEP_SEEDS)
def test_f64_config_sweep_has_no_budget(oracle_mod, seed):
    """The configuration sweep of tests/test_gpu_parity.py::test_config_fuzz_vs_oracle (ragged batch sizes, odd sub-step counts, other dt,
    fixed set-point x turbulence x controller placement x kernel flavour) through precision = f64: the SAME kernel text, and NO outlier
    accounting, drift budget or second yardstick - every env of every case within 1e-8 of the fp64 oracle (or, for seeds beyond the suite's,
    no further from it than the oracle is from itself when perturbed at 1e-15: see below).  What the fp32 sweep budgets for is therefore
    rounding, not logic: any branch, flavour or launch geometry the sweep reaches computes the reference's algorithm."""
    from oracle import flow_ref
    from .parity_util import fuzz_cases
    from .conftest import GOLDEN
    modes, coeffs = synthetic_spod(4, 64)
    ltm = np.load(os.path.join(GOLDEN, "ltm.npy"))
    base = flow_ref.reconstruct(modes, coeffs, ltm)
    coords = np.load(os.path.join(GOLDEN, "turbulence_coords.npy"))
    bdx, bdy = flow_ref.grid_spacing(coords)
    fd, fdx, fdy, fdt = flow_ref.scale(base, bdx, bdy, BASE_DT, 11., 1., 2.)
    uv = np.ascontiguousarray(fd[..., :2]).astype(np.float32).astype(np.float64)
    variants, worst_all = set(), 0.0
    for c in fuzz_cases(seed):
        dof, n, fixed, use_flow, kw = c["dof"], c["n"], c["fixed"], c["use_flow"], c["kw"]
        h = _lib.Handle(P.make_config("rov6" if dof == 6 else "rov3", n, dt=c["dt"], n_substeps=c["n_sub"], control_mode=c["mode"],
                                      fixed_setpoint=fixed, auto_reset=False, max_steps=10 ** 9, use_flow=use_flow, precision="f64", **kw))
        if use_flow:
            h.set_flow(uv, fdt, fdx, fdy)
        h.reset(init=c["init"].astype(np.float64))
        st = h.get_state()
        st[-2] = c["toff"]
        h.set_state(st)
        env = oracle_mod.OracleRovEnv(dof, n, "f64", dt=c["dt"], n_substeps=c["n_sub"], control_mode=c["mode"], fixed_setpoint=fixed,
                                      max_steps=10 ** 9, flow=oracle_mod.FlowTable(uv, fdt, fdx, fdy) if use_flow else None, **kw)
        env.reset(c["init"].astype(np.float64), toffset=c["toff"])
        worst, errs = 0.0, []
        for k in range(c["steps"]):
            a = c["actions"][k].astype(np.float64)
            o_ref, _, _ = env.step(a)
            o_gpu, _, _ = h.step(None if fixed else a)
            errs.append(np.maximum(circ_err(h.get_state()[: 2 * dof].T, env.y, [3, 4, 5] if dof == 6 else [2]).max(axis=1),
                                   np.abs(o_gpu - o_ref).max(axis=1)))
            worst = max(worst, float(errs[-1].max()))
        variants.add(h.variant)
        errs = np.array(errs)                                   # [step, env]
        out = np.nonzero((errs >= 1e-8).any(axis=0))[0]
        if len(out):
            # An env beyond 1e-8 (none in the suite's seeds; about one case in ten of the seeds beyond them, all in the fixed-set-point corner
            # where target attitudes send vehicles through theta = +-90 deg) is accepted only where the fp64 REFERENCE is as unstable as
            # that: perturbed at 1e-15 - its own rounding level - after every sub-step, 64 members, it must move at least a thirtieth as
            # far as the kernel is off, at the same step.  (Round 5, second sitting: this is the test that found the carried yaw error
            # wrong for a heading that turns by more than a circle within one RK stage - 1.4 against an ensemble spread of 4e-7.)
            env_kw = dict(dt=c["dt"], n_substeps=c["n_sub"], control_mode=c["mode"], fixed_setpoint=fixed,
                          flow=oracle_mod.FlowTable(uv, fdt, fdx, fdy) if use_flow else None, **kw)
            dev = ensemble_deviation(oracle_mod, dof, c["init"], [c["actions"][k] for k in range(c["steps"])], out, env_kw, c["toff"])
            for j, i in enumerate(out):
                ks = np.nonzero(errs[:, i] >= 1e-8)[0]
                ok = errs[ks, i] <= 30.0 * np.maximum.accumulate(dev[:, j])[ks]
                print(f"f64 sweep seed {seed} case {c['case']} env {i}: beyond 1e-8 from step {ks[0]} (worst {errs[:, i].max():.1e}); the fp64 oracle "
                      f"perturbed at 1e-15 moves by {dev[ks[0], j]:.1e} there, {dev[:, j].max():.1e} at most: {'unstable reference' if ok.all() else 'NOT explained'}")
                assert ok.all(), (seed, c["case"], h.variant, int(i), errs[:, i], dev[:, j])
        else:
            worst_all = max(worst_all, worst)
        h.close()
    print(f"f64 sweep seed {seed}: worst {worst_all:.1e} over 24 cases, kernels {sorted(variants)}")
    assert len(variants) >= 5
