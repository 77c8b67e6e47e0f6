"""GPU parity tests proper: the HIP path, called through the C ABI (libmvrl.so via ctypes), against
  (1) the committed golden vectors generated from the imported reference, and
  (2) the fp64 CPU oracle on the same seeded inputs.
Tolerance (BASELINE.json north_star): 1e-5 relative fp32, implemented as |a-b| / max(1, |b|) <= 1e-5.
"""
import os

import numpy as np
import pytest

from .conftest import GOLDEN, golden, max_scaled_err
from .parity_util import (FUZZ_BOUNDS, FUZZ_MAX_BAD_SHARE, FUZZ_MAX_DRIFT_SHARE, MAX_FALSE_EXCUSE, OutlierAudit, fuzz_cases, make_resolver,
                          random_rov_batch)
from marinevehiclereinforcementlearning_amd import _lib, params as P
from marinevehiclereinforcementlearning_amd.synthetic import BASE_DT, synthetic_spod

pytestmark = pytest.mark.gpu
TOL = 1e-5


def circ_err(a, b, ang_cols):
    """scaled error with angle columns compared on the circle (an angle of 2*pi-eps equals -eps)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    d = np.abs(a - b)
    d[..., ang_cols] = np.minimum(d[..., ang_cols], np.abs(d[..., ang_cols] - 2 * np.pi))
    return d / np.maximum(1.0, np.abs(b))


def rov_init(g, dof):
    n_env = g["actions"].shape[0]
    npos = 3 if dof == 6 else 2
    return np.concatenate([g["path"].reshape(n_env, 2 * npos), g["sp0"][:, npos:]], axis=1)


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
def test_reference_rk4_trajectories(oracle_mod, name, dof, n_sub, mode):
    """Reference derivs under the RK4 harness (goldens G9) vs the fused HIP step kernel, step by step.  Every env must stay
    within 1e-5 of the golden trajectory UNLESS the trajectory passed within fp32 reach of a discontinuity of the
    reference's RHS in the step where it left (tests/parity_util.py; the oracle - which reproduces these goldens to 1e-9,
    tests/test_oracle_trajectories.py - runs alongside to report the distances).  Observations, side outputs and PID
    memory are compared on the envs that are still on the golden trajectory."""
    g = golden(name)
    n_env, n_steps = g["actions"].shape[:2]
    fixed = bool(g["fixedSp"])
    cfg = P.make_config("rov6" if dof == 6 else "rov3", n_env, n_substeps=n_sub, control_mode=mode,
                        fixed_setpoint=fixed, auto_reset=False, max_steps=10 ** 9, use_flow=False)
    h = _lib.Handle(cfg)
    assert "baked" in h.variant  # default constants -> the literal-constant kernel
    h.enable_aux(True)
    obs0 = h.reset(init=rov_init(g, dof)).copy()
    assert max_scaled_err(obs0, g["obs"][:, 0]) < TOL
    env = oracle_mod.OracleRovEnv(dof, n_env, "f64", n_substeps=n_sub, control_mode=mode, fixed_setpoint=fixed, max_steps=10 ** 9)
    env.reset(rov_init(g, dof).astype(np.float64))
    ang = [3, 4, 5] if dof == 6 else [2]
    audit = OutlierAudit(n_env, TOL, dof=dof)
    umax = np.array([50., 50., 50., 1., 1., 2.] if dof == 6 else [150., 150., 100.])
    for s in range(n_steps):
        obs, rew, done = h.step(g["actions"][:, s])
        env.step(g["actions"][:, s].astype(np.float64))
        assert circ_err(env.y, g["states"][:, s + 1], ang).max() < 1e-8      # the oracle IS the golden trajectory
        st = h.get_state()
        y = st[: 2 * dof].T
        audit.update(circ_err(y, g["states"][:, s + 1], ang).max(axis=1), env.margins)
        on = ~audit.bad
        assert not done.any() and not rew.any()
        if not on.any():
            continue
        assert max_scaled_err(obs[on], g["obs"][on, s + 1]) < TOL, s
        aux = h.get_aux()
        # side outputs of the LAST RHS call (timeHistory columns): the PID's K_D/dt (= 800..1600 1/s) multiplies
        # the 1e-6 state differences, so these are compared against their full scale at 1e-3
        assert np.max(np.abs(aux[on, :dof] - g["gcf"][on, s]) / umax) < 1e-3, s
        assert np.max(np.abs(aux[on, dof:] - g["rpm"][on, s])) / 3500. < 1e-3, s
        # PID memory (eOld, eInt) lives in the SoA state too
        assert max_scaled_err(st[2 * dof:3 * dof].T[on], g["eOld"][on, s]) < TOL, s
        assert max_scaled_err(st[3 * dof:4 * dof].T[on], g["eInt"][on, s]) < TOL, s
    print(name, audit.report())
    audit.assert_explained(max_smooth_share=1.0 / n_env)
    # envs that left the golden trajectory (each one explained above): at most twice what was measured on MI355X in round 4 with the
    # binary angles (gpurun_out/r4_bam_suite.log: 1 of 64 in the x64 golden, none elsewhere; round 3: 1 of 6, 2 of 64), never a share of the batch
    measured = {"g09_rk4_6dof_faithful_nsub4_x64.npz": 1}.get(name, 0)
    assert audit.bad.sum() <= max(1, 2 * measured), audit.report()
    h.close()


@pytest.mark.parametrize("dof", [3, 6])
@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_reference_rk4_trajectories_with_current(oracle_mod, dof, precision):
    """G23 through the step KERNELS with turbulence on (`FLOW` instances): the reference's derivs executed with a current that changes
    every env step (the hooks of G21 / G22), served to the kernel by its own table lookup - a spatially uniform table whose slice k + 1
    is the current of step k, time spacing = the env's dt, zero offset - so lookup time, hold-over-the-step and the composition's
    right-hand side are pinned together against reference-executed trajectories.  fp64: 1e-9 on every env; fp32: 1e-5 with the usual
    accounting of envs that pass a discontinuity (the oracle, which reproduces G23 to 1e-9, reports the distances)."""
    from .parity_util import uniform_current_table
    g = golden(f"g23_rk4_{dof}dof_current.npz")
    n_env, n_steps = g["actions"].shape[:2]
    dt = float(g["dt"])
    tab = uniform_current_table(g["cur_seq"])
    h = _lib.Handle(P.make_config("rov6" if dof == 6 else "rov3", n_env, dt=dt, n_substeps=int(g["n_sub"]), auto_reset=False,
                                  max_steps=10 ** 9, use_flow=True, precision=precision))
    h.set_flow(tab.astype(h.dtype), dt, 1.0, 1.0)
    assert "+flow" in h.variant
    init = rov_init(g, dof)
    obs0 = h.reset(init=init.astype(h.dtype))            # explicit initial values: the time offset plane is 0
    assert max_scaled_err(obs0, g["obs"][:, 0]) < (1e-12 if precision == "f64" else TOL)
    env = oracle_mod.OracleRovEnv(dof, n_env, "f64", dt=dt, n_substeps=int(g["n_sub"]), max_steps=10 ** 9,
                                  flow=oracle_mod.FlowTable(tab, dt, 1.0, 1.0))
    env.reset(init.astype(np.float64), toffset=np.zeros(n_env))
    ang = [3, 4, 5] if dof == 6 else [2]
    audit = OutlierAudit(n_env, TOL, dof=dof)
    worst = 0.0
    for s in range(n_steps):
        a = g["actions"][:, s]
        obs, _, _ = h.step(a.astype(h.dtype))
        env.step(a.astype(np.float64))
        assert circ_err(env.y, g["states"][:, s + 1], ang).max() < 1e-8
        e = circ_err(h.get_state()[: 2 * dof].T, g["states"][:, s + 1], ang).max(axis=1)
        worst = max(worst, float(e.max()))
        audit.update(e, env.margins)
        on = ~audit.bad
        if on.any():
            assert max_scaled_err(obs[on], g["obs"][on, s + 1]) < (1e-9 if precision == "f64" else 2 * TOL), s
    print(f"g23 dof {dof} {precision}: worst {worst:.1e}; " + audit.report())
    if precision == "f64":
        assert worst < 1e-9, worst
    else:
        audit.assert_explained(max_smooth_share=1.0 / n_env)
        assert audit.bad.sum() <= 1, audit.report()
    h.close()


@pytest.mark.parametrize("dof", [3, 6])
@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_fully_reference_composed_trajectories(oracle_mod, dof, precision):
    """G24 through the step kernels: the benched kind of workload - 3/6-DoF + turbulence - composed ENTIRELY of executed reference code
    (ReconstructedFlow.interp with AuvEnv's scaling and sampling rule feeding the vehicles' own velCurrent lines; 6-DoF hook-assisted;
    RK4 harness, fixed set-points inside the table, 8 envs x 36 steps).  The kernel looks the current up in ITS table (re-packed
    stencil cells, sample time in fp64), holds it over the step and resolves it per RK stage: fp64 1e-9 on every env, fp32 1e-5 with
    the usual accounting (the oracle, which reproduces G24 to 1e-9, reports the distances to the discontinuities)."""
    from .test_oracle_traj import composed_flow_table
    g = golden(f"g24_composed_{dof}dof.npz")
    n_env, n_steps = g["states"].shape[0], g["states"].shape[1] - 1
    uv, fdt, fdx, fdy = composed_flow_table(g)
    npos = 3 if dof == 6 else 2
    sp = g["sp"]
    init = np.concatenate([sp[:, :npos], sp[:, :npos], sp[:, npos:]], axis=1)
    h = _lib.Handle(P.make_config("rov6" if dof == 6 else "rov3", n_env, dt=float(g["dt"]), n_substeps=int(g["n_sub"]), fixed_setpoint=True,
                                  auto_reset=False, max_steps=10 ** 9, use_flow=True, precision=precision))
    h.set_flow(uv.astype(h.dtype), fdt, fdx, fdy)
    h.reset(init=init.astype(h.dtype))
    st = h.get_state()
    st[: 2 * dof] = g["start"].T
    st[-2] = g["toff"]
    h.set_state(st)
    env = oracle_mod.OracleRovEnv(dof, n_env, "f64", dt=float(g["dt"]), n_substeps=int(g["n_sub"]), fixed_setpoint=True, max_steps=10 ** 9,
                                  flow=oracle_mod.FlowTable(uv, fdt, fdx, fdy))
    env.reset(init, toffset=g["toff"])
    env.y[:] = g["start"]
    ang = [3, 4, 5] if dof == 6 else [2]
    audit = OutlierAudit(n_env, TOL, dof=dof)
    worst = 0.0
    for s in range(n_steps):
        obs, _, _ = h.step(None)
        env.step(np.zeros((n_env, dof)))
        assert circ_err(env.y, g["states"][:, s + 1], ang).max() < 1e-8
        e = circ_err(h.get_state()[: 2 * dof].T, g["states"][:, s + 1], ang).max(axis=1)
        worst = max(worst, float(e.max()))
        audit.update(e, env.margins)
        on = ~audit.bad
        if on.any():
            assert max_scaled_err(obs[on], g["obs"][on, s + 1]) < (1e-9 if precision == "f64" else 2 * TOL), s
    print(f"g24 dof {dof} {precision}: worst {worst:.1e}; " + audit.report())
    if precision == "f64":
        assert worst < 1e-9, worst
    else:
        audit.assert_explained(max_smooth_share=1.0 / n_env)
        assert audit.bad.sum() <= 1, audit.report()
    h.close()


@pytest.mark.parametrize("dof,mode,n_sub", [(6, P.CTRL_FAITHFUL, 4), (6, P.CTRL_ZOH, 4), (3, P.CTRL_FAITHFUL, 4),
                                            (3, P.CTRL_ZOH, 4), (6, P.CTRL_FAITHFUL, 2), (6, P.CTRL_FAITHFUL, 8)])
def test_random_batch_vs_fp64_oracle(oracle_mod, dof, mode, n_sub):
    """4096 seeded envs x 25 steps against the fp64 oracle.  Envs beyond 1e-5 are not hidden and not merely counted: each
    one that jumped off the fp64 trajectory must have passed, in that very step, within the stated fp32 bound of a
    discontinuity of the reference's RHS (PID sign at a zero-dt stage, thruster dead-band, wind-up, yaw branch, 1/cos
    theta - tests/parity_util.py); the total is held to twice the measured rate."""
    n, steps = 4096, 25
    init, actions = random_rov_batch(dof, n, steps, 77 + dof)
    cfg = P.make_config("rov6" if dof == 6 else "rov3", n, n_substeps=n_sub, control_mode=mode, auto_reset=False,
                        max_steps=10 ** 9, use_flow=False)
    h = _lib.Handle(cfg)
    env = oracle_mod.OracleRovEnv(dof, n, "f64", n_substeps=n_sub, control_mode=mode, max_steps=10 ** 9)
    o_ref = env.reset(init.astype(np.float64))
    o_gpu = h.reset(init=init)
    assert max_scaled_err(o_gpu, o_ref) < TOL
    ang = [3, 4, 5] if dof == 6 else [2]
    audit = OutlierAudit(n, TOL, dof=dof)      # (rounds 2-3 needed a wider jump line at n_sub 8: the accepted drift reached 6-8e-5 there; with binary angles it is 10 x smaller)
    med = 0.0
    for s in range(steps):
        o_ref, _, _ = env.step(actions[s].astype(np.float64))
        o_gpu, _, _ = h.step(actions[s])
        e = circ_err(h.get_state()[: 2 * dof].T, env.y, ang).max(axis=1)
        audit.update(e, env.margins)
        med = max(med, float(np.median(e)))
        on = ~audit.bad
        assert max_scaled_err(o_gpu[on], o_ref[on]) < 2 * TOL, s
    print(f"dof={dof} mode={mode} n_sub={n_sub}: median err {med:.1e}; " + audit.report())
    # measured (gpurun_out/r2_margins2.log): 7, 2, 1, 1, 4, 29 of 4096 envs for the six parametrisations
    # twice the measured count of round 4 (4 / 12 / 0 of 4096 at n_sub 4 / 8 / 2, 0-1 elsewhere; rounds 2-3: 7 / 29 / 4)
    budget = {(6, P.CTRL_FAITHFUL, 4): 0.002, (6, P.CTRL_FAITHFUL, 8): 0.006, (6, P.CTRL_FAITHFUL, 2): 0.001}.get((dof, mode, n_sub), 0.001)
    # envs that drift past 1e-5 without ever jumping: 0.003 % (FAITHFUL, n_sub 4), 0.03 % (n_sub 8), 0.05 % (ZOH) of 65 536 envs
    # (tests/audit/err_quantiles.py, gpurun_out/r2_errq18.log) - bounded at twice the measured share
    audit.assert_explained(max_share=budget, max_smooth_share=0.001 if mode == P.CTRL_ZOH or n_sub == 8 else 0.0005,
                           resolver=make_resolver(oracle_mod, dof, init, actions, dict(n_substeps=n_sub, control_mode=mode)))
    assert med < 6e-7, med      # round 3: 1.5-2.9e-6; with binary angles 1.7-2.7e-7 (gpurun_out/r4_bam_suite.log)
    h.close()


def _audited_run(oracle_mod, dof, handle, n, steps, init, actions, **env_kw):
    env = oracle_mod.OracleRovEnv(dof, n, "f64", max_steps=10 ** 9, **env_kw)
    env.reset(init.astype(np.float64))
    handle.reset(init=init)
    audit = OutlierAudit(n, TOL, dof=dof)
    for s in range(steps):
        env.step(actions[s].astype(np.float64))
        handle.step(actions[s])
        audit.update(circ_err(handle.get_state()[: 2 * dof].T, env.y, [3, 4, 5] if dof == 6 else [2]).max(axis=1), env.margins)
    print(handle.variant, audit.report())
    audit.assert_explained(max_share=0.004, max_smooth_share=0.001, resolver=make_resolver(oracle_mod, dof, init, actions, env_kw))


def test_generic_kernel_with_modified_constants(oracle_mod):
    """Non-default constants (CG off-axis, cross-coupled damping, asymmetric thrusters) -> generic dense kernel."""
    over = dict(CG=[0.01, -0.015, 0.04], Yr=-0.3, Kv=-0.05, Nvv=-0.4, l_x=0.15, m=12.1, Xuu=-20.0,
                I=[[0.17, 0.002, 0.0], [0.002, 0.15, 0.001], [0.0, 0.001, 0.16]])
    p6 = P.rov6_params(**over)
    n, steps = 1024, 20
    init, actions = random_rov_batch(6, n, steps, 5)
    h = _lib.Handle(P.make_config("rov6", n, auto_reset=False, max_steps=10 ** 9, use_flow=False, rov6=p6))
    assert "generic" in h.variant
    _audited_run(oracle_mod, 6, h, n, steps, init, actions, rov6=p6)
    # structured but non-default numbers -> the "sym" run-time-constant kernel
    p6s = P.rov6_params(m=12.0, Xuu=-19.0, K_P=[20., 25., 25., 8., 10., 1.2])
    h2 = _lib.Handle(P.make_config("rov6", n, auto_reset=False, max_steps=10 ** 9, use_flow=False, rov6=p6s))
    assert "sym" in h2.variant
    _audited_run(oracle_mod, 6, h2, n, steps, init, actions, rov6=p6s)
    # the reference's vehicle with a retuned controller -> "ctrl": vehicle constants as literals, PID numbers at run time
    p6c = P.rov6_params(K_P=[20., 25., 30., 8., 10., 1.2], K_D=[18., 20., 22., 5., 4., 0.7], K_I=[1., 2., 3., 0.1, 0.2, 0.2],
                        forceMomentMaxMagnitudes=[40., 50., 45., 1., 1.5, 2.], windup=[1.5, 2., 2., 1.2, 1.5, 1.5])
    hc = _lib.Handle(P.make_config("rov6", n, auto_reset=False, max_steps=10 ** 9, use_flow=False, rov6=p6c))
    assert "ctrl" in hc.variant
    _audited_run(oracle_mod, 6, hc, n, steps, init, actions, rov6=p6c)
    hc.close()
    # 3-DoF with non-default numbers -> generic
    p3 = P.rov3_params(m=12.0, CG=[0.01, 0.02, 0.02], Yr=-0.2)
    init3, act3 = random_rov_batch(3, n, steps, 6)
    h3 = _lib.Handle(P.make_config("rov3", n, auto_reset=False, max_steps=10 ** 9, use_flow=False, rov3=p3))
    assert "generic" in h3.variant
    _audited_run(oracle_mod, 3, h3, n, steps, init3, act3, rov3=p3)
    for x in (h, h2, h3):
        x.close()


def test_specialised_kernels_for_non_default_constants(oracle_mod):
    """mvrl_specialize: the step kernel compiled at run time (hipcc child process) with the handle's own constants as literals - arbitrary
    constants (dense form), BlueROV2-structured ones and a retuned controller, FAITHFUL and ZOH, with and without the
    turbulence composition - against the fp64 oracle with the same constants, like the ahead-of-time flavours above; K-step
    roll-outs and lane-range launches go through the same compiled function."""
    over = dict(CG=[0.01, -0.015, 0.04], Yr=-0.3, Kv=-0.05, Nvv=-0.4, l_x=0.15, m=12.1, Xuu=-20.0,
                I=[[0.17, 0.002, 0.0], [0.002, 0.15, 0.001], [0.0, 0.001, 0.16]])
    cases = [("generic", P.rov6_params(**over), P.CTRL_FAITHFUL), ("sym", P.rov6_params(m=12.0, Xuu=-19.0, K_P=[20., 25., 25., 8., 10., 1.2]), P.CTRL_FAITHFUL),
             ("ctrl", P.rov6_params(K_P=[20., 25., 30., 8., 10., 1.2], K_D=[18., 20., 22., 5., 4., 0.7]), P.CTRL_ZOH)]
    n, steps = 1024, 20
    init, actions = random_rov_batch(6, n, steps, 5)
    for flavour, p6, mode in cases:
        h = _lib.Handle(P.make_config("rov6", n, control_mode=mode, auto_reset=False, max_steps=10 ** 9, use_flow=False, rov6=p6))
        assert flavour in h.variant and "jit" not in h.variant
        h.reset(init=init)
        for s in range(3):
            h.step(actions[s])
        y_aot = h.get_state()[:12].copy()
        assert h.jit_info()["compiler"] == "none" and h.jit_info()["specialized"] == 0
        assert f"jit-{flavour}" in h.specialize() and h.specialize() == h.variant        # idempotent
        # which compiler built it, and what it made of the kernel (mvrl_jit_info, from the code object's notes): the ROCm
        # installation's hipcc as a child process - this image has it - and no spills; the in-process hiprtc fallback of a
        # process that has PyTorch's older comgr loaded would show 74 SGPR spills and scratch here (profiles/r02_hiprtc_vs_hipcc.txt)
        info = h.jit_info()
        assert info["specialized"] == 1 and info["compiler"] == "hipcc", info
        # (a couple of SGPRs parked in VGPR lanes - the dense form at 3 waves per SIMD shows 2 - cost nothing; scratch would)
        assert info["scratch_bytes"] == 0 and info["vgpr_spills"] == 0 and info["sgpr_spills"] <= 16, info
        assert 64 <= info["vgprs"] <= (128 if info["min_waves_per_simd"] == 4 else 170) and info["lds_bytes"] == 10240, info
        print(f"jit {flavour}: {info}")
        h.reset(init=init)
        for s in range(3):
            h.step(actions[s])
        # same arithmetic, other instruction selection: agreement far inside the parity tolerance for the typical env
        d = np.abs(h.get_state()[:12] - y_aot)
        assert np.median(d.max(axis=0)) < 2e-6, np.median(d.max(axis=0))
        _audited_run(oracle_mod, 6, h, n, steps, init, actions, rov6=p6, control_mode=mode)
        h.close()
    # fixed set-point mode is a kernel flavour of its own (template flag FIXED: displacement coordinates + E0 in LDS): the
    # run-time compiled kernel is built for the handle's mode and agrees with the ahead-of-time one and with the oracle
    p6 = P.rov6_params(m=12.0, Xuu=-19.0)
    hf = _lib.Handle(P.make_config("rov6", n, fixed_setpoint=True, auto_reset=False, max_steps=10 ** 9, use_flow=False, rov6=p6))
    hf.reset(init=init)
    for s in range(5):
        hf.step(actions[s])
    y_aot = hf.get_state()[:12].copy()
    assert "jit-sym" in hf.specialize() and hf.jit_info()["compiler"] == "hipcc"
    hf.reset(init=init)
    for s in range(5):
        hf.step(actions[s])
    assert np.median(np.abs(hf.get_state()[:12] - y_aot).max(axis=0)) < 2e-6
    env = oracle_mod.OracleRovEnv(6, n, "f64", max_steps=10 ** 9, fixed_setpoint=True, rov6=p6)
    env.reset(init.astype(np.float64))
    hf.reset(init=init)
    worst = 0.0
    for s in range(8):
        env.step(actions[s].astype(np.float64))
        hf.step(actions[s])
        worst = max(worst, float(np.median(circ_err(hf.get_state()[:12].T, env.y, [3, 4, 5]).max(axis=1))))
    assert worst < 2e-6, worst
    hf.close()
    # default constants: nothing to do; other models and precisions: refused
    hb = _lib.Handle(P.make_config("rov6", 64, use_flow=False))
    assert hb.specialize() == hb.variant and "baked" in hb.variant
    hb.close()
    for kw in (dict(model="rov3"), dict(model="rov6", precision="f64")):
        hx = _lib.Handle(P.make_config(kw.pop("model"), 64, use_flow=False, **kw))
        with pytest.raises(_lib.MvrlError):
            hx.specialize()
        hx.close()


def test_hiprtc_fallback_is_reported_and_warned_about(monkeypatch):
    """MVRL_JIT_COMPILER=hiprtc: the in-process fallback.  Whatever libhiprtc / comgr this process ends up with, mvrl_jit_info says
    so, and MarineVecEnv warns exactly when that build spills (it then runs 8-20 % slower than the hipcc build)."""
    import warnings
    from marinevehiclereinforcementlearning_amd.vec_env import MarineVecEnv
    monkeypatch.setenv("MVRL_JIT_COMPILER", "hiprtc")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        env = MarineVecEnv("rov6", 256, vehicle_params=P.rov6_params(m=12.0, Xuu=-19.0), specialize=True)
    j = env.jit
    print("hiprtc build:", j)
    assert j["compiler"] == "hiprtc" and j["specialized"] == 1 and "jit-sym" in env.variant
    spills = j["scratch_bytes"] > 0 or j["sgpr_spills"] > 16 or j["vgpr_spills"] > 0
    assert spills == any("spills" in str(x.message) for x in w), (j, [str(x.message) for x in w])
    env.step_tensors(env.reset_tensors().new_zeros((256, 6)))          # and it runs
    env.close()


def test_specialised_kernel_through_vec_env_with_turbulence_chains_and_rollouts():
    """MarineVecEnv(specialize=True) with the turbulence table: lane-range chains and K-step roll-outs of the run-time compiled
    kernel are bit-identical to its whole-batch single steps (one function serves all three), auto-resets included."""
    import torch
    from marinevehiclereinforcementlearning_amd.chains import ChainStepper
    from marinevehiclereinforcementlearning_amd.flow import ReconstructedFlow
    from marinevehiclereinforcementlearning_amd.vec_env import MarineVecEnv
    n, steps = 4096 + 64, 12
    p6 = P.rov6_params(CG=[0.01, -0.015, 0.04], Yr=-0.3, m=12.0)

    def mk():
        f = ReconstructedFlow.synthetic(n_modes=4, n_time=64)
        f.scale(11., 1., 2., translate=(-1.65, -1.1))
        return MarineVecEnv("rov6", n, seed=4, maxSteps=5, flow=f, infos="lean", vehicle_params=p6, specialize=True)
    ea, eb, ec = mk(), mk(), mk()
    assert ea.variant == "rov6/jit-generic/faithful+flow"
    assert ea.jit["compiler"] == "hipcc" and ea.jit["scratch_bytes"] == 0, ea.jit
    # specialize="auto" is the default for non-default constants; False keeps the ahead-of-time kernel
    auto = MarineVecEnv("rov6", 64, vehicle_params=p6)
    aot = MarineVecEnv("rov6", 64, vehicle_params=p6, specialize=False)
    base = MarineVecEnv("rov6", 64)
    assert auto.variant == "rov6/jit-generic/faithful" and aot.variant == "rov6/generic/faithful" and base.variant == "rov6/baked/faithful"
    for e in (auto, aot, base):
        e.close()
    acts = torch.rand((steps, n, 6), device="cuda") * 2 - 1
    for e in (ea, eb, ec):
        e.reset_tensors()
    outs = [tuple(t.clone() for t in ea.step_tensors(acts[k])) for k in range(steps)]
    st = ChainStepper(eb, n_chains=2)
    st.fork()
    bufs = [tuple(torch.empty_like(t) for t in outs[0]) for _ in range(steps)]
    for k in range(steps):
        st.step(acts[k], out=bufs[k])
    st.join()
    torch.cuda.synchronize()
    ro = ec.rollout_tensors(acts)
    for k in range(steps):
        for ta, tb, tc in zip(outs[k], bufs[k], (ro[0][k], ro[1][k], ro[2][k])):
            assert torch.equal(ta, tb) and torch.equal(ta, tc), k
    assert np.array_equal(ea.get_state(), eb.get_state()) and np.array_equal(ea.get_state(), ec.get_state())
    assert int(ea.handle.episode_counter().max()) >= 2
    for e in (ea, eb, ec):
        e.close()


# ---- turbulence field + AuvEnv ---------------------------------------------------------------------
@pytest.fixture(scope="module")
def base_flow():
    from oracle import flow_ref
    g = golden("g12_flow_interp.npz")
    modes, coeffs = synthetic_spod(int(g["K"]), int(g["nT"]))
    ltm = np.load(os.path.join(GOLDEN, "ltm.npy"))
    coords = np.load(os.path.join(GOLDEN, "turbulence_coords.npy"))
    base = flow_ref.reconstruct(modes, coeffs, ltm)
    dx, dy = flow_ref.grid_spacing(coords)
    return base, dx, dy, modes, coeffs, ltm


@pytest.mark.parametrize("tag", ["unit", "auv", "slow"])
def test_flow_interp_vs_reference(base_flow, tag):
    """ReconstructedFlow.interp golden (G12) vs the HIP interpolation kernel; table built by the HIP
    reconstruct kernel (modes x coeffs + ltm, fused with scale())."""
    from oracle import flow_ref
    g = golden("g12_flow_interp.npz")
    base, bdx, bdy, modes, coeffs, ltm = base_flow
    sc = g[f"{tag}_scale"]
    fd, dx, dy, dt = flow_ref.scale(base, bdx, bdy, BASE_DT, sc[0], sc[1], sc[2])
    V, T = sc[1], sc[2]
    mul = [V * T, V * T, 1.0 / max(1e-6, (V * T) ** 2)]
    add = [V - V * T, 0.0, 0.0]
    table = _lib.flow_reconstruct(modes, coeffs, ltm, mul, add)
    assert max_scaled_err(table, fd) < 2e-6
    out = _lib.flow_interp(table, dt, dx, dy, g[f"{tag}_t"], g[f"{tag}_x"], g[f"{tag}_y"])
    ref = g[f"{tag}_out"]
    # far-extrapolated samples (weights up to ~1e2) amplify fp32 rounding of the table: scale by the weight size
    wx = np.maximum(1.0, np.abs(g[f"{tag}_x"] / dx)) * np.maximum(1.0, np.abs(g[f"{tag}_y"] / dy))
    inside = (g[f"{tag}_x"] >= 0) & (g[f"{tag}_x"] <= dx * 60) & (g[f"{tag}_y"] >= 0) & (g[f"{tag}_y"] <= dy * 40) & \
             (g[f"{tag}_t"] >= 0) & (g[f"{tag}_t"] <= dt * (fd.shape[0] - 1))
    assert max_scaled_err(out[inside], ref[inside]) < TOL
    assert np.max(np.abs(out - ref) / (np.maximum(1.0, np.abs(ref)) * wx[:, None])) < TOL


@pytest.mark.parametrize("e", range(6))
def test_auvenv_reference_trajectory(base_flow, e):
    """AuvEnv.step golden trajectories (G13) vs the HIP AuvEnv kernel incl. flow lookup, reward terms, done."""
    from oracle import flow_ref
    g = golden("g13_auvenv.npz")
    base, bdx, bdy = base_flow[:3]
    vs, ts = g["flow_scale"][e]
    fd, dx, dy, dt = flow_ref.scale(base, bdx, bdy, BASE_DT, 11., vs, ts)
    auv = P.auv_params(stopOnBoundsExceeded=bool(g["stop_on_bounds"][e]))
    h = _lib.Handle(P.make_config("auv", 1, dt=float(g["dt"]), auto_reset=False, use_flow=True, auv=auv))
    h.set_flow(np.ascontiguousarray(fd[..., :2], dtype=np.float32), dt, dx, dy)
    h.enable_aux(True)
    init = np.concatenate([g["init"][e], [g["t_offset"][e]], g["mult"][e]])[None]
    obs = h.reset(init=init)
    assert np.max(np.abs(obs[0] - g["obs"][e, 0])) < TOL
    n = int(g["n_steps"][e])
    for s in range(n):
        obs, rew, done = h.step(g["actions"][e, s][None])
        st = h.get_state()[:, 0]
        assert circ_err(st[:6][None], g["pose"][e, s + 1][None], [2]).max() < TOL, s
        assert np.max(np.abs(obs[0] - g["obs"][e, s + 1])) < 2e-5, s
        aux = h.get_aux()[0]
        assert max_scaled_err(aux[3:5], g["vel_current"][e, s]) < TOL, s
        assert max_scaled_err(aux[:3], g["fhydro"][e, s]) < 2e-4, s   # dF/dv ~ 40 N s/m times 1e-6 m/s
        assert abs(aux[5] - g["rms_ac"][e, s]) < 1e-6, s
        assert max_scaled_err(aux[6:11], g["terms"][e, s]) < 2e-5, s
        assert abs(rew[0] - g["reward"][e, s]) < 2e-5 * max(1.0, abs(g["reward"][e, s])), s
        assert bool(done[0]) == bool(g["done"][e, s]), s
    h.close()


def test_flow_interp_bounded_facade(oracle_mod, base_flow):
    """ReconstructedFlow.interp_bounded (the composition's sampling rule, on the host facade) against the oracle's sampler, in
    and far outside the table."""
    import ctypes as C
    from marinevehiclereinforcementlearning_amd.flow import ReconstructedFlow
    from marinevehiclereinforcementlearning_amd.synthetic import synthetic_spod
    from oracle import flow_ref
    base, bdx, bdy = base_flow[:3]
    g = golden("g12_flow_interp.npz")
    modes, coeffs = synthetic_spod(int(g["K"]), int(g["nT"]))
    flow = ReconstructedFlow(modes=modes, coeffs=coeffs, lt_mean=np.load(os.path.join(GOLDEN, "ltm.npy")),
                             coords=np.load(os.path.join(GOLDEN, "turbulence_coords.npy")))
    flow.scale(11., 1., 2., translate=(-1.65, -1.1))
    fd, dx, dy, dt = flow_ref.scale(base, bdx, bdy, BASE_DT, 11., 1., 2.)
    uv = np.ascontiguousarray(fd[..., :2])
    o = oracle_mod.Oracle("f64")
    f = o._f("orc_flow_sample_bounded")
    f.argtypes = [C.c_void_p] + [C.c_int] * 4 + [C.c_double] * 6 + [C.c_void_p]
    f.restype = None
    rng = np.random.default_rng(12)
    n = 400
    t = rng.random(n) * 80.0 - 5.0
    xy = (rng.random((n, 2)) - 0.4) * 12.0
    got = flow.interp_bounded(t, xy)[:, :2]
    ref = np.zeros((n, 2))
    for i in range(n):
        f(uv.ctypes.data, uv.shape[0], uv.shape[1], uv.shape[2], 2, dt, dx, dy, float(t[i]), float(xy[i, 0]), float(xy[i, 1]), ref[i].ctypes.data)
    assert np.max(np.abs(got - ref)) < 2e-5 and np.abs(got).max() < 10.0
    one = flow.interp_bounded(3.0, [1.0, 1.0])
    assert one.shape == (3,) and np.max(np.abs(one - flow.interp(3.0, [1.0, 1.0]))) < 1e-6      # inside the table it IS interp


@pytest.mark.parametrize("where,pos_scale,toff_scale", [("inside the table", 0.05, 2.0), ("outside it, in space and time", 1.0, 60.0)])
def test_rov6_with_turbulence_vs_oracle(oracle_mod, base_flow, where, pos_scale, toff_scale):
    """The 6-DoF + current composition (no reference counterpart) against the fp64 oracle - inside the table (= the
    reference's interp) and outside it (the vehicles start on the table's corner, so half of them are outside at once and the mean
    current carries the rest out within the 20 steps; time offsets up to 60 s - the range of a BASELINE episode - against the
    table's 13 s here, i.e. up to four reflections: boundary value held
    in space, time reflected - DESIGN.md section 1)."""
    from oracle import flow_ref
    base, bdx, bdy = base_flow[:3]
    fd, dx, dy, dt = flow_ref.scale(base, bdx, bdy, BASE_DT, 11., 1., 2.)
    uv = np.ascontiguousarray(fd[..., :2])
    for dof in (6, 3):
        n, steps = 1024, 20
        init, actions = random_rov_batch(dof, n, steps, 31)
        init[:, : (3 if dof == 6 else 2)] *= pos_scale
        toff = np.random.default_rng(2).random(n) * toff_scale
        h = _lib.Handle(P.make_config("rov6" if dof == 6 else "rov3", n, auto_reset=False, max_steps=10 ** 9, use_flow=True))
        h.set_flow(uv.astype(np.float32), dt, dx, dy)
        h.reset(init=init)
        st = h.get_state()
        st[-2] = toff.astype(np.float32)  # flow time offset plane
        h.set_state(st)
        env = oracle_mod.OracleRovEnv(dof, n, "f64", max_steps=10 ** 9, flow=oracle_mod.FlowTable(uv, dt, dx, dy))
        env.reset(init.astype(np.float64), toffset=toff.astype(np.float32))
        audit = OutlierAudit(n, TOL, dof=dof)
        for s in range(steps):
            env.step(actions[s].astype(np.float64))
            h.step(actions[s])
            audit.update(circ_err(h.get_state()[: 2 * dof].T, env.y, [3, 4, 5] if dof == 6 else [2]).max(axis=1), env.margins)
        assert np.isfinite(env.y).all() and np.abs(env.y[:, dof:dof + 2]).max() < 5.0      # a bounded current, bounded speeds
        print(f"dof {dof} + current ({where}): " + audit.report())
        audit.assert_explained(max_share=0.006, max_smooth_share=0.001,
                               resolver=make_resolver(oracle_mod, dof, init, actions, dict(flow=env.flow), toffset=toff))
        h.close()


# 25 seeds x 24 cases in the suite; seeds 25..64 were run once more by hand in round 4 (MVRL_FUZZ_SEEDS=..., gpurun_out/r4_fuzz_more.log): green
FUZZ_SEEDS = [2024] + list(range(1, 25))
if os.environ.get("MVRL_FUZZ_SEEDS"):      # exploratory: MVRL_FUZZ_SEEDS=9,10,11 python -m pytest tests/test_gpu_parity.py -m gpu -k config_fuzz
    FUZZ_SEEDS = [int(x) for x in os.environ["MVRL_FUZZ_SEEDS"].split(",")]


@pytest.mark.parametrize("seed", FUZZ_SEEDS)
def test_config_fuzz_vs_oracle(oracle_mod, base_flow, seed):
    """Seeded sweep over combinations no other test pins: ragged batch sizes (1, 63, 65, 257, 1000: partial tail waves),
    odd sub-step counts, other dt, fixed set-point x turbulence x controller placement x kernel flavour.  Every case is
    compared with the fp64 oracle for 8 env steps; tolerance 1e-5 with the usual outlier-lane accounting.  All seeds are part of
    the suite (round 3 kept one and called the others exploratory)."""
    from oracle import flow_ref
    base, bdx, bdy = base_flow[:3]
    fd, fdx, fdy, fdt = flow_ref.scale(base, bdx, bdy, BASE_DT, 11., 1., 2.)
    uv = np.ascontiguousarray(fd[..., :2])
    report = []
    for c in fuzz_cases(seed):
        case, dof, n, fixed, use_flow, kw, steps = c["case"], c["dof"], c["n"], c["fixed"], c["use_flow"], c["kw"], c["steps"]
        init, actions, toff = c["init"], c["actions"], c["toff"]
        cfg = P.make_config("rov6" if dof == 6 else "rov3", n, dt=c["dt"], n_substeps=c["n_sub"], control_mode=c["mode"],
                            fixed_setpoint=fixed, auto_reset=False, max_steps=10 ** 9, use_flow=use_flow, **kw)
        h = _lib.Handle(cfg)
        if use_flow:
            h.set_flow(uv.astype(np.float32), fdt, fdx, fdy)
        h.reset(init=init)
        st = h.get_state()
        st[-2] = toff
        h.set_state(st)
        env = oracle_mod.OracleRovEnv(dof, n, "f64", dt=c["dt"], n_substeps=c["n_sub"], control_mode=c["mode"], fixed_setpoint=fixed,
                                      max_steps=10 ** 9, flow=oracle_mod.FlowTable(uv, fdt, fdx, fdy) if use_flow else None,
                                      **kw)
        env.reset(init.astype(np.float64), toffset=toff)
        # the SAME C restatement compiled with REAL=float: what any straight fp32 evaluation of the reference's formulation does
        # on this case.  It is the yardstick for envs that drift: fixed set-points with arbitrary target attitude keep half of the
        # vehicles pitched beyond 60 deg and tumbling at 1-2 rad/s with the PID in its linear range, where one ulp of an angle near
        # 2 pi (4.8e-7) x K_D / h x Minv x h = 19 is already 9e-6 rad/s per sub-step (tests/audit/fuzz_isolate.py).
        low = oracle_mod.OracleRovEnv(dof, n, "f32", dt=c["dt"], n_substeps=c["n_sub"], control_mode=c["mode"], fixed_setpoint=fixed,
                                      max_steps=10 ** 9, flow=oracle_mod.FlowTable(uv, fdt, fdx, fdy) if use_flow else None, **kw)
        low.reset(init.astype(np.float64), toffset=toff)
        ang = [3, 4, 5] if dof == 6 else [2]
        audit = OutlierAudit(n, TOL, bounds=FUZZ_BOUNDS, dof=dof)
        audit32 = OutlierAudit(n, TOL, bounds=FUZZ_BOUNDS, dof=dof)
        med = 0.0
        for k in range(steps):
            o_ref, _, _ = env.step(actions[k].astype(np.float64))
            low.step(actions[k])
            audit32.update(circ_err(low.y, env.y, ang).max(axis=1), env.margins)
            o_gpu, _, _ = h.step(None if fixed else actions[k])
            y_gpu = h.get_state()[: 2 * dof].T
            e = circ_err(y_gpu, env.y, ang).max(axis=1)
            audit.update(e, env.margins)
            med = max(med, float(np.median(e)))
            # observations are O(1) quantities with an ABSOLUTE bar, the state's bar is relative above 1: an angle of 3 rad may be 3e-5 rad
            # off and within tolerance, which is 3.8e-5 in the observation (x 4 / pi).  The observation is therefore compared on envs whose
            # angles agree to 1e-5 rad in absolute terms (exploratory seed 14: an env at cos(theta) = 0.06, state error 9.3e-6 relative)
            d_ang = np.abs(y_gpu[:, ang].astype(np.float64) - env.y[:, ang])
            d_ang = np.minimum(d_ang, np.abs(d_ang - 2 * np.pi)).max(axis=1)
            good = ~audit.bad & (d_ang < TOL)
            if good.any():
                assert max_scaled_err(o_gpu[good], o_ref[good]) < 2 * TOL, (case, k)
        bad = audit.bad
        report.append((case, dof, n, c["n_sub"], c["dt"], c["mode"], fixed, use_flow, h.variant, int(bad.sum()), med))
        bad32, drift32 = int(audit32.bad.sum()), int(audit32.smooth().sum())
        print("fuzz seed %d case %2d dof %d n %4d n_sub %d dt %.1f mode %d fixed %d flow %d %-28s: " % ((seed,) + report[-1][:9]) + audit.report()
              + f"\n   fp32 build of the oracle on the same case: {bad32} beyond tol, {drift32} drifted")
        # every env that jumped did so next to a discontinuity; envs that merely drifted past 1e-5 in 8 steps are bounded in number:
        # 1 % of the batch, or - where the case is ill-conditioned for fp32 as such - what the fp32 oracle build drifts (+25 % + 1)
        un = np.nonzero(audit.unexplained())[0]
        n_un, n_ens = len(un), 0
        if len(un):
            # A jump with no recorded discontinuity within the bounds (about 1 jump in 100 happens a few bounds away).  FIRST the model-free
            # test every other trajectory test applies (parity_util.ensemble_sensitive: does the fp64 reference itself, perturbed at the
            # kernel's accepted noise level, leave its own trajectory there?) - used only where it rarely excuses an ordinary env of the
            # same case.  Round 5, second sitting: seeds 25..128 run by hand had 4 such envs in 2 496 cases that the next clause refused.
            env_kw = dict(dt=c["dt"], n_substeps=c["n_sub"], control_mode=c["mode"], fixed_setpoint=fixed,
                          flow=oracle_mod.FlowTable(uv, fdt, fdx, fdy) if use_flow else None, **kw)
            resolver = make_resolver(oracle_mod, dof, init, [actions[k] for k in range(steps)], env_kw, toffset=toff)
            calm = np.nonzero(~audit.bad)[0][:128]
            fer = float(np.mean(resolver(calm, np.full(len(calm), steps - 1)))) if len(calm) else 1.0
            if fer <= MAX_FALSE_EXCUSE:
                sens = np.asarray(resolver(un, audit.first_jump[un]), bool)
                audit.margin_at_jump[un[sens]] = 0.0
                n_ens = int(sens.sum())
                un = un[~sens]
            print(f"fuzz seed {seed} case {case}: {n_un} jumps beyond the distance bounds, ensemble false-excuse rate {100 * fer:.1f} % "
                  f"({'used' if fer <= MAX_FALSE_EXCUSE else 'refused'}): {n_ens} sensitive")
        if len(un):
            # what the ensemble does not excuse (exploratory seed 36: one env of 1000 at 7.5e-5, cos(theta) 0.62, in the fixed-set-point x
            # turbulence x ZOH corner where the fp32 oracle build loses 56 envs to the kernel's 17): accepted only where the fp32 build of
            # the oracle has at least as many of the same kind on the same case - the case, not the kernel, is ill-conditioned
            un32 = int(audit32.unexplained().sum())
            assert len(un) <= un32, (report[-1], f"{len(un)} unexplained jumps, fp32 oracle build {un32}", audit.report())
            audit.margin_at_jump[un] = 0.0
        # how often the sweep actually leans on its second yardstick (the fp32 build of the oracle): printed per case, summed up in DESIGN.md 4
        drifted = int(audit.smooth().sum())
        needs = []
        if drifted > max(1.0, FUZZ_MAX_DRIFT_SHARE * n):
            needs.append(f"drift {drifted} > {max(1.0, FUZZ_MAX_DRIFT_SHARE * n):.0f} (fp32 oracle build: {drift32})")
        if bad.sum() > max(1, int(FUZZ_MAX_BAD_SHARE * n)):
            needs.append(f"beyond tol {int(bad.sum())} > {max(1, int(FUZZ_MAX_BAD_SHARE * n))} (fp32 oracle build: {bad32})")
        if n_ens:
            needs.append(f"{n_ens} jumps beyond the distance bounds excused by the perturbation ensemble (model-free, not the second yardstick)")
        if len(un):
            needs.append(f"{len(un)} jumps without a recorded discontinuity (fp32 oracle build: {int(audit32.unexplained().sum())})")
        if needs:
            print(f"fuzz-yardstick seed {seed} case {case}: " + "; ".join(needs))
        audit.assert_explained(max_smooth_share=max(1.0 / n, FUZZ_MAX_DRIFT_SHARE, (1.25 * drift32 + 1) / n))
        assert bad.sum() <= max(1, int(FUZZ_MAX_BAD_SHARE * n), int(1.25 * bad32 + 1)), (report[-1], audit.report())
        assert med < (3e-6 if n > 1 else TOL), report[-1]      # a batch of one env has no median: it must simply be within tolerance
        h.close()
    for r in report:
        print("fuzz case %2d dof %d n %4d n_sub %d dt %.1f mode %d fixed %d flow %d %-28s outliers %d median %.1e" % r)
    assert len({r[8] for r in report}) >= 6      # the sweep actually reached many kernel instances


@pytest.mark.parametrize("cyl", [False, True])
def test_auv_ragged_batches_vs_oracle(oracle_mod, base_flow, cyl):
    """AuvEnv / AuvEnvCyl batches of ragged size (full waves through the LDS-transposed observation store, partial
    tail waves through the row-per-lane store) against the fp64 oracle: observations, rewards, done flags, poses."""
    from oracle import flow_ref
    base, bdx, bdy = base_flow[:3]
    fd, dx, dy, dt = flow_ref.scale(base, bdx, bdy, BASE_DT, 11., 1., 2.)
    uv = np.ascontiguousarray(fd[..., :2])
    auv = P.auv_params(noiseMagCoeffs=0.1, noiseMagActuation=0.1, cyl=cyl)
    for n in (1, 65, 257, 1000):
        rng = np.random.default_rng(50 + n)
        init = np.zeros((n, 16))
        init[:, :2] = (rng.random((n, 2)) - 0.5) * (1.5 if cyl else 0.5)
        init[:, 2] = rng.random(n) * 2 * np.pi
        init[:, 3] = rng.integers(0, 5, n) if cyl else rng.random(n) * 2 * np.pi
        init[:, 4] = rng.random(n) * 2.0
        init[:, 5:] = 1.0 + 0.05 - rng.random((n, 11)) * 0.1
        steps = 40
        actions = rng.uniform(-1, 1, size=(steps, n, 3)).astype(np.float32)
        h = _lib.Handle(P.make_config("auv", n, dt=0.02, auto_reset=False, max_steps=30, use_flow=True, auv=auv))
        h.set_flow(uv.astype(np.float32), dt, dx, dy)
        env = oracle_mod.OracleAuvEnv(n, "f64", dt=0.02, max_steps=30, flow=oracle_mod.FlowTable(uv, dt, dx, dy), auv=auv)
        o_gpu = h.reset(init=init.astype(np.float32))
        o_ref = env.reset(init.astype(np.float32).astype(np.float64))
        assert max_scaled_err(o_gpu, o_ref) < TOL
        alive = np.ones(n, bool)
        for k in range(steps):
            o_ref, r_ref, d_ref = env.step(actions[k].astype(np.float64))
            o_gpu, r_gpu, d_gpu = h.step(actions[k])
            st = h.get_state()
            pose = st[:6].T
            # a lane within fp32 resolution of the +-bounds / way-point threshold may legitimately fall on the other side
            near = alive & ((d_gpu != 0) != (d_ref != 0))
            assert near.sum() <= max(1, n // 200), (n, k, int(near.sum()))
            alive &= ~near
            a = alive
            assert np.array_equal((d_gpu[a] != 0), (d_ref[a] != 0))
            assert circ_err(pose[a], env.pose[a], [2]).max(initial=0.0) < TOL, (n, k)
            # the "V0" observation of AuvEnvCyl scales error CHANGES by 1/0.025 and 1/(2 deg): 40 x the fp32 resolution
            # of a pose near 2 (2.4e-7) and of an angle near 2 pi (4.8e-7), two roundings each
            assert np.max(np.abs(o_gpu[a] - o_ref[a]), initial=0.0) < (1.5e-4 if cyl else 3e-5), (n, k)
            assert np.max(np.abs(r_gpu[a] - r_ref[a]) / np.maximum(1.0, np.abs(r_ref[a])), initial=0.0) < 3e-5, (n, k)
            if cyl:
                iwp = st[P.STATE_PLANES[P.MODEL_AUV]["iwp"]].view(np.int32)
                switched = alive & (iwp != env.iwp)
                assert switched.sum() <= max(1, n // 200)
                alive &= ~switched
        assert alive.mean() > 0.98
        h.close()


# ---- env API semantics ---------------------------------------------------------------------------------
def test_auto_reset_terminal_obs_and_rng_shard_invariance():
    n, max_steps = 512, 5
    rng = np.random.default_rng(0)
    actions = rng.uniform(-1, 1, size=(12, n, 6)).astype(np.float32)

    def run(offset, count, seed=11):
        h = _lib.Handle(P.make_config("rov6", count, max_steps=max_steps, auto_reset=True, seed=seed, env_offset=offset,
                                      use_flow=False))
        out = [h.reset().copy()]
        dones, terms = [], []
        for s in range(12):
            o, r, d = h.step(actions[s, offset:offset + count])
            out.append(o.copy()); dones.append(d.copy()); terms.append(h.terminal_obs())
        st = h.get_state()
        h.close()
        return np.array(out), np.array(dones), np.array(terms), st

    full = run(0, n)
    # done exactly every max_steps steps; the returned obs on a done step is the first obs of the NEW episode
    assert np.array_equal(full[1].any(axis=1), [(s + 1) % max_steps == 0 for s in range(12)])
    assert full[1][4].all()
    ist = full[3][-1].view(np.int32)
    assert np.all(ist == 12 % max_steps)
    # new episode: state zero -> obs = clip(path / 3L) and the angle error of the new target
    o_new = full[0][5]
    assert np.all(np.abs(o_new) <= 1.0) and np.abs(o_new).sum() > 0
    assert not np.allclose(full[2][4], o_new)          # terminal obs differs from the reset obs
    # sharding: 2 handles of n/2 with env_offset reproduce the single handle bit for bit (RNG keyed by global id)
    a, b = run(0, n // 2), run(n // 2, n // 2)
    assert np.array_equal(np.concatenate([a[0], b[0]], axis=1), full[0])
    assert np.array_equal(np.concatenate([a[3], b[3]], axis=1), full[3])
    # a different seed gives different episodes
    other = run(0, n, seed=12)
    assert not np.array_equal(other[0][0], full[0][0])


def test_masked_reset_and_state_roundtrip():
    n = 256
    h = _lib.Handle(P.make_config("rov3", n, auto_reset=False, seed=3, use_flow=False))
    obs0 = h.reset().copy()
    a = np.random.default_rng(1).uniform(-1, 1, size=(n, 3)).astype(np.float32)
    for _ in range(3):
        h.step(a)
    st = h.get_state()
    mask = np.zeros(n, np.uint8)
    mask[::4] = 1
    sentinel = np.full((n, 5), 7.0, np.float32)
    obs = h.reset(mask=mask, obs_out=sentinel)
    st2 = h.get_state()
    assert np.all(obs[mask == 0] == 7.0)                 # untouched rows keep the caller's values
    assert np.array_equal(st2[:, mask == 0], st[:, mask == 0])
    assert np.all(st2[:6][:, mask == 1] == 0)            # systemState back to zero (3DoF.py:443)
    assert np.all(st2[-1].view(np.int32)[mask == 1] == 0)
    h.set_state(st)
    assert np.array_equal(h.get_state(), st)
    # a CHECKPOINT is the raw planes: the decoded default re-encodes the binary angles from fp32 radians after the intervening
    # get_state above (up to 2.4e-7 rad), the raw planes restore the bit patterns themselves - also on ANOTHER handle
    raw = h.get_state(raw=True)
    for _ in range(2):
        h.step(a)
    h.get_state()
    h.set_state(raw, raw=True)
    assert np.array_equal(h.get_state(raw=True).view(np.uint32), raw.view(np.uint32))
    h2 = _lib.Handle(P.make_config("rov3", n, auto_reset=False, seed=99, use_flow=False))
    h2.reset()
    h2.set_state(raw, raw=True)
    assert np.array_equal(h2.get_state(raw=True).view(np.uint32), raw.view(np.uint32))
    o1, o2 = h.step(a), h2.step(a)
    assert np.array_equal(o1[0], o2[0]) and np.array_equal(h.get_state(raw=True).view(np.uint32), h2.get_state(raw=True).view(np.uint32))
    h2.close()
    # a second step_async while one is pending is refused before its actions are staged over the pending step's
    h.set_state(raw, raw=True)
    h.step_async(a)
    with pytest.raises(_lib.MvrlError):
        h.step_async(-a)
    pending = h.step_wait()
    h.set_state(raw, raw=True)
    assert np.array_equal(pending[0], h.step(a)[0])
    with pytest.raises(_lib.MvrlError):
        h.lib  # noqa: B018
        _lib.check(h.lib.mvrl_step_wait(h.h, None, None, None), h.h)  # step_wait without step_async -> ESTATE
    h.close()


# ---- whole episodes ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("which,ceiling", [("c4", 0.49), ("c3", 0.05), ("c2", 0.006)])
def test_whole_episode_parity_not_below_the_fp32_storage_floor(oracle_mod, which, ceiling):
    """A full 250-step episode (6DoF.py:569-571) of the bench's population, 4096 envs, against the fp64 oracle: the share of envs that
    have left 1e-5 by the end.  Under random actions the loop is chaotic, so that share is large by then - the statement
    tested is RELATIVE: the kernel must lose no more envs than the same fp64 oracle whose state words are rounded to fp32 once per env
    step (what any implementation that keeps plain fp32 state between steps could reach at best; the binary angles are why the kernel
    does better), and stay under a ceiling 20 % above what 65 536 envs measured (profiles/r05_error_audit_episode.txt: 40.8 / 2.78 /
    0.18 %; round 4, fp32 turbulence sample time: 42.8 %; before the binary angles 72.7 / 7.1 / 0.37 %).  The mode that does follow the
    reference for whole episodes is precision = f64 (tests/test_gpu_f64.py::test_f64_whole_episode_follows_the_reference)."""
    from marinevehiclereinforcementlearning_amd.flow import ReconstructedFlow
    n, steps = 4096, 250
    dof = 3 if which == "c2" else 6
    npos = 3 if dof == 6 else 2
    h = _lib.Handle(P.make_config("rov6" if dof == 6 else "rov3", n, auto_reset=False, max_steps=10 ** 9, use_flow=which == "c4", seed=12345))
    ft = None
    if which == "c4":
        flow = ReconstructedFlow.synthetic(n_modes=8, n_time=2000)
        flow.scale(11., 1., 2., translate=(-1.65, -1.1))
        uv = flow.table_uv()
        h.set_flow(uv, flow.dt, flow.dx, flow.dy)
        ft = oracle_mod.FlowTable(uv.astype(np.float64), flow.dt, flow.dx, flow.dy)
    h.reset()
    st = h.get_state()
    init = np.concatenate([st[5 * dof:5 * dof + 2 * npos].T, st[4 * dof:5 * dof].T[:, npos:]], axis=1).astype(np.float64)
    toff = st[-2].copy()
    ref = oracle_mod.OracleRovEnv(dof, n, "f64", max_steps=10 ** 9, flow=ft)
    flo = oracle_mod.OracleRovEnv(dof, n, "f64", max_steps=10 ** 9, flow=ft)
    ref.reset(init, toffset=toff)
    flo.reset(init, toffset=toff)
    ang = [3, 4, 5] if dof == 6 else [2]
    rng = np.random.default_rng(2024)
    bad, bad_floor = np.zeros(n, bool), np.zeros(n, bool)
    for s in range(steps):
        a = rng.uniform(-1, 1, (n, dof)).astype(np.float32)
        ref.step(a.astype(np.float64))
        flo.step(a.astype(np.float64))
        for arr in (flo.y, flo.eold, flo.eint, flo.sp):
            arr[:] = arr.astype(np.float32)
        h.step(a)
        bad |= circ_err(h.get_state()[:2 * dof].T, ref.y, ang).max(axis=1) > TOL
        bad_floor |= circ_err(flo.y, ref.y, ang).max(axis=1) > TOL
    print(f"{which}: {100 * bad.mean():.2f} % of {n} envs beyond 1e-5 after {steps} steps; fp64 oracle with fp32 state storage: {100 * bad_floor.mean():.2f} %")
    assert np.isfinite(h.get_state()[:2 * dof]).all()
    assert bad.mean() <= bad_floor.mean() + 3.0 / np.sqrt(n) * np.sqrt(max(bad_floor.mean(), 1e-3)), (bad.mean(), bad_floor.mean())
    assert bad.mean() <= ceiling, bad.mean()
    h.close()
