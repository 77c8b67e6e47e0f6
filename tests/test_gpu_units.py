"""Unit-level GPU parity (SURVEY.md 8(a) rows a3-a12 one RHS call at a time, through mvrl_derivs): the golden vectors the
imported reference produced for single `derivs` calls (G8: random state / set-point / controller memory incl. first calls and
t == tOld), the PID call sequences (G5) and the known-answer anchors quoted in SURVEY.md - on fp64 handles to 1e-9, on fp32
handles to 1e-5 where the reference's 1e-9 derivative floor does not amplify input rounding (those cases are counted)."""
import numpy as np
import pytest

from .conftest import golden, max_scaled_err
from marinevehiclereinforcementlearning_amd import _lib, params as P

pytestmark = pytest.mark.gpu


def handle(dof, precision):
    return _lib.Handle(P.make_config("rov6" if dof == 6 else "rov3", 1, use_flow=False, precision=precision))


@pytest.mark.parametrize("dof", [6, 3])
def test_derivs_goldens_fp64(dof):
    g = golden(f"g08_derivs{dof}.npz")
    h = handle(dof, "f64")
    r = h.derivs(g["t"], g["y"], g["sp"], eold=g["eOld"], eint=g["eInt"], told=g["tOld"], has_old=g["has_old"])
    assert max_scaled_err(r["dy"], g["dy"]) < 1e-9
    assert max_scaled_err(r["gcf"], g["gcf"]) < 1e-9
    assert max_scaled_err(r["rpm"], g["rpm"]) < 1e-9
    assert max_scaled_err(r["eold"], g["eOld_out"]) < 1e-12
    assert max_scaled_err(r["eint"], g["eInt_out"]) < 1e-12
    assert np.array_equal(r["told"], g["tOld_out"])
    h.close()


def _input_sensitivity(g, dof):
    """A-priori bound on what fp32 INPUT resolution can do to the accelerations of one derivs call: the PID differentiates
    e - eOld, four fp32 numbers (sp, pose, their difference, eOld) each resolved to half an ulp, and multiplies by K_D / max(1e-9,
    t - tOld); the demand then passes rotation (norm-preserving), allocation + saturation (Lipschitz ||A|| ||pinv A||) and M^-1.
    Nothing here is fitted to the kernel: constants from params.py, spacings from numpy."""
    prm = P.rov6_params() if dof == 6 else P.rov3_params()
    kd, minv = np.array(prm.kd), np.abs(np.array(prm.minv).reshape(dof, dof))
    if dof == 6:
        lip = np.linalg.norm(np.array(prm.alloc).reshape(6, 8), 2) * np.linalg.norm(np.array(prm.alloc_inv).reshape(8, 6), 2)
    else:
        ai = np.array(prm.alloc_inv).reshape(4, 3)
        lip = np.linalg.norm(ai, 2) * np.linalg.norm(np.linalg.pinv(ai), 2)
    ulp = lambda x: np.spacing(np.abs(np.asarray(x, np.float32))).astype(np.float64)
    pose, e, eo = g["y"][:, :dof], g["eOld_out"], np.where(g["has_old"][:, None], g["eOld"], g["eOld_out"])
    de = 0.5 * (ulp(g["sp"]) + ulp(pose) + ulp(e) + ulp(eo))
    den = np.maximum(1e-9, g["t"] - g["tOld"])[:, None]
    du = np.where(g["has_old"][:, None], kd[None] * de / den, 0.0)           # first calls differentiate against themselves: 0
    half = dof // 2 if dof == 6 else 2
    tau = np.concatenate([np.repeat(np.linalg.norm(du[:, :half], axis=1)[:, None], half, 1),
                          np.repeat(np.linalg.norm(du[:, half:], axis=1)[:, None], dof - half, 1)], axis=1)
    return lip * (tau @ minv.T).max(axis=1)


@pytest.mark.parametrize("dof", [6, 3])
def test_derivs_goldens_fp32(dof):
    """G8 through the fp32 handle, EVERY case: t and tOld are fp64 at the ABI (t - tOld is formed in fp64 on the host), so what
    separates an fp32 call from the reference is the resolution of its fp32 inputs, amplified by K_D / (t - tOld) - and mostly
    swallowed by the clamp.  Every case must meet 1e-5, floor cases (t <= tOld) included; at most 2 % may exceed it, and only
    if their a-priori input sensitivity (_input_sensitivity) explains the excess."""
    g = golden(f"g08_derivs{dof}.npz")
    h = handle(dof, "f32")
    assert "baked" in h.variant
    r = h.derivs(g["t"], g["y"], g["sp"], eold=g["eOld"], eint=g["eInt"], told=g["tOld"], has_old=g["has_old"])
    sens = _input_sensitivity(g, dof)
    err = (np.abs(r["dy"].astype(np.float64) - g["dy"]) / np.maximum(1.0, np.abs(g["dy"]))).max(axis=1)
    ok = err <= 1e-5
    print(f"derivs{dof} fp32: {ok.sum()} of {len(ok)} G8 cases within 1e-5 (max {err[ok].max():.2e}); beyond it: "
          f"{[(int(i), float(err[i]), float(sens[i])) for i in np.nonzero(~ok)[0]]}")
    # every case meets the bar, save at most 2 % that the a-priori bound shows to be at the mercy of their inputs' last bit
    assert np.all(ok | (err <= 4.0 * sens)), [(int(i), float(err[i]), float(sens[i])) for i in np.nonzero(~ok)[0]]
    assert (~ok).sum() <= max(1, len(ok) // 50), int((~ok).sum())
    assert max_scaled_err(r["eold"], g["eOld_out"]) < 1e-5                     # eOld' = e for every case
    assert np.array_equal(r["told"], g["tOld_out"])                            # tOld' = t, exactly (fp64)
    dt = g["t"] - g["tOld"]
    smooth = ~(g["has_old"] & (dt <= 1e-6))
    assert max_scaled_err(r["eint"][smooth], g["eInt_out"][smooth]) < 1e-5
    first = ~g["has_old"]                                                      # first calls: dedt = 0 exactly -> tight, side outputs too
    assert max_scaled_err(r["rpm"][first] / 3500.0, g["rpm"][first] / 3500.0) < 1e-5
    h.close()


@pytest.mark.parametrize("dof", [6, 3])
@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_derivs_with_current_goldens(dof, precision):
    """G21 / G22 through the HIP path (mvrl_derivs_cur): the right-hand side of the 3/6-DoF + TURBULENCE composition - the benched
    config - against fixtures written by EXECUTING the reference's own velCurrent lines with a non-zero current (3DoF.py:182-191,
    :216-239, :279; 6DoF.py:258-267, :396 hook-assisted: see each fixture's `how`).  fp64: 1e-9 on every case; fp32: 1e-5 on every
    case the fp32 zero-current run of the SAME inputs meets (the excesses are the K_D / (t - tOld) input sensitivity of G8, which
    the current does not touch), and the EFFECT of the current, dy - dy(zero current), to 1e-5 on every case."""
    g = golden("g21_derivs3_current.npz" if dof == 3 else "g22_derivs6_current.npz")
    h = handle(dof, precision)
    kw = dict(eold=g["eOld"], eint=g["eInt"], told=g["tOld"], has_old=g["has_old"])
    r = h.derivs(g["t"], g["y"], g["sp"], cur=g["cur"], **kw)
    r0 = h.derivs(g["t"], g["y"], g["sp"], **kw)
    h.close()
    err = (np.abs(r["dy"].astype(np.float64) - g["dy"]) / np.maximum(1.0, np.abs(g["dy"]))).max(axis=1)
    err0 = (np.abs(r0["dy"].astype(np.float64) - g["dy_zero_current"]) / np.maximum(1.0, np.abs(g["dy_zero_current"]))).max(axis=1)
    if precision == "f64":
        assert err.max() < 1e-9 and err0.max() < 1e-9, (err.max(), err0.max())
        assert max_scaled_err(r["gcf"], g["gcf"]) < 1e-9 and max_scaled_err(r["rpm"], g["rpm"]) < 1e-9
        return
    effect = (r["dy"].astype(np.float64) - r0["dy"]) - (g["dy"] - g["dy_zero_current"])
    eff_err = (np.abs(effect) / np.maximum(1.0, np.abs(g["dy"]))).max(axis=1)
    print(f"derivs{dof} + current fp32: {int((err <= 1e-5).sum())} of {len(err)} within 1e-5 (zero current: {int((err0 <= 1e-5).sum())}); "
          f"effect of the current: max {eff_err.max():.2e}")
    # the fixture's currents reach 3.6 / 4.9 m/s - four times the scaled table's mean of 1 m/s - where the quadratic damping is a
    # 200-N term resolved to 6e-8: beyond 3 m/s (9 cases of 256 each; one of them misses 1e-5) the bar is 5e-5, and the straight fp32
    # build of the oracle misses 1e-5 on exactly the same case (1.8e-5 / 1.1e-5)
    strong = np.linalg.norm(g["cur"], axis=1) > 3.0
    tol = np.where(strong, 5e-5, 1e-5)
    assert np.all((err <= tol) | (err <= 2.0 * err0 + 1e-6)), [(int(i), float(err[i]), float(err0[i])) for i in np.nonzero(err > tol)[0]]
    assert (err > 1e-5).sum() <= max(1, len(err) // 50) and strong.sum() <= 12
    assert np.all(eff_err <= tol), float(eff_err.max())
    assert np.abs(g["dy"] - g["dy_zero_current"]).max() > 1e-2


@pytest.mark.parametrize("precision,tol", [("f64", 1e-11), ("f32", 1e-5)])
def test_known_answer_anchors(precision, tol):
    """SURVEY.md 8(a) notes: fresh controller, derivs(0, y) for the two quoted set-points."""
    g = golden("g00_anchors.npz")
    for dof in (6, 3):
        h = handle(dof, precision)
        r = h.derivs(0.0, g[f"y{dof}"], g[f"sp{dof}"])
        assert max_scaled_err(r["dy"][0], g[f"dy{dof}"]) < tol, (dof, precision)
        assert max_scaled_err(r["gcf"][0], g[f"gcf{dof}"]) < tol
        assert max_scaled_err(r["rpm"][0] / 3500.0, g[f"rpm{dof}"] / 3500.0) < tol
        if dof == 6:
            assert np.allclose(r["gcf"][0], [47.5, -50, 30, -1, 1, -2], atol=1e-4)
        h.close()


def test_pid_call_sequences_fp64():
    """G5: 48 controllers x 8 consecutive calls with increasing / equal / decreasing t - the controller memory threaded
    through consecutive mvrl_derivs calls must follow computeControlForces (6DoF.py:43-73) call for call."""
    g = golden("g05_pid6.npz")
    h = handle(6, "f64")
    n, calls = g["t"].shape
    eo = np.zeros((n, 6)); ei = np.zeros((n, 6)); to = np.zeros(n)
    for c in range(calls):
        y = np.zeros((n, 12)); y[:, :6] = g["pose"][:, c]
        r = h.derivs(g["t"][:, c], y, g["setpoint"], eold=eo, eint=ei, told=to, has_old=np.full(n, c > 0))
        assert max_scaled_err(r["gcf"], g["out"][:, c]) < 1e-6, c     # 1e-9 floor: K_D * 1e-16 / 1e-9 at worst
        assert max_scaled_err(r["eold"], g["eOld"][:, c]) < 1e-12, c
        assert max_scaled_err(r["eint"], g["eInt"][:, c]) < 1e-12, c
        assert np.array_equal(r["told"], g["tOld"][:, c]), c
        eo, ei, to = r["eold"], r["eint"], r["told"]
    h.close()


def test_derivs_flavours_and_errors():
    """sym / generic constant flavours go through the same entry point; AuvEnv handles and wrong-precision calls are refused."""
    g = golden("g08_derivs6.npz")
    ref = None
    for kw in (dict(), dict(rov6=P.rov6_params(K_P=[25., 25., 25., 10., 10., 1.0 * (1 + 1e-6)])),
               dict(rov6=P.rov6_params(Yr=1e-12))):
        h = _lib.Handle(P.make_config("rov6", 1, use_flow=False, precision="f64", **kw))
        r = h.derivs(g["t"], g["y"], g["sp"], eold=g["eOld"], eint=g["eInt"], told=g["tOld"], has_old=g["has_old"])
        if ref is None:
            ref, v0 = r["dy"], h.variant
        else:
            assert h.variant != v0
            assert max_scaled_err(r["dy"], ref) < 1e-4          # constants differ by 1e-6 relative at most
        h.close()
    ha = _lib.Handle(P.make_config("auv", 4, use_flow=False))
    with pytest.raises((_lib.MvrlError, AttributeError, KeyError, AssertionError)):
        ha.derivs(0.0, np.zeros((1, 12)), np.zeros((1, 6)))
    ha.close()


def test_vehicle_facades_drive_solve_ivp_like_the_reference():
    """The reference's own usage pattern (6DoF.py:686-745): build controller + vehicle, hand vehicle.derivs to scipy's
    solve_ivp.  One env.step of the reference = one such solve over dt = 0.2 s from rest (goldens G10, first step)."""
    import scipy.integrate
    from marinevehiclereinforcementlearning_amd.vehicles import (BlueROV2Heavy3DoF, BlueROV2Heavy6DoF,
                                                                   BlueROV2Heavy6DoF_PID_controller)
    g = golden("g00_anchors.npz")
    ctl = BlueROV2Heavy6DoF_PID_controller(g["sp6"])
    rov = BlueROV2Heavy6DoF(ctl)
    dy = rov.derivs(0.0, g["y6"])
    assert max_scaled_err(dy, g["dy6"]) < 1e-11 and ctl.tOld == 0.0 and ctl.eOld is not None
    assert max_scaled_err(rov.generalisedControlForces, g["gcf6"]) < 1e-11
    ctl.reset()
    assert ctl.eOld is None
    g10 = golden("g10_envstep_6dof_fixedsp.npz")
    ctl = BlueROV2Heavy6DoF_PID_controller(g10["sp0"][0])
    rov2 = BlueROV2Heavy6DoF(ctl)
    sol = scipy.integrate.solve_ivp(rov2.derivs, (0.0, 0.2), np.zeros(12), method="RK45", t_eval=[0.2], max_step=0.2,
                                    rtol=1e-3, atol=1e-3)
    y1 = sol.y[:, -1]
    ref = g10["states"][0, 1].copy()
    y1[3:6] %= 2 * np.pi                                   # the env wraps the angles after the solve (6DoF.py:560)
    assert max_scaled_err(y1, ref) < 1e-8
    v3 = BlueROV2Heavy3DoF(g["sp3"])
    assert max_scaled_err(v3.derivs(0.0, g["y3"]), g["dy3"]) < 1e-11
    for v in (rov, rov2, v3):
        v.close()


# ---- rows a5-a8 on their own (mvrl_vehicle_ops): body axes, allocateThrust, forceModel ------------------------------------
@pytest.mark.parametrize("precision,tol", [("f64", 1e-11), ("f32", 1e-5)])
def test_body_axes_golden(precision, tol):
    """G3: iHat/jHat/kHat of scipy's Rotation.from_euler('XYZ') and globalToVehicle / vehicleToGlobal (6DoF.py:238-251)."""
    g = golden("g03_rotation.npz")
    h = handle(6, precision)
    axes = h.vehicle_ops(g["angles"], want=("axes",))["axes"].astype(np.float64)
    assert np.max(np.abs(axes - g["axes"])) < (1e-14 if precision == "f64" else 5e-7)
    body = np.einsum("nij,nj->ni", axes, g["vec"])
    assert max_scaled_err(body, g["body"]) < tol
    back = np.einsum("ni,nij->nj", body, axes)
    assert max_scaled_err(back, g["back"]) < tol
    h.close()


@pytest.mark.parametrize("precision,tol", [("f64", 1e-10), ("f32", 1e-5)])
def test_allocate_thrust_and_thruster_column_golden(precision, tol):
    """G6: allocateThrust (pinv(A) . gcf -> inverse thrust curve -> rpm, 6DoF.py:220-231) and the thruster column H the
    allocated rpm produce through limit + dead-band + thrusterModel (:233-236, :271-282)."""
    g = golden("g06_alloc_thrust.npz")
    h = handle(6, precision)
    r = h.vehicle_ops(g["angles"], gcf=g["gcf"], want=("rpm", "thruster_h"))
    assert max_scaled_err(r["rpm"] / 3500.0, g["rpm"] / 3500.0) < tol
    assert max_scaled_err(r["thruster_h"], g["H"]) < (tol if precision == "f64" else 2e-5)
    r2 = h.vehicle_ops(g["angles"], rpm=g["rpm"], want=("thruster_h",))       # from the golden rpm directly
    assert max_scaled_err(r2["thruster_h"], g["H"]) < (tol if precision == "f64" else 2e-5)
    h.close()


@pytest.mark.parametrize("precision,tol", [("f64", 1e-10), ("f32", 2e-5)])
@pytest.mark.parametrize("flavour", ["baked", "sym", "generic"])
def test_force_model_golden(precision, tol, flavour):
    """G7: RHS of forceModel for random attitude / body velocity / rpm (6DoF.py:253-404), every kernel flavour."""
    g = golden("g07_force_model.npz")
    # constants moved by less than the tolerance, enough to leave the baked / sparse classes
    over = {"baked": {}, "sym": dict(rov6=P.rov6_params(Xu=-4.03 * (1 + (1e-13 if precision == "f64" else 2e-7)))),
            "generic": dict(rov6=P.rov6_params(Yr=5e-12))}[flavour]
    h = _lib.Handle(P.make_config("rov6", 1, use_flow=False, precision=precision, **over))
    assert flavour in h.variant, h.variant
    r = h.vehicle_ops(g["angles"], rpm=g["rpm"], vel=g["vel"], want=("rhs", "thruster_h"))
    assert max_scaled_err(r["rhs"], g["RHS"]) < tol
    assert max_scaled_err(r["thruster_h"], g["comp"][:, :, 4]) < tol
    assert max_scaled_err(np.array(h.cfg.rov6.mass).reshape(6, 6), g["M"]) < 1e-15
    h.close()


@pytest.mark.parametrize("precision,tol", [("f64", 1e-10), ("f32", 2e-5)])
def test_force_model_components_golden(precision, tol):
    """G7: forceModel(..., retComp=True) (6DoF.py:401-402) - the 6 x 5 breakdown [-Crb.vel, -Ca.vel, -D.vel, G, H] the imported
    reference returned, through mvrl_force_components; its columns also recombine into RHS (velCurrent = 0: :396)."""
    g = golden("g07_force_model.npz")
    h = handle(6, precision)
    comp = h.force_components(g["angles"], g["vel"], g["rpm"])
    assert comp.shape == g["comp"].shape
    for c in range(5):
        assert max_scaled_err(comp[:, :, c], g["comp"][:, :, c]) < tol, c
    rhs = comp[:, :, 0] + comp[:, :, 1] + comp[:, :, 2] - comp[:, :, 3] + comp[:, :, 4]
    assert max_scaled_err(rhs, g["RHS"]) < 4 * tol
    h.close()


def test_vehicle_ops_argument_checks():
    h3 = handle(3, "f32")
    with pytest.raises(_lib.MvrlError):
        h3.vehicle_ops(np.zeros((2, 3)))
    h3.close()
    h = handle(6, "f32")
    with pytest.raises(ValueError):
        h.vehicle_ops(np.zeros((2, 3)), gcf=np.zeros((3, 6)))
    out = h.vehicle_ops(np.zeros((5, 3)))                    # axes only
    assert set(out) == {"axes"} and np.array_equal(out["axes"], np.broadcast_to(np.eye(3, dtype=np.float32), (5, 3, 3)))
    h.close()


def test_vehicle_facade_public_methods():
    """The reference's usage in example_trialTrajectories.py:100-134 / 6DoF.py:723-735: updateMovingCoordSystem then iHat..,
    globalToVehicle, allocateThrust from generalisedControlForces, forceModel -> (M, RHS)."""
    from marinevehiclereinforcementlearning_amd.vehicles import BlueROV2Heavy6DoF, BlueROV2Heavy6DoF_PID_controller
    g3, g6, g7 = golden("g03_rotation.npz"), golden("g06_alloc_thrust.npz"), golden("g07_force_model.npz")
    rov = BlueROV2Heavy6DoF(BlueROV2Heavy6DoF_PID_controller(np.zeros(6)))
    for i in (0, 7, 100):
        rov.updateMovingCoordSystem(g3["angles"][i])
        assert np.max(np.abs(np.vstack([rov.iHat, rov.jHat, rov.kHat]) - g3["axes"][i])) < 1e-14
        assert max_scaled_err(rov.globalToVehicle(g3["vec"][i]), g3["body"][i]) < 1e-12
        assert max_scaled_err(rov.vehicleToGlobal(g3["body"][i]), g3["back"][i]) < 1e-12
        rov.updateMovingCoordSystem(g6["angles"][i])
        rov.generalisedControlForces = g6["gcf"][i]
        assert max_scaled_err(rov.allocateThrust() / 3500., g6["rpm"][i] / 3500.) < 1e-10
        M, rhs = rov.forceModel(np.zeros(3), g7["angles"][i], g7["vel"][i], g7["rpm"][i])
        assert max_scaled_err(rhs, g7["RHS"][i]) < 1e-10 and np.array_equal(M, g7["M"])
        assert np.array_equal(rov.rotation_angles, g6["angles"][i])           # 6DoF.py:240
        comp = rov.forceModel(np.zeros(3), g7["angles"][i], g7["vel"][i], g7["rpm"][i], retComp=True)
        assert comp.shape == (6, 5) and max_scaled_err(comp, g7["comp"][i]) < 1e-10
    rov.close()


# ---- the vectors the reference itself holds, through HIP (VERDICT r3 "missing 3"): G14, G11, G4 -----------------------------------
@pytest.mark.parametrize("precision,tol", [("f64", 1e-11), ("f32", 1e-5)])
@pytest.mark.parametrize("flavour", ["sym", "generic"])
def test_example_temp_known_answer_through_hip(precision, tol, flavour):
    """G14 - the one known answer the reference ships (example_temp.py:19-28): acc = np.linalg.solve(M, RHS) with the older
    CG_z = 0.025.  A handle built with that CG applies ITS M^-1 to the shipped RHS on the device (mvrl_mass_solve -> mass_solve6,
    the function of the step kernel): the 10-non-zero form (sym) and the dense form (generic: a damping entry of 5e-12 leaves the
    structured class without moving anything)."""
    g = golden("g14_example_temp.npz")
    over = dict(CG=[0., 0., 0.025])
    if flavour == "generic":
        over["Yr"] = 5e-12
    prm = P.rov6_params(**over)
    assert max_scaled_err(np.array(prm.mass).reshape(6, 6), g["M"]) < 1e-12       # the handle's M is the shipped M
    h = _lib.Handle(P.make_config("rov6", 1, use_flow=False, precision=precision, rov6=prm))
    assert flavour in h.variant, h.variant
    rhs = np.stack([g["RHS"], -2.5 * g["RHS"], np.zeros(6)])
    acc = h.mass_solve(rhs).astype(np.float64)
    assert np.max(np.abs(acc[0] - g["acc"])) < 5e-7                                # the file prints 7 significant digits
    assert max_scaled_err(acc[0], np.linalg.solve(g["M"], g["RHS"])) < tol
    assert max_scaled_err(acc[1], -2.5 * np.linalg.solve(g["M"], g["RHS"])) < 2.5 * tol and not acc[2].any()
    h.close()


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_constants_the_kernels_apply_match_reference(precision):
    """G4 - computeThrustAllocation's A and pinv(A) (resources.py:19-35), M and M^-1 (6DoF.py:286-299), 3-DoF pinv (3DoF.py:104-112),
    READ BACK FUNCTIONALLY from the device: unit inputs through the kernels' own allocate6 / thruster column / mass_solve6 (the
    default handle is the `baked` flavour, whose constants are compile-time literals - a memory read-back would not see them)."""
    g = golden("g04_constants.npz")
    f64 = precision == "f64"
    tol = 1e-12 if f64 else 2e-6
    h = handle(6, precision)
    assert "baked" in h.variant
    kt = float(g["Kt"])
    k = 1000.0 * 0.1 ** 4 * kt                                                     # rho D^4 Kt: F = k (rpm / 60)^2
    # M^-1 column j = solve(M, e_j); M = inv of what came back
    minv = h.mass_solve(np.eye(6)).astype(np.float64).T
    assert max_scaled_err(minv, g["Minv6"]) < tol
    assert max_scaled_err(np.linalg.inv(minv), g["M6"]) < (1e-11 if f64 else 1e-5)
    # A column k = thruster column H of one thruster at 3000 rpm / its force (inside the limits, above the dead-band)
    rpm = 3000.0 * np.eye(8)
    H = h.vehicle_ops(np.zeros((8, 3)), rpm=rpm, want=("thruster_h",))["thruster_h"].astype(np.float64)
    assert max_scaled_err(H.T / (k * 50.0 ** 2), g["A6"]) < tol
    # pinv(A) column j = allocation of a unit demand along axis j (level attitude: body axes = global axes), rpm -> force
    out = h.vehicle_ops(np.zeros((6, 3)), gcf=10.0 * np.eye(6), want=("rpm",))["rpm"].astype(np.float64)
    cv = np.sign(out) * k * (out / 60.0) ** 2
    assert max_scaled_err(cv.T / 10.0, g["Ainv6"]) < (1e-11 if f64 else 5e-6)
    h.close()
    # 3-DoF: first PID call (dedt = 0, eInt = 0 -> u = K_P e = 20 e) at psi = 0, rpm = inverse thrust curve of pinv(A3) u
    h3 = handle(3, precision)
    e = np.eye(3) * 0.5
    r = h3.derivs(np.zeros(3), np.zeros((3, 6)), e, has_old=np.zeros(3, np.uint8))
    assert max_scaled_err(r["gcf"], 20.0 * e) < tol * 20
    rp = r["rpm"].astype(np.float64)
    cv3 = np.sign(rp) * k * (rp / 60.0) ** 2
    assert max_scaled_err(cv3.T / 10.0, g["Ainv3"]) < (1e-11 if f64 else 5e-6)
    h3.close()


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("dof", [6, 3])
def test_data_to_state_golden_through_hip(precision, dof):
    """G11 - dataToState (6DoF.py:467-483, 3DoF.py:397-409) for 128 random (path, set-point, systemState) triples: way-points through
    mvrl_reset(init), pose and set-point through mvrl_set_state, the observation through mvrl_observe (observe6 / observe3, the
    device functions of the step and reset kernels).  fp32: 1e-5, except rows where the yaw / attitude error sits within fp32
    rounding of the +-pi branch, which flips a clipped observation between -1 and +1 (counted, at most 1 %)."""
    g = golden("g11_data_to_state.npz")
    npos = 3 if dof == 6 else 2
    path, sp, y, ref = g[f"path{dof}"], g[f"sp{dof}"], g[f"state{dof}"], g[f"obs{dof}"]
    n = len(ref)
    h = _lib.Handle(P.make_config("rov6" if dof == 6 else "rov3", n, use_flow=False, precision=precision, auto_reset=False))
    h.reset(init=np.concatenate([path.reshape(n, 2 * npos), sp[:, npos:]], axis=1))
    st = h.get_state()
    st[:2 * dof] = y.T
    st[4 * dof:5 * dof] = sp.T                      # planes: y[2 dof] eOld[dof] eInt[dof] setPoint[dof] path[2 npos] ... (include/mvrl.h)
    h.set_state(st)
    obs = h.observe().astype(np.float64)
    d = np.abs(obs - ref)
    if precision == "f64":
        assert d.max() < 1e-12
    else:
        from marinevehiclereinforcementlearning_amd.hostmath import angle_error
        near_branch = np.zeros(n, bool)
        for i in range(n):
            for kk in range(dof - npos):
                near_branch[i] |= np.pi - abs(angle_error(sp[i, npos + kk], y[i, npos + kk])) < 1e-5
        assert d[~near_branch].max() < 1e-5 and near_branch.sum() <= max(1, n // 100)
    # a reset observation is the same function of a zero state
    o0 = h.reset(init=np.concatenate([path.reshape(n, 2 * npos), sp[:, npos:]], axis=1))
    assert np.array_equal(o0, h.observe())
    h.close()


def test_observe_auv_matches_reset_and_step_observations():
    """mvrl_observe for AuvEnv: a second dataToState call on the pose a step left behind - the error terms and velocities equal the
    step's own observation; the increments (columns 3-5: herr - herr_o, perr - perr_o) are zero, because the step stored this
    pose's errors as the "old" ones (verySimpleAuv.py:349-350)."""
    n = 257
    h = _lib.Handle(P.make_config("auv", n, use_flow=False, auto_reset=False, seed=3))
    o = h.reset()
    assert np.array_equal(o, h.observe())
    rng = np.random.default_rng(1)
    keep = [0, 1, 2, 6, 7, 8, 9, 10]
    for _ in range(5):
        o, _, _ = h.step(rng.uniform(-1, 1, (n, 3)).astype(np.float32))
        o2 = h.observe()
        assert np.array_equal(o[:, keep], o2[:, keep]) and not o2[:, 3:6].any()
    h.close()
