"""The reference's edge cases for the kinematic transform and the angle error, driven THROUGH THE HIP PATH (mvrl_derivs):

  a10  resources.coordinateTransform (resources.py:98-143): columns of J read back as eta_dot = J(eta) nu for unit nu,
       on the 404 attitude triples of golden G2 - incl. the |cos theta| < 1e-6 guard cases (:116-120), where entries of J
       reach 1e6;
  a11  resources.angleError (resources.py:75-95): the yaw error the PID stores (eOld' = e) for the 1100 pairs of golden G1 -
       incl. differences of exactly 0, +-pi (tie -> -pi), +-2 pi and multiples.

fp64 handles against the goldens of the imported reference (1e-9 scaled for J, 1e-12 for the angle error); fp32 handles
against the fp64 oracle evaluated at the SAME fp32-rounded inputs (the guard decides on cos(theta) of the value the kernel
actually receives; theta = pi/2 +- 1e-7 is not representable in fp32), 1e-5 scaled."""
import numpy as np
import pytest

from .conftest import golden, max_scaled_err
from marinevehiclereinforcementlearning_amd import _lib, params as P

pytestmark = pytest.mark.gpu


def _columns_of_J(h, angles):
    """J[:, :, k] for every attitude triple through vehicle.derivs: y = [0, 0, 0, phi, theta, psi, e_k], dy[:6] = J e_k."""
    n = len(angles)
    y = np.zeros((n, 6, 12))
    y[:, :, 3:6] = angles[:, None, :]
    y[:, np.arange(6), 6 + np.arange(6)] = 1.0
    r = h.derivs(0.0, y.reshape(n * 6, 12), np.zeros((n * 6, 6)))
    return np.transpose(r["dy"][:, :6].reshape(n, 6, 6), (0, 2, 1))      # [n, row, column k]


def test_coordinate_transform_guard_cases_fp64():
    g = golden("g02_coord_transform.npz")
    h = _lib.Handle(P.make_config("rov6", 1, use_flow=False, precision="f64"))
    J = _columns_of_J(h, g["angles"])
    guard = np.abs(np.cos(g["angles"][:, 1])) < 1e-6
    assert guard.sum() >= 8 and np.abs(g["J"][guard]).max() > 1e5        # the fixture really holds guard cases
    assert max_scaled_err(J, g["J"]) < 1e-9
    assert max_scaled_err(J[guard], g["J"][guard]) < 1e-9
    h.close()


def test_coordinate_transform_guard_cases_fp32(oracle_mod):
    g = golden("g02_coord_transform.npz")
    a32 = g["angles"].astype(np.float32)
    # three kernel flavours share the J code; run the baked one (default constants) and the sym one
    p6 = P.rov6_params()
    p6.kp[0] = 25.0 * (1 + 1e-7)
    for cfg in (P.make_config("rov6", 1, use_flow=False), P.make_config("rov6", 1, use_flow=False, rov6=p6)):
        h = _lib.Handle(cfg)
        J = _columns_of_J(h, a32)
        ref = oracle_mod.Oracle("f64").coord_transform6(a32.astype(np.float64))
        guard = np.abs(np.cos(a32[:, 1].astype(np.float64))) < 1e-6
        assert guard.sum() >= 4
        # away from the guard: fp32 rounding of sin/cos, amplified by 1/cos(theta) (up to 1e3 in the fixture)
        far = np.abs(np.cos(a32[:, 1].astype(np.float64))) > 1e-3
        assert max_scaled_err(J[far], ref[far]) < 1e-5 * 50
        plain = np.abs(np.cos(a32[:, 1].astype(np.float64))) > 0.2
        assert max_scaled_err(J[plain], ref[plain]) < 1e-5
        # inside the guard the divisor is exactly +-1e-6: entries are sin/cos products times 1e6, relative error = fp32 trig
        assert max_scaled_err(J[guard], ref[guard]) < 1e-5, np.abs(J[guard] - ref[guard]).max()
        h.close()


def _yaw_error_through_pid(h, psi_d, psi):
    n = len(psi)
    y = np.zeros((n, 12)); sp = np.zeros((n, 6))
    y[:, 5] = psi; sp[:, 5] = psi_d
    r = h.derivs(0.0, y, sp)           # first call: eOld' = e = [sp - pose (5), angleError(sp_psi, psi)]
    return r["eold"][:, 5]


def test_angle_error_ties_fp64():
    g = golden("g01_angle_error.npz")
    d = g["psi_d"] - g["psi"]
    ties = np.isclose(np.abs(np.mod(d, 2 * np.pi) - np.pi), 0, atol=1e-12) | np.isclose(np.mod(d, 2 * np.pi), 0, atol=1e-12)
    assert ties.sum() >= 10                                              # 0, +-pi, +-2 pi, ... are in the fixture
    h = _lib.Handle(P.make_config("rov6", 1, use_flow=False, precision="f64"))
    out = _yaw_error_through_pid(h, g["psi_d"], g["psi"])
    assert np.max(np.abs(out - g["out"])) < 1e-12
    exact_pi = np.abs(np.abs(g["out"]) - np.pi) < 1e-15
    assert exact_pi.sum() >= 2 and np.all(out[exact_pi] < 0)            # the tie resolves to -pi, as the reference's
    h.close()
    h3 = _lib.Handle(P.make_config("rov3", 1, use_flow=False, precision="f64"))
    y = np.zeros((len(d), 6)); sp = np.zeros((len(d), 3))
    y[:, 2] = g["psi"]; sp[:, 2] = g["psi_d"]
    assert np.max(np.abs(h3.derivs(0.0, y, sp)["eold"][:, 2] - g["out"])) < 1e-12
    h3.close()


def test_angle_error_ties_fp32(oracle_mod):
    g = golden("g01_angle_error.npz")
    pd32, ps32 = g["psi_d"].astype(np.float32), g["psi"].astype(np.float32)
    h = _lib.Handle(P.make_config("rov6", 1, use_flow=False))
    out = _yaw_error_through_pid(h, pd32, ps32).astype(np.float64)
    # the reference's formula on the fp32 difference the kernel forms (d = fl32(psi_d - psi)), in fp64
    d = (pd32 - ps32).astype(np.float64)
    ref = oracle_mod.Oracle("f64").angle_error(d, np.zeros_like(d))
    err = np.abs(out - ref)
    err = np.minimum(err, np.abs(err - 2 * np.pi))                        # -pi and +pi are the same heading
    assert err.max() < 1e-5, err.max()
    assert np.all(out >= -np.pi - 1e-6) and np.all(out < np.pi + 1e-6)
    zero = d == 0
    assert zero.sum() >= 1 and np.all(out[zero] == 0)
    h.close()


def test_turbulence_lookup_at_and_beyond_the_table_edges(oracle_mod):
    """The step kernels read the turbulence table through re-packed stencil cells (mvrl_set_flow: one 64-byte cell per
    (t, y, x) holding the 2 x 2 x 2 stencil).  Lookups with the index clamped at either end and the weight NOT clamped
    (flowGenerator.py:108-117: linear extrapolation) must read the same eight values as the plain table: AuvEnv envs placed
    below, inside, on the last cell of and far beyond the grid, at time offsets before the first, at the last and beyond the
    last snapshot; stop_on_bounds off so that they keep stepping.  fp32 and fp64 handles, host and device table entry points."""
    import torch
    rng = np.random.default_rng(17)
    n_t, n_y, n_x, dt, dx, dy = 12, 9, 11, 0.05, 0.25, 0.3
    uv = rng.normal(0, 0.3, (n_t, n_y, n_x, 2))
    xs = np.array([-0.6, -1e-3, 0.0, 0.1, (n_x - 2) * dx, (n_x - 2) * dx + 0.2, (n_x - 1) * dx, (n_x - 1) * dx + 0.7])
    ys = np.array([-0.5, 0.0, 0.7, (n_y - 2) * dy + 0.1, (n_y - 1) * dy, (n_y - 1) * dy + 0.8])
    toffs = np.array([0.0, 0.26, (n_t - 2) * dt - 0.02, (n_t - 1) * dt - 0.02, (n_t - 1) * dt + 0.2])
    grid = np.array([(x, y, t) for x in xs for y in ys for t in toffs])
    n = len(grid)
    init = np.zeros((n, 16))
    init[:, 0], init[:, 1], init[:, 4] = grid[:, 0], grid[:, 1], grid[:, 2]
    init[:, 2] = rng.random(n) * 2 * np.pi
    init[:, 3] = rng.random(n) * 2 * np.pi
    init[:, 5:] = 1.0
    auv = P.auv_params(stopOnBoundsExceeded=False)
    acts = rng.uniform(-1, 1, (3, n, 3))
    # fp32: extrapolation weights of 3 - 5 in each of the three directions multiply the rounding of the interpolation
    for precision, tol in (("f64", 1e-10), ("f32", 2e-4)):
        dtype = np.float64 if precision == "f64" else np.float32
        env = oracle_mod.OracleAuvEnv(n, "f64", dt=0.02, max_steps=10 ** 6, flow=oracle_mod.FlowTable(uv, dt, dx, dy), auv=auv)
        env.reset(init.astype(dtype).astype(np.float64))
        ref = [tuple(np.copy(a) for a in env.step(acts[k].astype(dtype).astype(np.float64))) + (env.pose.copy(), env.aux.copy()) for k in range(3)]
        for entry in ("host", "device"):
            h = _lib.Handle(P.make_config("auv", n, dt=0.02, auto_reset=False, max_steps=10 ** 6, use_flow=True, auv=auv, precision=precision))
            if entry == "host":
                h.set_flow(uv.astype(dtype), dt, dx, dy)
            else:
                tbl = torch.as_tensor(uv.astype(dtype), device="cuda").contiguous()
                h.set_flow_dev(tbl.data_ptr(), n_t, n_y, n_x, dt, dx, dy)
                del tbl                                        # the handle keeps its own re-packed copy
                torch.cuda.empty_cache()
            h.reset(init=init.astype(dtype))
            for k in range(3):
                o, r, d = h.step(acts[k].astype(dtype))
                o_ref, r_ref, d_ref, pose_ref, aux_ref = ref[k]
                # velCurrent (u, v) is among the step's side outputs: the looked-up value itself
                assert max_scaled_err(h.get_state()[:6].T, pose_ref) < tol, (precision, entry, k)
                assert max_scaled_err(o, o_ref) < max(tol, 1e-9) and max_scaled_err(r, r_ref) < max(tol * 3, 1e-9), (precision, entry, k)
            h.close()


# ---- binary angles (ABI 3): the Euler-angle planes of the fp32 rigid-body handles -------------------------------------------------
@pytest.mark.parametrize("dof", [6, 3])
def test_binary_angle_planes_round_trip_and_wrap(oracle_mod, dof):
    """The angle words of an fp32 handle are uint32 binary angles (include/mvrl.h): 0 after reset; Handle.get_state decodes them to
    fp32 radians in [0, 2 pi) and set_state encodes (2.4e-7 rad, the resolution of the fp32 number handed in); unchanged entries are
    restored bit for bit; and a vehicle turning THROUGH 0 / 2 pi - the wrap of 6DoF.py:560 / 3DoF.py:480, here an integer overflow - stays
    on the fp64 oracle's trajectory on the circle, set-point and observation included."""
    npos = 3 if dof == 6 else 2
    planes = list(P.ANGLE_PLANES[P.MODEL_ROV6 if dof == 6 else P.MODEL_ROV3])
    n = 256
    model = "rov6" if dof == 6 else "rov3"
    h = _lib.Handle(P.make_config(model, n, auto_reset=False, max_steps=10 ** 9, use_flow=False))
    rng = np.random.default_rng(4)
    init = np.concatenate([(rng.random((n, 2 * npos)) - 0.5) * 2.0, rng.random((n, dof - npos)) * 2 * np.pi], axis=1).astype(np.float32)
    h.reset(init=init)
    raw = h.get_state(raw=True)
    assert not raw[planes].view(np.uint32).any()                                   # pose angles 0 after reset: bit pattern 0
    # encode / decode: headings just below 2 pi, just above 0, and everywhere else
    ang = (rng.random((len(planes), n)) * 2 * np.pi).astype(np.float32)
    ang[:, :8] = np.float32(2 * np.pi) - np.float32([1e-6, 5e-7, 1e-4, 1e-3, 1e-2, 0.1, 0.2, 0.3])
    ang[:, 8:12] = np.float32([0.0, 1e-7, 1e-4, 1e-2])
    st = h.get_state()
    st[planes] = ang
    h.set_state(st)
    back = h.get_state()
    d = np.abs(back[planes].astype(np.float64) - ang)
    d = np.minimum(d, 2 * np.pi - d)
    assert d.max() < 5e-7 and np.all(back[planes] >= 0) and np.all(back[planes] < np.float32(2 * np.pi))
    bits = h.get_state(raw=True)[planes].view(np.uint32).copy()
    assert np.allclose(bits.astype(np.float64) * (2 * np.pi / 2 ** 32), np.mod(ang.astype(np.float64), 2 * np.pi), atol=8e-10, rtol=0)
    h.set_state(back)                                                              # unchanged entries: the same bits go back
    assert np.array_equal(h.get_state(raw=True)[planes].view(np.uint32), bits)
    st2 = back.copy()
    st2[-2] += 0.25                                                                # edit ANOTHER plane (flow time offset): angles keep their bits
    h.set_state(st2)
    assert np.array_equal(h.get_state(raw=True)[planes].view(np.uint32), bits)
    # through the wrap: every env starts 0.02 rad either side of 0 / 2 pi in every angle and is steered across it
    start = np.where(rng.random((n, dof - npos)) < 0.5, 2 * np.pi - 0.02, 0.02) + (rng.random((n, dof - npos)) - 0.5) * 0.01
    st = h.get_state()
    st[planes] = start.T.astype(np.float32)
    h.set_state(st)
    st = h.get_state()                                                             # what the handle holds now (decoded), as the oracle's start
    env = oracle_mod.OracleRovEnv(dof, n, "f64", max_steps=10 ** 9)
    env.reset(init.astype(np.float64))
    env.y[:, npos:dof] = st[planes].T.astype(np.float64)
    # exact start angles for the oracle: the bits, not their fp32 decoding
    env.y[:, npos:dof] = (h.get_state(raw=True)[planes].view(np.uint32).astype(np.float64) * (2 * np.pi / 2 ** 32)).T
    crossed = np.zeros(n, bool)
    for k in range(12):
        a = rng.uniform(-1, 1, (n, dof)).astype(np.float32)
        a[:, npos:] = np.where(start > np.pi, 1.0, -1.0) * np.abs(a[:, npos:])       # towards and across the wrap
        o_ref, _, _ = env.step(a.astype(np.float64))
        o_gpu, _, _ = h.step(a)
        got = h.get_state()
        y = got[: 2 * dof].T.astype(np.float64)
        d = np.abs(y - env.y)
        d[:, npos:dof] = np.minimum(d[:, npos:dof], 2 * np.pi - d[:, npos:dof])
        err = (d / np.maximum(1.0, np.abs(env.y))).max(axis=1)
        crossed |= (np.abs(env.y[:, npos:dof] - start) > np.pi).any(axis=1)
        ok = err < 1e-5
        assert ok.mean() > 0.98, (k, float(np.sort(err)[-5:].min()))
        assert np.all(got[planes] >= 0) and np.all(got[planes] < np.float32(2 * np.pi))
        assert np.max(np.abs(o_gpu[ok] - o_ref[ok])) < 2e-5
        sp_gpu = got[4 * dof:5 * dof].T.astype(np.float64)                         # set-point planes: a * scale + angle in [0, 2 pi)
        assert max_scaled_err(sp_gpu[ok], env.sp[ok]) < 1e-5
    assert crossed.mean() > 0.5                                                    # the wrap really was exercised
    h.close()
