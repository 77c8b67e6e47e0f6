"""Model constants of the three reference environments, derived in fp64 on the host and handed to
libmvrl.so as the POD structs of include/mvrl.h.

The numbers restate the literals of the reference's constructors:
  6-DoF vehicle  dynamicsModel_BlueROV2_Heavy_6DoF.py:83-218 ; PID :43-53
  3-DoF vehicle  dynamicsModel_BlueROV2_Heavy_3DoF.py:26-112 ; PID :141-154
  AuvEnv         tag_00_Dec2023_simpleControlTurbulence/verySimpleAuv.py:106-127
Derived matrices (A, pinv(A), M, inv(M), damping) are computed here exactly the way the reference
computes them (numpy pinv / the literal matrix layouts), so golden set g04 pins them.
"""
import ctypes as C

import numpy as np

MODEL_AUV, MODEL_ROV3, MODEL_ROV6 = 0, 1, 2
CTRL_FAITHFUL, CTRL_ZOH = 0, 1
ABI_VERSION = 4

MODEL_NAMES = {"auv": MODEL_AUV, "rov3": MODEL_ROV3, "rov6": MODEL_ROV6}
#            act, obs, init, state_words, aux
MAX_WAYPOINTS = 32
MODEL_DIMS = {MODEL_AUV: (3, 11, 16, 56, 11), MODEL_ROV3: (3, 5, 5, 24, 7), MODEL_ROV6: (6, 9, 9, 41, 14)}
# named planes of the SoA state returned by mvrl_get_state (first index / slice); integer planes are bit patterns
STATE_PLANES = {
    MODEL_ROV6: dict(y=slice(0, 12), eold=slice(12, 18), eint=slice(18, 24), setpoint=slice(24, 30), path=slice(30, 36),
                     episode=36, told=37, time=38, toffset=39, istep=40),
    MODEL_ROV3: dict(y=slice(0, 6), eold=slice(6, 9), eint=slice(9, 12), setpoint=slice(12, 15), path=slice(15, 19),
                     episode=19, told=20, time=21, toffset=22, istep=23),
    MODEL_AUV: dict(pose=slice(0, 6), heading_target=6, herr_o=7, perr_o=slice(8, 10), mult=slice(10, 21), toffset=21,
                    hist=slice(22, 52), istep=52, iwp=53, episode=54, phase=55),
}
# planes that hold BINARY ANGLES in an fp32 handle's state (bit pattern b of a uint32: angle = b * 2 pi / 2^32; include/mvrl.h)
ANGLE_PLANES = {MODEL_ROV6: (3, 4, 5), MODEL_ROV3: (2,), MODEL_AUV: ()}
PREC_F32, PREC_F64 = 0, 1
INTEG_RK4, INTEG_RK45 = 0, 1

d = C.c_double


class Rov6Params(C.Structure):
    _fields_ = [("m", d), ("length", d), ("cg", d * 3), ("cb", d * 3), ("inertia", d * 9), ("weight", d),
                ("buoyancy", d), ("added", d * 6), ("minv", d * 36), ("mass", d * 36), ("dlin", d * 36),
                ("dquad", d * 36), ("alloc", d * 48), ("alloc_inv", d * 48), ("thrust_k", d), ("rpm_max", d),
                ("rpm_deadband", d), ("kp", d * 6), ("ki", d * 6), ("kd", d * 6), ("windup", d * 6),
                ("umax", d * 6), ("act_scale", d * 6), ("obs_pos_scale", d), ("obs_ang_scale", d)]


class Rov3Params(C.Structure):
    _fields_ = [("m", d), ("length", d), ("cg", d * 3), ("izz", d), ("added", d * 3), ("minv", d * 9),
                ("mass", d * 9), ("dlin", d * 9), ("dquad", d * 9), ("alloc_inv", d * 12), ("thrust_k", d),
                ("rpm_max", d), ("rpm_deadband", d), ("cos_alpha", d), ("sin_alpha", d), ("yaw_arm", d),
                ("jet_area_k", d), ("jet_c1", d), ("jet_k1", d), ("jet_c2", d), ("jet_k2", d), ("jet_drag_k", d),
                ("kp", d * 3), ("ki", d * 3), ("kd", d * 3), ("windup", d * 3), ("umax", d * 3),
                ("act_scale", d * 3), ("obs_pos_scale", d), ("obs_ang_scale", d)]


class AuvParams(C.Structure):
    _fields_ = [("m", d), ("izz", d), ("xuu", d), ("yvv", d), ("nrr", d), ("xu", d), ("yv", d), ("nr", d),
                ("max_force", d), ("max_moment", d), ("x_min", d), ("x_max", d), ("y_min", d), ("y_max", d),
                ("noise_mag_coeffs", d), ("noise_mag_actuation", d), ("stop_on_bounds", C.c_int32),
                ("n_waypoints", C.c_int32), ("obs_scale", d * 9), ("wp_threshold", d), ("waypoints", d * (3 * 32))]


class JitReport(C.Structure):
    """mvrl_jit_report (include/mvrl.h): what mvrl_specialize built."""
    _fields_ = [("specialized", C.c_int32), ("min_waves_per_simd", C.c_int32), ("vgprs", C.c_int32), ("sgprs", C.c_int32),
                ("vgpr_spills", C.c_int32), ("sgpr_spills", C.c_int32), ("scratch_bytes", C.c_int32), ("lds_bytes", C.c_int32),
                ("code_bytes", C.c_int64), ("compiler", C.c_char * 16)]

    def as_dict(self):
        out = {k: getattr(self, k) for k, _ in self._fields_}
        out["compiler"] = self.compiler.decode()
        return out


class FlowDesc(C.Structure):
    _fields_ = [("n_t", C.c_int32), ("n_y", C.c_int32), ("n_x", C.c_int32), ("_pad", C.c_int32),
                ("dt", d), ("dx", d), ("dy", d)]


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("model", C.c_int32), ("device", C.c_int32), ("n_substeps", C.c_int32),
                ("n_envs", C.c_int64), ("env_offset", C.c_int64), ("dt", d), ("max_steps", C.c_int32),
                ("control_mode", C.c_int32), ("fixed_setpoint", C.c_int32), ("auto_reset", C.c_int32),
                ("seed", C.c_uint64), ("use_flow", C.c_int32), ("precision", C.c_int32), ("integrator", C.c_int32),
                ("_pad", C.c_int32),
                ("rov6", Rov6Params), ("rov3", Rov3Params), ("auv", AuvParams)]


def _fill(dst, arr):
    a = np.asarray(arr, dtype=np.float64).ravel()
    assert len(a) == len(dst), (len(a), len(dst))
    for i, v in enumerate(a):
        dst[i] = float(v)


def thrust_allocation(positions, normals, x0=None):
    """A[:, i] = [n_i ; (r_i - x0) x n_i],  Ainv = pinv(A)   (resources.py:19-35)."""
    positions = np.asarray(positions, dtype=np.float64)
    normals = np.asarray(normals, dtype=np.float64)
    if x0 is None:
        x0 = np.zeros(3)
    A = np.zeros((6, positions.shape[0]))
    for i in range(positions.shape[0]):
        A[:, i] = np.append(normals[i], np.cross(positions[i] - x0, normals[i]))
    return A, np.linalg.pinv(A)


def rov6_params(**overrides):
    """6-DoF BlueROV2 Heavy constants (6DoF.py:83-218) -> Rov6Params.  `overrides` may replace any of the
    reference's scalar attributes by name (e.g. Xuu=-20.0, m=12.0, CG=[0,0,0.04])."""
    g = dict(
        rho_f=1000., m=11.4, Length=0.457, CB=[0., 0., 0.], CG=[0., 0., 0.05], I=np.eye(3) * 0.16,
        Xudot=-5.5, Yvdot=-12.7, Zwdot=-14.57, Kpdot=-0.12, Mqdot=-0.12, Nrdot=-0.12,
        Yrdot=0., Zvdot=0., Nvdot=0.,
        Xuu=-18.18, Yvv=-21.66, Zww=-36.99, Kpp=-1.55, Mqq=-1.55, Nrr=-1.55, Yrr=0., Ypp=0., Zqq=0., Kvv=0.,
        Krr=0., Mww=-1.55, Nvv=0., Npp=0.,
        Xu=-4.03, Yv=-6.22, Zw=-5.18, Kp=-0.07, Mq=-0.07, Nr=-0.07, Yr=0., Yp=0., Zq=0., Kv=0., Kr=0., Mw=0.,
        Nv=0., Np=0.,
        D_thruster=0.1, alphaThruster=33. / 180. * np.pi, l_x=0.1475, l_y=0.101, l_z=0.068, l_x_v=0.120,
        l_y_v=0.22, l_z_v=0.0,
        K_P=[25., 25., 25., 10., 10., 1.], K_I=[2., 2., 2., 0.1, 0.1, 0.2], K_D=[20., 20., 20., 5., 5., 0.65],
        windup=[2., 2., 2., np.pi / 2, np.pi / 2, np.pi / 2], forceMomentMaxMagnitudes=[50., 50., 50., 1., 1., 2.],
        rpm_max=3500., rpm_deadband=300.)
    for k, v in overrides.items():
        if k not in g:
            raise KeyError(f"unknown 6-DoF parameter {k!r}")
        g[k] = v
    m, CG, CB, I = g["m"], np.asarray(g["CG"], float), np.asarray(g["CB"], float), np.asarray(g["I"], float)
    dispVol = m / g["rho_f"]
    Kt = 40. / (1000. * (3500. / 60.) ** 2. * g["D_thruster"] ** 4.)  # 6DoF.py:184
    ca, sa = np.cos(g["alphaThruster"]), np.sin(g["alphaThruster"])
    lx, ly, lz, lxv, lyv, lzv = g["l_x"], g["l_y"], g["l_z"], g["l_x_v"], g["l_y_v"], g["l_z_v"]
    pos = np.array([[lx, ly, lz], [lx, -ly, lz], [-lx, ly, lz], [-lx, -ly, lz],
                    [lxv, lyv, lzv], [lxv, -lyv, lzv], [-lxv, lyv, lzv], [-lxv, -lyv, lzv]])  # :193-202
    nrm = np.array([[ca, -sa, 0.], [ca, sa, 0.], [-ca, -sa, 0.], [-ca, sa, 0.],
                    [0., 0., -1.], [0., 0., 1.], [0., 0., 1.], [0., 0., -1.]])               # :203-212
    A, Ainv = thrust_allocation(pos, nrm)
    Mrb = np.array([
        [m, 0., 0., 0., m * CG[2], -m * CG[1]],
        [0., m, 0., -m * CG[2], 0., m * CG[0]],
        [0., 0., m, m * CG[1], -m * CG[0], 0.],
        [0., -m * CG[2], m * CG[1], 0., 0., 0.],
        [m * CG[2], 0., -m * CG[0], 0., 0., 0.],
        [-m * CG[1], m * CG[0], 0., 0., 0., 0.]])                                              # :286-293
    Mrb[3:, 3:] = I
    # NOTE the reference puts Zvdot (=0), not Zwdot, on the heave diagonal (6DoF.py:297) - kept.
    Ma = -1. * np.diag([g["Xudot"], g["Yvdot"], g["Zvdot"], g["Kpdot"], g["Mqdot"], g["Nrdot"]])
    M = Mrb + Ma
    Dl = -1. * np.array([
        [g["Xu"], 0., 0., 0., 0., 0.],
        [0., g["Yv"], 0., g["Yp"], 0., g["Yr"]],
        [0., 0., g["Zw"], 0., g["Zq"], 0.],
        [0., g["Kv"], 0., g["Kp"], 0., g["Kr"]],
        [0., 0., g["Mw"], 0., g["Mq"], 0.],
        [0., g["Nv"], 0., g["Np"], 0., g["Nr"]]])                                              # :345-352
    Dq = -1. * np.array([
        [g["Xuu"], 0., 0., 0., 0., 0.],
        [0., g["Yvv"], 0., g["Ypp"], 0., g["Yrr"]],
        [0., 0., g["Zww"], 0., g["Zqq"], 0.],
        [0., g["Kvv"], 0., g["Kpp"], 0., g["Krr"]],
        [0., 0., g["Mww"], 0., g["Mqq"], 0.],
        [0., g["Nvv"], 0., g["Npp"], 0., g["Nrr"]]])                                           # :354-361
    p = Rov6Params()
    p.m, p.length = m, g["Length"]
    _fill(p.cg, CG); _fill(p.cb, CB); _fill(p.inertia, I)
    p.weight = m * 9.81
    p.buoyancy = dispVol * g["rho_f"] * 9.81
    _fill(p.added, [g["Xudot"], g["Yvdot"], g["Zwdot"], g["Kpdot"], g["Mqdot"], g["Nrdot"]])
    _fill(p.minv, np.linalg.inv(M)); _fill(p.mass, M); _fill(p.dlin, Dl); _fill(p.dquad, Dq)
    _fill(p.alloc, A); _fill(p.alloc_inv, Ainv)
    p.thrust_k = g["rho_f"] * g["D_thruster"] ** 4. * Kt
    p.rpm_max, p.rpm_deadband = g["rpm_max"], g["rpm_deadband"]
    _fill(p.kp, g["K_P"]); _fill(p.ki, g["K_I"]); _fill(p.kd, g["K_D"]); _fill(p.windup, g["windup"])
    _fill(p.umax, g["forceMomentMaxMagnitudes"])
    L = g["Length"]
    _fill(p.act_scale, [2. * L, 2. * L, 2. * L, 45. / 180. * np.pi, 45. / 180. * np.pi, 45. / 180. * np.pi])
    p.obs_pos_scale = L * 3.
    p.obs_ang_scale = 45. / 180. * np.pi
    return p


def rov3_params(**overrides):
    """3-DoF BlueROV2 Heavy constants (3DoF.py:26-126, :141-154) -> Rov3Params."""
    g = dict(rho_f=1000., m=11.4, Length=0.457, CG=[0., 0., 0.02], Izz=0.16,
             Xudot=-5.5, Yvdot=-12.7, Nrdot=-0.12,
             Xuu=-18.18, Yvv=-21.66, Yrr=0., Nvv=0., Nrr=-1.55,
             Xu=-4.03, Yv=-6.22, Yr=0., Nv=0., Nr=-0.07,
             D_thruster=0.1, alphaThruster=45. / 180. * np.pi, l_x=0.156, l_y=0.111,
             K_P=[20., 20., 20.], K_I=[0.1, 0.1, 0.1], K_D=[5., 5., 0.5], windup=[2., 2., np.pi / 2],
             umax=[150., 150., 100.], rpm_max=3500., rpm_deadband=300.)
    for k, v in overrides.items():
        if k not in g:
            raise KeyError(f"unknown 3-DoF parameter {k!r}")
        g[k] = v
    m, CG, L = g["m"], np.asarray(g["CG"], float), g["Length"]
    dispVol = m / g["rho_f"]
    Kt = 40. / (1000. * (3500. / 60.) ** 2. * g["D_thruster"] ** 4.)
    al = g["alphaThruster"]
    A = np.array([[1., 1., -1., -1.], [1., -1., 1., -1.], [1., 1., 1., 1.]])
    A[0, :] = A[0, :] * np.cos(al)
    A[1, :] = A[1, :] * np.sin(al)
    A[2, :] = A[2, :] * np.sin(al) * L / 2.
    Ainv = np.linalg.pinv(A)                                                                  # 3DoF.py:104-112
    Mrb = np.array([[m, 0., -m * CG[1]], [0., m, m * CG[0]], [-m * CG[1], m * CG[0], g["Izz"]]])
    M = Mrb + -1. * np.diag([g["Xudot"], g["Yvdot"], g["Nrdot"]])
    Dl = -1. * np.array([[g["Xu"], 0., 0.], [0., g["Yv"], g["Yr"]], [0., g["Nv"], g["Nr"]]])
    Dq = -1. * np.array([[g["Xuu"], 0., 0.], [0., g["Yvv"], g["Yrr"]], [0., g["Nvv"], g["Nrr"]]])
    p = Rov3Params()
    p.m, p.length, p.izz = m, L, g["Izz"]
    _fill(p.cg, CG)
    _fill(p.added, [g["Xudot"], g["Yvdot"], g["Nrdot"]])
    _fill(p.minv, np.linalg.inv(M)); _fill(p.mass, M); _fill(p.dlin, Dl); _fill(p.dquad, Dq)
    _fill(p.alloc_inv, Ainv)
    p.thrust_k = g["rho_f"] * g["D_thruster"] ** 4. * Kt
    p.rpm_max, p.rpm_deadband = g["rpm_max"], g["rpm_deadband"]
    p.cos_alpha, p.sin_alpha = np.cos(al), np.sin(al)
    p.yaw_arm = np.sqrt(g["l_x"] ** 2. + g["l_y"] ** 2.)
    p.jet_area_k = 0.5 * g["rho_f"] * np.pi * g["D_thruster"] ** 2
    p.jet_c1, p.jet_k1, p.jet_c2, p.jet_k2 = 0.56599, 7.60891, 0.05654, 0.89679
    p.jet_drag_k = 0.5 * g["rho_f"] * dispVol ** (2. / 3.)
    _fill(p.kp, g["K_P"]); _fill(p.ki, g["K_I"]); _fill(p.kd, g["K_D"]); _fill(p.windup, g["windup"])
    _fill(p.umax, g["umax"])
    _fill(p.act_scale, [2. * L, 2. * L, 45. / 180. * np.pi])
    p.obs_pos_scale = L * 3.
    p.obs_ang_scale = 45. / 180. * np.pi
    return p


def cylinder_waypoints(Rcyl=1.33, xCyl=(2.5, 0.)):
    """The 21 way-points round the cylinder (tag/verySimpleAuv_cyl.py:29-40): [x, y, target heading]."""
    Rwp = Rcyl * 1.3
    t = np.linspace(-30, 30, 21) * np.pi / 180.
    x = -Rwp * np.cos(t) + xCyl[0]
    y = Rwp * np.sin(t) + xCyl[1]
    return np.vstack([x, y, -t]).T, Rcyl * 0.05


def auv_params(noiseMagCoeffs=0.0, noiseMagActuation=0.0, stopOnBoundsExceeded=True,
               xMinMax=(-1., 1.), yMinMax=(-1., 1.), cyl=False):
    """AuvEnv constants (tag/verySimpleAuv.py:106-127); cyl=True: AuvEnvCyl (tag/verySimpleAuv_cyl.py:29-111 -
    way-points, +-2 m bounds, "V0" observation scaling)."""
    p = AuvParams()
    deg = np.pi / 180.
    if cyl:
        xMinMax, yMinMax = (-2., 2.), (-2., 2.)
        wps, thr = cylinder_waypoints()
        p.n_waypoints = len(wps)
        p.wp_threshold = thr
        _fill(p.waypoints, np.concatenate([wps.ravel(), np.zeros(3 * MAX_WAYPOINTS - wps.size)]))
        _fill(p.obs_scale, [1 / 0.2, 1 / 0.2, 1 / (45. * deg), 1 / (2. * deg), 1 / 0.025, 1 / 0.025, 1 / 0.2, 1 / 0.2,
                            1 / (30. * deg)])
    else:
        p.n_waypoints = 0
        _fill(p.obs_scale, [1., 1., 1 / (45. * deg), 1., 1., 1., 1., 1., 1.])
    p.m, p.izz = 11.4, 0.16
    p.xuu, p.yvv, p.nrr = -18.18 * 2.21, -21.66 * 4.87, -1.55
    p.xu, p.yv, p.nr = -4.03 * 2.21, -6.22 * 4.87, -0.07
    p.max_force, p.max_moment = 150., 20.
    p.x_min, p.x_max = float(xMinMax[0]), float(xMinMax[1])
    p.y_min, p.y_max = float(yMinMax[0]), float(yMinMax[1])
    p.noise_mag_coeffs, p.noise_mag_actuation = float(noiseMagCoeffs), float(noiseMagActuation)
    p.stop_on_bounds = 1 if stopOnBoundsExceeded else 0
    return p


def make_config(model, n_envs, *, dt=None, n_substeps=4, max_steps=250, control_mode=CTRL_FAITHFUL,
                fixed_setpoint=False, auto_reset=True, seed=0, use_flow=None, device=0, env_offset=0,
                rov6=None, rov3=None, auv=None, precision="f32", integrator="rk4"):
    if isinstance(model, str):
        model = MODEL_NAMES[model]
    cfg = Config()
    cfg.abi_version = ABI_VERSION
    cfg.model = model
    cfg.device = device
    cfg.n_substeps = n_substeps
    cfg.n_envs = n_envs
    cfg.env_offset = env_offset
    cfg.dt = (0.02 if model == MODEL_AUV else 0.2) if dt is None else dt
    cfg.max_steps = max_steps
    cfg.control_mode = control_mode
    cfg.fixed_setpoint = 1 if fixed_setpoint else 0
    cfg.auto_reset = 1 if auto_reset else 0
    cfg.seed = seed
    cfg.use_flow = int(model == MODEL_AUV) if use_flow is None else int(bool(use_flow))
    cfg.precision = {"f32": PREC_F32, "f64": PREC_F64}[precision] if isinstance(precision, str) else int(precision)
    cfg.integrator = {"rk4": INTEG_RK4, "rk45": INTEG_RK45}[integrator] if isinstance(integrator, str) else int(integrator)
    cfg.rov6 = rov6 if rov6 is not None else rov6_params()
    cfg.rov3 = rov3 if rov3 is not None else rov3_params()
    cfg.auv = auv if auv is not None else auv_params()
    return cfg
