"""fp64 host parameters (params.py / include/mvrl.h) -> the fp32 device structs of csrc/mvrl_device.hpp.

The C++ side (mvrl_abi.hip::to_dev) performs the same narrowing at mvrl_create; this Python mirror exists so that
tools/gen_baked.py can emit the default constants as compile-time literals, and tests can check both agree.
Field order == struct order in mvrl_device.hpp.
"""
import numpy as np


def _a(x):
    return np.array(list(x), dtype=np.float64)


def sym_layout(A, Ainv, tol=1e-9):
    """Detect the BlueROV2-Heavy sign-symmetric thruster layout; returns (sym_a[8], sym_ainv[8]) or None."""
    A = np.asarray(A, float).reshape(6, 8)
    Ai = np.asarray(Ainv, float).reshape(8, 6)
    sa = np.array([abs(A[0, 0]), abs(A[1, 0]), abs(A[2, 4]), abs(A[3, 0]), abs(A[3, 4]), abs(A[4, 0]), abs(A[4, 4]),
                   abs(A[5, 0])])
    sb = np.array([abs(Ai[0, 0]), abs(Ai[0, 1]), abs(Ai[0, 5]), abs(Ai[4, 0]), abs(Ai[4, 1]), abs(Ai[4, 2]),
                   abs(Ai[4, 3]), abs(Ai[4, 4])])
    pA, pB, pC = np.array([1, 1, -1, -1.]), np.array([-1, 1, -1, 1.]), np.array([-1, 1, 1, -1.])
    vB, vC = np.array([-1, -1, 1, 1.]), np.array([1, -1, 1, -1.])
    z = np.zeros(4)
    A_ref = np.array([np.r_[sa[0] * pA, z], np.r_[sa[1] * pB, z], np.r_[z, sa[2] * pC],
                      np.r_[-sa[3] * pB, sa[4] * vB], np.r_[sa[5] * pA, sa[6] * vC], np.r_[sa[7] * pC, z]])
    Ai_ref = np.zeros((8, 6))
    Ai_ref[:4, 0], Ai_ref[:4, 1], Ai_ref[:4, 5] = sb[0] * pA, sb[1] * pB, sb[2] * pC
    Ai_ref[4:, 0], Ai_ref[4:, 1], Ai_ref[4:, 2] = sb[3] * pB, sb[4] * vB, sb[5] * pC
    Ai_ref[4:, 3], Ai_ref[4:, 4] = sb[6] * vB, sb[7] * vC
    if np.max(np.abs(A - A_ref)) > tol * max(1.0, np.max(np.abs(A))):
        return None
    if np.max(np.abs(Ai - Ai_ref)) > tol * max(1.0, np.max(np.abs(Ai))):
        return None
    return sa, sb


def rov6_structured(p, tol=1e-12):
    """True when the constants have the structure the SYM kernel assumes (x_g=y_g=0, CB=0 in x,y, diagonal inertia,
    diagonal damping + the (4,2) entry, (u,q)/(v,p)-only mass coupling, sign-symmetric thrusters)."""
    cg, cb, I = _a(p.cg), _a(p.cb), _a(p.inertia).reshape(3, 3)
    if abs(cg[0]) > tol or abs(cg[1]) > tol or abs(cb[0]) > tol or abs(cb[1]) > tol:
        return False
    if np.max(np.abs(I - np.diag(np.diag(I)))) > tol:
        return False
    for M in (_a(p.dlin).reshape(6, 6), _a(p.dquad).reshape(6, 6)):
        M = M.copy()
        M[4, 2] = 0
        if np.max(np.abs(M - np.diag(np.diag(M)))) > tol:
            return False
    mi = _a(p.minv).reshape(6, 6).copy()
    for (i, j) in [(0, 0), (0, 4), (4, 0), (1, 1), (1, 3), (3, 1), (2, 2), (3, 3), (4, 4), (5, 5)]:
        mi[i, j] = 0
    if np.max(np.abs(mi)) > tol:
        return False
    return sym_layout(p.alloc, p.alloc_inv) is not None


def rov6_dev_fields(p):
    """[(field, value)] in the order of struct Rov6Dev."""
    cg, cb = _a(p.cg), _a(p.cb)
    W, B = p.weight, p.buoyancy
    sym = sym_layout(p.alloc, p.alloc_inv)
    sa, sb = sym if sym is not None else (np.zeros(8), np.zeros(8))
    k = p.thrust_k
    ad, I = _a(p.added), _a(p.inertia)
    sc = np.array([p.m * cg[2], p.m - ad[2], p.m - ad[1], p.m - ad[0], I[8] - I[4] + ad[4] - ad[5],
                   I[0] - I[8] + ad[5] - ad[3], I[4] - I[0] + ad[3] - ad[4], 0.0])
    return [("m", p.m), ("wb", W - B), ("cg", cg), ("I", _a(p.inertia)),
            ("gw", cg * W - cb * B), ("added", _a(p.added)), ("minv", _a(p.minv)), ("dlin", _a(p.dlin)),
            ("dquad", _a(p.dquad)), ("A", _a(p.alloc)), ("Ainv", _a(p.alloc_inv)), ("sym_a", sa), ("sym_ainv", sb), ("sym_c", sc),
            ("thrust_k", k), ("inv_thrust_k", 1.0 / k), ("rpm_max", p.rpm_max), ("rpm_dead", p.rpm_deadband),
            ("f_max", k * (p.rpm_max / 60.) ** 2), ("f_dead", k * (p.rpm_deadband / 60.) ** 2),
            ("kp", _a(p.kp)), ("ki", _a(p.ki)), ("kd", _a(p.kd)), ("windup", _a(p.windup)), ("umax", _a(p.umax)),
            ("act_scale", _a(p.act_scale)), ("inv_obs_pos", 1.0 / p.obs_pos_scale), ("inv_obs_ang", 1.0 / p.obs_ang_scale)]


def rov3_dev_fields(p):
    cg = _a(p.cg)
    k = p.thrust_k
    return [("m", p.m), ("cgx", cg[0]), ("cgy", cg[1]), ("xud", p.added[0]), ("yvd", p.added[1]),
            ("minv", _a(p.minv)), ("dlin", _a(p.dlin)), ("dquad", _a(p.dquad)), ("Ainv", _a(p.alloc_inv)),
            ("thrust_k", k), ("inv_thrust_k", 1.0 / k), ("rpm_max", p.rpm_max), ("rpm_dead", p.rpm_deadband),
            ("f_max", k * (p.rpm_max / 60.) ** 2), ("f_dead", k * (p.rpm_deadband / 60.) ** 2),
            ("cos_a", p.cos_alpha), ("sin_a", p.sin_alpha), ("yaw_arm", p.yaw_arm), ("inv_jet_area_k", 1.0 / p.jet_area_k),
            ("jet_c1", p.jet_c1), ("jet_k1", p.jet_k1), ("jet_c2", p.jet_c2), ("jet_k2", p.jet_k2),
            ("jet_drag_k", p.jet_drag_k),
            ("kp", _a(p.kp)), ("ki", _a(p.ki)), ("kd", _a(p.kd)), ("windup", _a(p.windup)), ("umax", _a(p.umax)),
            ("act_scale", _a(p.act_scale)), ("inv_obs_pos", 1.0 / p.obs_pos_scale), ("inv_obs_ang", 1.0 / p.obs_ang_scale)]
