"""ctypes front-end of the CPU oracle (oracle/mvrl_oracle.c).

TEST INFRASTRUCTURE ONLY - the checker, never the product.  Importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; the package
`marinevehiclereinforcementlearning_amd` never imports this module.

`build()` compiles the library with gcc (oracle/Makefile).  `Oracle(precision)` exposes each restated
reference function on numpy arrays; `OracleRovEnv` / `OracleAuvEnv` hold batched per-env state and mirror
the reference's reset()/step() semantics for N environments.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from marinevehiclereinforcementlearning_amd import params as P  # noqa: E402  (POD structs + constants only)

# MVRL_ORACLE_LIB: another build of the same restatement (a CPU-only checking build, tests/sanitize/)
LIB_PATH = os.environ.get("MVRL_ORACLE_LIB") or os.path.join(HERE, "_build", "libmvrl_oracle.so")
TWO_PI = 2.0 * np.pi


def build(force=False):
    src = os.path.join(HERE, "mvrl_oracle.c")
    hdr = os.path.join(REPO, "include", "mvrl.h")
    if os.environ.get("MVRL_ORACLE_LIB"):
        return LIB_PATH
    if (not force and os.path.exists(LIB_PATH)
            and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return LIB_PATH
    subprocess.check_call(["make", "-C", HERE, "-s"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        build()
        _LIB = C.CDLL(LIB_PATH)
    return _LIB


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype)) if a is not None else None


class Oracle:
    """One precision ('f64' or 'f32') of the restatement."""

    def __init__(self, precision="f64", rov6=None, rov3=None, auv=None):
        assert precision in ("f64", "f32")
        self.suf = "_" + precision
        self.dtype = np.float64 if precision == "f64" else np.float32
        self.creal = C.c_double if precision == "f64" else C.c_float
        self.rov6 = rov6 if rov6 is not None else P.rov6_params()
        self.rov3 = rov3 if rov3 is not None else P.rov3_params()
        self.auv = auv if auv is not None else P.auv_params()
        self.L = lib()
        r = self.creal

        class Pid(C.Structure):
            _fields_ = [("eold", r * 6), ("eint", r * 6), ("told", C.c_double), ("has_old", C.c_int32)]
        self.Pid = Pid
        f = self._f("orc_angle_error")
        f.restype = r
        f.argtypes = [r, r]

    def _f(self, name):
        return getattr(self.L, name + self.suf)

    def arr(self, x, shape=None):
        a = np.ascontiguousarray(x, dtype=self.dtype)
        if shape is not None:
            a = a.reshape(shape)
        return a

    def rp(self, a):
        return _ptr(a, self.creal)

    # ---- element functions --------------------------------------------------------------------
    def angle_error(self, psi_d, psi):
        f = self._f("orc_angle_error")
        pd, ps = np.broadcast_arrays(np.asarray(psi_d, dtype=self.dtype), np.asarray(psi, dtype=self.dtype))
        out = np.array([f(self.creal(a), self.creal(b)) for a, b in zip(pd.ravel(), ps.ravel())], dtype=self.dtype)
        return out.reshape(pd.shape)

    def body_axes(self, ang):
        ang = self.arr(ang, (-1, 3))
        out = np.zeros((len(ang), 3, 3), dtype=self.dtype)
        f = self._f("orc_body_axes")
        for i in range(len(ang)):
            f(self.rp(ang[i]), self.rp(out[i]))
        return out

    def coord_transform6(self, ang):
        ang = self.arr(ang, (-1, 3))
        out = np.zeros((len(ang), 6, 6), dtype=self.dtype)
        f = self._f("orc_coord_transform6")
        f.argtypes = [self.creal] * 3 + [C.POINTER(self.creal)]
        for i in range(len(ang)):
            f(ang[i, 0], ang[i, 1], ang[i, 2], self.rp(out[i]))
        return out

    def make_pid(self, eold=None, eint=None, told=0.0):
        s = self.Pid()
        n = 0 if eold is None else len(eold)
        for i in range(n):
            s.eold[i] = float(eold[i])
        if eint is not None:
            for i in range(len(eint)):
                s.eint[i] = float(eint[i])
        s.told = float(told)
        s.has_old = 0 if eold is None else 1
        return s

    def pid6(self, sp, pose, t, pid):
        sp, pose = self.arr(sp), self.arr(pose)
        out = np.zeros(6, dtype=self.dtype)
        f = self._f("orc_pid6")
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
        f(C.addressof(self.rov6), sp.ctypes.data, pose.ctypes.data, float(t), C.addressof(pid), out.ctypes.data)
        return out

    def alloc6(self, ang, gcf):
        axes = self.body_axes(ang)[0]
        gcf = self.arr(gcf)
        rpm = np.zeros(8, dtype=self.dtype)
        f = self._f("orc_alloc6")
        f.argtypes = [C.c_void_p] * 4
        f(C.addressof(self.rov6), axes.ctypes.data, gcf.ctypes.data, rpm.ctypes.data)
        return rpm

    def force_model6(self, ang, vel, rpm, cur_body=None):
        ang, vel, rpm = self.arr(ang), self.arr(vel), self.arr(rpm)
        rhs = np.zeros(6, dtype=self.dtype)
        comp = np.zeros((6, 5), dtype=self.dtype)
        cb = None if cur_body is None else self.arr(cur_body)
        f = self._f("orc_force_model6")
        f.argtypes = [C.c_void_p] * 7
        f(C.addressof(self.rov6), ang.ctypes.data, vel.ctypes.data, rpm.ctypes.data,
          None if cb is None else cb.ctypes.data, rhs.ctypes.data, comp.ctypes.data)
        return rhs, comp

    def derivs(self, dof, t, y, sp, pid, cur=None):
        y, sp = self.arr(y), self.arr(sp)
        dy = np.zeros(2 * dof, dtype=self.dtype)
        gcf = np.zeros(dof, dtype=self.dtype)
        rpm = np.zeros(8 if dof == 6 else 4, dtype=self.dtype)
        cu = None if cur is None else self.arr(cur)
        f = self._f("orc_derivs6" if dof == 6 else "orc_derivs3")
        f.argtypes = [C.c_void_p, C.c_double] + [C.c_void_p] * 7
        prm = self.rov6 if dof == 6 else self.rov3
        f(C.addressof(prm), float(t), y.ctypes.data, sp.ctypes.data, C.addressof(pid),
          None if cu is None else cu.ctypes.data, dy.ctypes.data, gcf.ctypes.data, rpm.ctypes.data)
        return dy, gcf, rpm

    def obs_rov(self, dof, y, path, sp):
        y, path, sp = self.arr(y), self.arr(path), self.arr(sp)
        obs = np.zeros(9 if dof == 6 else 5, dtype=self.dtype)
        f = self._f("orc_obs6" if dof == 6 else "orc_obs3")
        f.argtypes = [C.c_void_p] * 5
        prm = self.rov6 if dof == 6 else self.rov3
        f(C.addressof(prm), y.ctypes.data, path.ctypes.data, sp.ctypes.data, obs.ctypes.data)
        return obs

    def flow_interp(self, table, dt, dx, dy, t, x, y):
        table = self.arr(table)
        n_t, n_y, n_x, n_comp = table.shape
        t, x, y = self.arr(t).ravel(), self.arr(x).ravel(), self.arr(y).ravel()
        out = np.zeros((len(t), n_comp), dtype=self.dtype)
        f = self._f("orc_flow_interp_batch")
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        f(table.ctypes.data, n_t, n_y, n_x, n_comp, float(dt), float(dx), float(dy), t.ctypes.data, x.ctypes.data,
          y.ctypes.data, len(t), out.ctypes.data)
        return out


class FlowTable:
    """(u, v) table + spacings as the oracle / product consume it."""

    def __init__(self, table_uv, dt, dx, dy):
        self.table = table_uv
        self.dt, self.dx, self.dy = float(dt), float(dx), float(dy)


class OracleRovEnv:
    """N independent 3/6-DoF environments stepped by the oracle (reset()/step() as the reference's env)."""

    def __init__(self, dof, n, precision="f64", dt=0.2, n_substeps=4, integrator="rk4", control_mode=P.CTRL_FAITHFUL,
                 fixed_setpoint=False, max_steps=250, flow=None, rov6=None, rov3=None):
        self.o = Oracle(precision, rov6=rov6, rov3=rov3)
        self.dof, self.n, self.dt = dof, n, dt
        self.n_sub, self.integrator, self.control_mode = n_substeps, integrator, control_mode
        self.fixed_sp, self.max_steps = fixed_setpoint, max_steps
        self.npos = 3 if dof == 6 else 2
        self.nobs = 9 if dof == 6 else 5
        self.nthr = 8 if dof == 6 else 4
        dt_ = self.o.dtype
        self.y = np.zeros((n, 2 * dof), dt_)
        self.sp = np.zeros((n, dof), dt_)
        self.path = np.zeros((n, 2 * self.npos), dt_)
        self.eold = np.zeros((n, dof), dt_)
        self.eint = np.zeros((n, dof), dt_)
        self.told = np.zeros(n, np.float64)
        self.has_old = np.zeros(n, np.int32)
        self.istep = np.zeros(n, np.int32)
        self.time = np.zeros(n, np.float64)
        self.toffset = np.zeros(n, dt_)
        self.flow = flow
        self._flow_table = None if flow is None else np.ascontiguousarray(flow.table, dtype=dt_)
        self.gcf = np.zeros((n, dof), dt_)
        self.rpm = np.zeros((n, self.nthr), dt_)
        self.nfev = np.zeros(n, np.int64)
        # smallest distance between each env's trajectory and a discontinuity of the reference's RHS during the LAST step
        # (mvrl_oracle.c "Distance-to-discontinuity bookkeeping"): [sign of a zero-dt PID increment, thruster dead-band,
        # integrator wind-up, yaw-error branch, |cos(theta)|]
        self.margins = np.full((n, 5), np.inf)

    def reset(self, init, toffset=None):
        """init [n, init_dim]: wp0, wp1, target angles (see include/mvrl.h).  Mirrors 6DoF.py:485-529."""
        init = np.asarray(init, dtype=np.float64).reshape(self.n, -1)
        np_ = self.npos
        self.path[:] = init[:, :2 * np_]
        self.sp[:, :np_] = init[:, :np_]
        self.sp[:, np_:] = init[:, 2 * np_:]
        self.y[:] = 0
        self.eold[:] = 0
        self.eint[:] = 0
        self.told[:] = 0
        self.has_old[:] = 0
        self.istep[:] = 0
        self.time[:] = 0
        if toffset is not None:
            self.toffset[:] = toffset
        return self.observe()

    def observe(self):
        return np.stack([self.o.obs_rov(self.dof, self.y[i], self.path[i], self.sp[i]) for i in range(self.n)])

    noise, noise_seed = 0.0, 0   # > 0: perturbation-ensemble member (mvrl_oracle.c orc_set_noise); tests only

    def step(self, actions):
        o = self.o
        a = o.arr(actions, (self.n, self.dof))
        obs = np.zeros((self.n, self.nobs), o.dtype)
        rew = np.zeros(self.n, o.dtype)
        done = np.zeros(self.n, np.uint8)
        f = o._f("orc_rov_step")
        f.restype = C.c_int
        f.argtypes = ([C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_double] + [C.c_int] * 5 + [C.c_void_p] * 10
                      + [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p]
                      + [C.c_void_p] * 7)
        fl = self.flow
        setn = o._f("orc_set_noise")
        setn.restype, setn.argtypes = None, [C.c_double, C.c_uint64]
        setn(float(self.noise), int(self.noise_seed))
        st = f(self.dof, C.addressof(o.rov6), C.addressof(o.rov3), self.n, float(self.dt),
               0 if self.integrator == "rk4" else 1, int(self.n_sub), int(self.control_mode), int(self.fixed_sp),
               int(self.max_steps),
               a.ctypes.data, self.y.ctypes.data, self.sp.ctypes.data, self.path.ctypes.data, self.eold.ctypes.data,
               self.eint.ctypes.data, self.told.ctypes.data, self.has_old.ctypes.data, self.istep.ctypes.data,
               self.time.ctypes.data,
               None if fl is None else self._flow_table.ctypes.data,
               0 if fl is None else fl.table.shape[0], 0 if fl is None else fl.table.shape[1],
               0 if fl is None else fl.table.shape[2], 0.0 if fl is None else fl.dt, 0.0 if fl is None else fl.dx,
               0.0 if fl is None else fl.dy, self.toffset.ctypes.data,
               obs.ctypes.data, rew.ctypes.data, done.ctypes.data, self.gcf.ctypes.data, self.rpm.ctypes.data,
               self.nfev.ctypes.data, self.margins.ctypes.data)
        setn(0.0, 0)
        if st != 0:
            raise RuntimeError("oracle RK45: step size too small")
        return obs, rew, done


class OracleAuvEnv:
    """N independent AuvEnv instances stepped by the oracle (tag/verySimpleAuv.py:216-410)."""

    def __init__(self, n, precision="f64", dt=0.02, max_steps=250, flow=None, auv=None):
        self.o = Oracle(precision, auv=auv)
        self.n, self.dt, self.max_steps = n, dt, max_steps
        d = self.o.dtype
        self.pose = np.zeros((n, 6), d)
        self.tgt = np.zeros((n, 3), d)          # positionTarget (2), headingTarget
        self.iwp = np.zeros(n, np.int32)        # AuvEnvCyl way-point index
        self.err_o = np.zeros((n, 3), d)
        self.mult = np.ones((n, 11), d)
        self.toffset = np.zeros(n, d)
        self.hist = np.zeros((n, 30), d)
        self.istep = np.zeros(n, np.int32)
        self.flow = flow
        self._flow_table = None if flow is None else np.ascontiguousarray(flow.table, dtype=d)
        self.aux = np.zeros((n, 11), d)

    def reset(self, init):
        """init [n,16] = x y heading headingTarget|iWp tOffset mult(11)   (verySimpleAuv.py:216-262, _cyl.py:113-163)."""
        init = np.asarray(init, dtype=np.float64).reshape(self.n, 16)
        self.pose[:] = 0
        self.pose[:, :3] = init[:, :3]
        nwp = int(self.o.auv.n_waypoints)
        if nwp > 0:
            self.iwp[:] = init[:, 3].astype(np.int32)
            wps = np.array(self.o.auv.waypoints)[:3 * nwp].reshape(nwp, 3)
            self.tgt[:] = wps[self.iwp]
        else:
            self.tgt[:] = 0
            self.tgt[:, 2] = init[:, 3]
        self.toffset[:] = init[:, 4]
        self.mult[:] = init[:, 5:16]
        self.hist[:] = 0
        self.istep[:] = 0
        obs = np.zeros((self.n, 11), self.o.dtype)
        f = self.o._f("orc_auv_obs")
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        for i in range(self.n):
            f(C.addressof(self.o.auv), self.pose[i].ctypes.data, self.tgt[i].ctypes.data, self.err_o[i].ctypes.data, 0,
              obs[i].ctypes.data)
        return obs

    def step(self, actions):
        o = self.o
        a = o.arr(actions, (self.n, 3))
        obs = np.zeros((self.n, 11), o.dtype)
        rew = np.zeros(self.n, o.dtype)
        done = np.zeros(self.n, np.uint8)
        f = o._f("orc_auv_step")
        f.argtypes = ([C.c_void_p, C.c_int64, C.c_double, C.c_int] + [C.c_void_p] * 9
                      + [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double] + [C.c_void_p] * 4)
        fl = self.flow
        f(C.addressof(o.auv), self.n, float(self.dt), int(self.max_steps), a.ctypes.data, self.pose.ctypes.data,
          self.tgt.ctypes.data, self.iwp.ctypes.data, self.err_o.ctypes.data, self.mult.ctypes.data,
          self.toffset.ctypes.data,
          self.hist.ctypes.data, self.istep.ctypes.data,
          None if fl is None else self._flow_table.ctypes.data,
          0 if fl is None else fl.table.shape[0], 0 if fl is None else fl.table.shape[1],
          0 if fl is None else fl.table.shape[2], 0.0 if fl is None else fl.dt, 0.0 if fl is None else fl.dx,
          0.0 if fl is None else fl.dy,
          obs.ctypes.data, rew.ctypes.data, done.ctypes.data, self.aux.ctypes.data)
        return obs, rew, done
