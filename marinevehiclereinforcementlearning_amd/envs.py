"""Single-environment Gym (0.21 API) facades with the reference's class names, constructor arguments, methods and
the attributes its helper scripts poke - each one a 1-lane instance of the HIP kernels behind libmvrl.so.

    BlueROV2Heavy6DoFEnv   dynamicsModel_BlueROV2_Heavy_6DoF.py:445-594
    BlueROV2Heavy3DoFEnv   dynamicsModel_BlueROV2_Heavy_3DoF.py:375-514
    AuvEnv                 tag_00_Dec2023_simpleControlTurbulence/verySimpleAuv.py:76-416  (+ make_env :419-433)

`reset(...) -> obs`, `step(action) -> (obs, reward, done, {})`.  The integrator of the 3/6-DoF models is this build's
fixed-step RK4 (`n_substeps` sub-steps per env step, default 4) instead of the reference's adaptive
scipy RK45 - see DESIGN.md; everything else (set-point from action, PID inside the RHS, angle wrap, observation,
time limit, the `timeHistory` columns) follows the reference line by line.
"""
import numpy as np

from . import _lib, params as P
from .spaces import unit_box


class _Namespace(object):
    pass


class _RovEnvBase(object):
    _DOF = 6

    def __init__(self, seed=None, dt=0.2, maxSteps=250, n_substeps=4, control_mode="faithful", device=0,
                 vehicle_params=None, integrator="rk4", precision=None):
        """seed, dt, maxSteps: the reference's arguments (6DoF.py:446 / 3DoF.py:376).  integrator="rk45" (implies
        precision="f64") selects the reference's own adaptive solve_ivp(RK45) and reproduces its env.step
        trajectories; the default is this build's fixed-step RK4 in fp32."""
        self._integ = integrator
        self._prec = precision or ("f64" if integrator == "rk45" else "f32")
        self.seed = seed
        self.dt = dt
        self._max_episode_steps = maxSteps
        dof = self._DOF
        self.lenAction = dof
        self.lenObs = 9 if dof == 6 else 5
        self.action_space = unit_box(self.lenAction)
        self.observation_space = unit_box(self.lenObs)
        self._n_sub = n_substeps
        self._cm = {"faithful": P.CTRL_FAITHFUL, "zoh": P.CTRL_ZOH}[control_mode]
        self._device = device
        self._vp = vehicle_params
        self._h = None
        self._fixed = None
        self.steps_beyond_done = 0

    # -- handle management: fixedSp is a launch-time switch of the kernel -----------------------------------
    def _handle(self, fixed):
        if self._h is None or self._fixed != fixed:
            if self._h is not None:
                self._h.close()
            kw = {}
            if self._vp is not None:
                kw["rov6" if self._DOF == 6 else "rov3"] = self._vp
            cfg = P.make_config("rov6" if self._DOF == 6 else "rov3", 1, dt=self.dt, n_substeps=self._n_sub,
                                max_steps=self._max_episode_steps, control_mode=self._cm, fixed_setpoint=fixed,
                                auto_reset=False, use_flow=False, device=self._device, precision=self._prec,
                                integrator=self._integ, **kw)
            self._h = _lib.Handle(cfg)
            self._h.enable_aux(True)
            self._fixed = fixed
        return self._h

    def _pull(self):
        dof = self._DOF
        st = self._h.get_state()[:, 0].astype(np.float64)
        self.systemState = st[: 2 * dof].copy()
        sp = st[4 * dof:5 * dof].copy()
        if dof == 6:
            self.vehicle.controller.setPoint = sp
        else:
            self.vehicle.setPoint = sp
        aux = self._h.get_aux()[0].astype(np.float64)
        self.vehicle.generalisedControlForces = aux[:dof].copy()
        self.vehicle.controlVector = aux[dof:].copy()

    def reset(self, initialSetpoint=None):
        dof = self._DOF
        npos = 3 if dof == 6 else 2
        self.iStep = 0
        self.time = 0.
        self.iWp = 0
        if initialSetpoint is None:
            # 3DoF.py:419-427 (the 6-DoF line :497 is broken in the reference; same recipe with three coordinates).
            # Draws come from the GLOBAL numpy generator, exactly like the reference (the `seed` kwarg is inert there).
            self.path = (np.random.rand(npos * 2).reshape((2, npos)) - 0.5) * 10.
            target = np.random.rand(dof - npos) * 2. * np.pi
            sp = np.append(self.path[0, :], target)
            self.fixedSp = False
        else:
            sp = np.array(initialSetpoint, dtype=np.float64)
            self.path = np.vstack([sp[:npos], sp[:npos]])
            target = sp[npos:]
            self.fixedSp = True
        if dof == 6:
            self.targetOrientation = target
        else:
            self.targetHeading = float(target[0])
        self.vehicle = _Namespace()
        self.vehicle.Length = 0.457
        if dof == 6:
            self.vehicle.controller = _Namespace()
        h = self._handle(self.fixedSp)
        init = np.concatenate([self.path.ravel(), target])[None]
        obs = h.reset(init=init)[0].astype(np.float64)
        self._pull()
        self.vehicle.generalisedControlForces = np.zeros(dof)
        self.vehicle.controlVector = np.zeros(8 if dof == 6 else 4)
        self.timeHistory = [self._history_row()]
        self.state = obs
        self.steps_beyond_done = 0
        return self.state

    def _history_row(self):
        sp = self.vehicle.controller.setPoint if self._DOF == 6 else self.vehicle.setPoint
        return np.concatenate([[self.time], self.systemState, self.vehicle.generalisedControlForces,
                               self.vehicle.controlVector, sp])

    def dataToState(self, systemState):
        """6DoF.py:467-483 / 3DoF.py:397-409 (host arithmetic: a pure function of its argument)."""
        from .hostmath import angle_error
        dof = self._DOF
        npos = 3 if dof == 6 else 2
        L3 = self.vehicle.Length * 3.
        sp = self.vehicle.controller.setPoint if dof == 6 else self.vehicle.setPoint
        vals = [(self.path[self.iWp, k] - systemState[k]) / L3 for k in range(npos)]
        vals += [(self.path[self.iWp + 1, k] - systemState[k]) / L3 for k in range(npos)]
        vals += [angle_error(sp[npos + k], systemState[npos + k]) / (45. / 180. * np.pi) for k in range(dof - npos)]
        return np.clip(vals, -1., 1.)

    def step(self, action):
        self.iStep += 1
        self.time += self.dt
        a = np.asarray(action, dtype=np.float64).reshape(1, self._DOF)
        obs, rew, done = self._h.step(a)
        self._pull()
        self.state = obs[0].astype(np.float64)
        done = bool(done[0])
        reward = 0.
        self.timeHistory.append(self._history_row())
        if done:
            import pandas
            dof = self._DOF
            if dof == 6:
                cols = ["t"] + ["x", "y", "z", "phi", "theta", "psi"] + ["u", "v", "w", "p", "q", "r"] \
                    + [f"F{i:d}" for i in range(6)] + [f"u{i:d}" for i in range(8)] \
                    + ["x_d", "y_d", "z_d", "phi_d", "theta_d", "psi_d"]          # 6DoF.py:581-587
            else:
                cols = ["t"] + [f"x{i:d}" for i in range(6)] + [f"F{i:d}" for i in range(3)] \
                    + [f"u{i:d}" for i in range(4)] + ["x_d", "y_d", "psi_d"]     # 3DoF.py:501-507
            self.timeHistory = pandas.DataFrame(data=np.array(self.timeHistory), columns=cols)
            self.steps_beyond_done += 1
        else:
            self.steps_beyond_done = 0
        return self.state, reward, done, {}

    def render(self, mode="human"):
        pass

    def close(self):
        if self._h is not None:
            self._h.close()
            self._h = None


class BlueROV2Heavy6DoFEnv(_RovEnvBase):
    _DOF = 6


class BlueROV2Heavy3DoFEnv(_RovEnvBase):
    _DOF = 3


class AuvEnv(object):
    """tag/verySimpleAuv.py:76-416.  `flow` may be a ready ReconstructedFlow; by default the constructor loads
    "./turbulenceData" like the reference (verySimpleAuv.py:102-104)."""
    _CYL = False
    _MAX_STEPS = 250

    def __init__(self, seed=None, dt=0.02, noiseMagCoeffs=0.0, noiseMagActuation=0.0, currentVelScale=1.0,
                 currentTurbScale=2.0, stopOnBoundsExceeded=True, flow=None, device=0):
        from .flow import ReconstructedFlow
        self.seed = seed
        self._max_episode_steps = self._MAX_STEPS
        self.stopOnBoundsExceeded = stopOnBoundsExceeded
        self.iStep = 0
        self.dt = dt
        self.state = None
        self.steps_beyond_done = None
        self.flow = flow if flow is not None else ReconstructedFlow("./turbulenceData", device=device)
        self.flow.scale(11., currentVelScale, currentTurbScale, translate=(-1.65, -1.1))
        self.timeHistory = []
        self.xMinMax = [-2, 2] if self._CYL else [-1, 1]
        self.yMinMax = [-2, 2] if self._CYL else [-1, 1]
        if self._CYL:                                   # verySimpleAuv_cyl.py:29-41
            self.Rcyl = 1.33
            self.xCyl = np.array([2.5, 0.])
            self.waypoints, self.wpThreshold = P.cylinder_waypoints(self.Rcyl, self.xCyl)
            self.iWp = 0
        self.m, self.Izz = 11.4, 0.16
        self.Xuu, self.Yvv, self.Nrr = -18.18 * 2.21, -21.66 * 4.87, -1.55
        self.Xu, self.Yv, self.Nr = -4.03 * 2.21, -6.22 * 4.87, -0.07
        self.maxForce, self.maxMoment = 150., 20.
        self.noiseMagCoeffs = noiseMagCoeffs
        self.noiseMagActuation = noiseMagActuation
        self.lenAction = 3
        self.action_space = unit_box(3)
        self.observation_space = unit_box(11)
        cfg = P.make_config("auv", 1, dt=dt, max_steps=self._max_episode_steps, auto_reset=False, use_flow=True,
                            device=device, auv=P.auv_params(noiseMagCoeffs, noiseMagActuation, stopOnBoundsExceeded,
                                                            cyl=self._CYL))
        self._h = _lib.Handle(cfg)
        self._h.set_flow(self.flow.table_uv(), self.flow.dt, self.flow.dx, self.flow.dy)
        self._h.enable_aux(True)

    _MULT = ["mMult", "IMult", "XuuMult", "YvvMult", "NrrMult", "XuMult", "YvMult", "NrMult", "XactMult", "YactMult",
             "NactMult"]

    def _push(self):
        """Write the public attributes a caller may have edited (multipliers, pose, targets) into the lane."""
        st = self._h.get_state()
        st[0:2, 0] = self.position
        st[2, 0] = self.heading
        st[3:6, 0] = self.velocities
        st[6, 0] = self.headingTarget
        st[10:21, 0] = [getattr(self, k) for k in self._MULT]
        st[21, 0] = self.flowDataTimeOffset
        if self._CYL:
            st[53:54, 0].view(np.int32)[0] = int(self.iWp)
        self._h.set_state(st)

    def _pull(self):
        st = self._h.get_state()[:, 0].astype(np.float64)
        self.position, self.heading, self.velocities = st[0:2].copy(), float(st[2]), st[3:6].copy()
        self.herr_o, self.perr_o = float(st[7]), st[8:10].copy()
        if self._CYL:
            raw = self._h.get_state()
            object.__setattr__(self, "iWp", int(raw[53:54, 0].view(np.int32)[0]))
            object.__setattr__(self, "positionTarget", self.waypoints[self.iWp, :2])
            object.__setattr__(self, "headingTarget", float(self.waypoints[self.iWp, 2]))

    def reset(self, keepTimeHistory=False, applyNoise=True, fixedInitialValues=None):
        # draw order and formulas of verySimpleAuv.py:222-245, from the global numpy generator like the reference
        if applyNoise:
            mult = np.concatenate([1. + self.noiseMagCoeffs / 2. - np.random.rand(8) * self.noiseMagCoeffs,
                                   1. + self.noiseMagActuation / 2. - np.random.rand(3) * self.noiseMagActuation])
        else:
            mult = np.ones(11)
        for k, v in zip(self._MULT, mult):
            setattr(self, k, float(v))
        if fixedInitialValues is None:
            self.position = (np.random.rand(2) - 0.5) * 0.5 * np.array([self.xMinMax[1] - self.xMinMax[0],
                                                                        self.yMinMax[1] - self.yMinMax[0]])
            self.heading = np.random.rand() * 2. * np.pi
            if not self._CYL:
                self.headingTarget = np.random.rand() * 2. * np.pi
        else:
            self.position = np.array(fixedInitialValues[0], dtype=np.float64)
            self.heading = float(fixedInitialValues[1])
            if not self._CYL:
                self.headingTarget = float(fixedInitialValues[2])
        self.positionStart = self.position.copy()
        if self._CYL:   # the target follows the way-point list; iWp is NOT reset (verySimpleAuv_cyl.py:41,141-142)
            self.positionTarget = self.waypoints[self.iWp, :2]
            self.headingTarget = float(self.waypoints[self.iWp, 2])
        else:
            self.positionTarget = np.zeros(2)
        self.headingStart = self.heading
        self.flowDataTimeOffset = np.random.rand() * self.flow.time[self.flow.time.shape[0] // 4]
        self.velocities = np.zeros(3)
        self.time = 0
        self.iStep = 0
        self.steps_beyond_done = 0
        self.timeHistory = []
        slot3 = float(self.iWp) if self._CYL else self.headingTarget
        init = np.concatenate([self.position, [self.heading, slot3, self.flowDataTimeOffset], mult])
        self.state = self._h.reset(init=init[None].astype(np.float32))[0].astype(np.float64)
        self._pull()
        self._dirty = False
        return self.state

    def __setattr__(self, k, v):
        object.__setattr__(self, k, v)
        if k in AuvEnv._MULT or k in ("flowDataTimeOffset", "headingTarget", "iWp"):
            object.__setattr__(self, "_dirty", True)

    def dataToState(self, pos, heading, velocities):
        """verySimpleAuv.py:147-214 ("V3"), host arithmetic on its arguments + herr_o/perr_o."""
        from .hostmath import angle_error
        perr = self.positionTarget - np.asarray(pos)
        herr = angle_error(self.headingTarget, heading)
        if self.herr_o is None:
            self.herr_o, self.perr_o = herr, perr
        c = lambda x: min(1., max(-1., x))  # noqa: E731
        return np.concatenate([np.array([c(perr[0]), c(perr[1]), c(herr / (45. / 180. * np.pi)), c(herr - self.herr_o),
                                         c(perr[0] - self.perr_o[0]), c(perr[1] - self.perr_o[1])]),
                               np.clip(velocities, -1., 1.), np.zeros(2)])

    def step(self, action):
        if getattr(self, "_dirty", False):  # scripts set multipliers / offsets after reset (script_4, tests)
            self._push()
            object.__setattr__(self, "_dirty", False)
        self.iStep += 1
        self.time += self.dt
        a = np.asarray(action, dtype=np.float32).reshape(1, 3)
        obs, rew, done = self._h.step(a)
        aux = self._h.get_aux()[0].astype(np.float64)
        self._pull()
        self.state = obs[0].astype(np.float64)
        reward, done = float(rew[0]), bool(done[0])
        Fset = a[0, :2] * self.maxForce * np.array([self.XactMult, self.YactMult])
        Nset = a[0, 2] * self.maxMoment * self.NactMult
        names = ["step", "time", "reward", "x", "y", "psi", "x_d", "y_d", "psi_d", "Fx", "Fy", "N", "Fx_set", "Fy_set",
                 "N_set", "u", "v", "r", "u_current", "v_current", "rmsAc"] + [f"r{i:d}" for i in range(5)] \
            + [f"a{i:d}" for i in range(3)] + [f"s{i:d}" for i in range(11)]            # verySimpleAuv.py:389-397
        vals = np.concatenate([[self.iStep, self.time, reward], self.position, [self.heading], self.positionTarget,
                               [self.headingTarget], aux[0:3], Fset, [Nset], self.velocities, aux[3:5], [aux[5]],
                               aux[6:11], a[0], self.state])
        self.timeHistory.append(dict(zip(names, vals)))
        if done:
            import pandas
            self.timeHistory = pandas.DataFrame(self.timeHistory)
            self.steps_beyond_done += 1
        else:
            self.steps_beyond_done = 0
        return self.state, reward, done, {}

    def render(self, mode="human"):
        pass

    def close(self):
        if self._h is not None:
            self._h.close()
            self._h = None


class AuvEnvCyl(AuvEnv):
    """tag/verySimpleAuv_cyl.py:22-344 - AuvEnv following 21 way-points round a cylinder: 1200-step episodes, +-2 m
    bounds, the scaled "V0" observation, target switching inside step()."""
    _CYL = True
    _MAX_STEPS = 1200


def make_env(rank, seed=0, env_kwargs={}):
    """verySimpleAuv.py:419-433 - kept for callers that still build a list of constructors."""
    def _init():
        return AuvEnv(seed=seed + rank, **env_kwargs)
    return _init
