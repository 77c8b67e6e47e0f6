"""Vehicle-level mirrors of the reference's model classes for callers that drive `derivs(t, state)` themselves (e.g. with
their own `solve_ivp`, as dynamicsModel_BlueROV2_Heavy_6DoF.py:686-745 and example_trialTrajectories.py:66-146 do):

    BlueROV2Heavy6DoF_PID_controller(setPoint)   6DoF.py:27-73   (memory: eOld / eInt / tOld, reset())
    BlueROV2Heavy6DoF(controller)                6DoF.py:75-442  (derivs(t, state); generalisedControlForces, rpm;
                                                 updateMovingCoordSystem / globalToVehicle / vehicleToGlobal,
                                                 allocateThrust, forceModel -> (M, RHS): the public methods
                                                 example_trialTrajectories.py:100-134 calls)
    BlueROV2Heavy3DoF(setPoint)                  3DoF.py:25-296  (controller inlined in the vehicle, as in the reference)

`derivs` is evaluated by the HIP kernels through `mvrl_derivs` (one lane per call here; `Handle.derivs` takes batches).
The controller memory lives in these Python objects exactly like the reference's attributes, so a caller may inspect or
reset it between calls.  precision="f64" reproduces the reference's numbers to 1e-9 (tests/test_gpu_units.py)."""
import numpy as np

from . import _lib, params as P


class BlueROV2Heavy6DoF_PID_controller(object):
    def __init__(self, setPoint):
        self.setPoint = np.asarray(setPoint, dtype=np.float64)
        self.reset()

    def reset(self):
        self.eOld = None
        self.eInt = np.zeros(6)
        self.tOld = 0.


class _Vehicle(object):
    _dof = 6

    def _open(self, precision, **overrides):
        key = "rov6" if self._dof == 6 else "rov3"
        kw = {}
        if overrides:
            kw[key] = (P.rov6_params if self._dof == 6 else P.rov3_params)(**overrides)
        self._h = _lib.Handle(P.make_config(key, 1, use_flow=False, precision=precision, **kw))
        self.generalisedControlForces = np.zeros(self._dof)
        self.rpm = np.zeros(8 if self._dof == 6 else 4)

    def _memory(self):
        raise NotImplementedError

    def derivs(self, t, state):
        mem = self._memory()
        first = mem.eOld is None
        r = self._h.derivs(float(t), np.asarray(state, np.float64)[None], np.asarray(mem.setPoint, np.float64)[None],
                           eold=np.zeros((1, self._dof)) if first else np.asarray(mem.eOld, np.float64)[None],
                           eint=np.asarray(mem.eInt, np.float64)[None], told=float(mem.tOld), has_old=not first)
        mem.eOld = r["eold"][0].astype(np.float64)
        mem.eInt = r["eint"][0].astype(np.float64)
        mem.tOld = float(r["told"][0])
        self.generalisedControlForces = r["gcf"][0].astype(np.float64)
        self.rpm = r["rpm"][0].astype(np.float64)
        return r["dy"][0].astype(np.float64)

    def close(self):
        self._h.close()


class BlueROV2Heavy6DoF(_Vehicle):
    _dof = 6

    def __init__(self, controller, precision="f64", **overrides):
        self.controller = controller
        self._open(precision, **overrides)
        self._angles = np.zeros(3)
        self.rotation_angles = np.zeros(3)
        self.iHat, self.jHat, self.kHat = np.eye(3)
        self.controlVector = np.zeros(8)

    def _memory(self):
        return self.controller

    # ---- the vehicle's other public methods, evaluated on the GPU through mvrl_vehicle_ops -----------------------------------
    def updateMovingCoordSystem(self, rotation_angles):
        """6DoF.py:238-242: sets iHat, jHat, kHat for the XYZ-intrinsic attitude."""
        self.rotation_angles = rotation_angles            # 6DoF.py:240: the stored orientation is a public attribute
        self._angles = np.asarray(rotation_angles, np.float64)
        ax = self._h.vehicle_ops(self._angles[None], want=("axes",))["axes"][0].astype(np.float64)
        self.iHat, self.jHat, self.kHat = ax[0], ax[1], ax[2]

    def globalToVehicle(self, vecGlobal):
        """6DoF.py:244-248"""
        return np.array([np.dot(vecGlobal, self.iHat), np.dot(vecGlobal, self.jHat), np.dot(vecGlobal, self.kHat)])

    def vehicleToGlobal(self, vecVehicle):
        """6DoF.py:250-251"""
        return vecVehicle[0] * self.iHat + vecVehicle[1] * self.jHat + vecVehicle[2] * self.kHat

    def allocateThrust(self):
        """6DoF.py:220-231: rpm demands for self.generalisedControlForces at the attitude of the last updateMovingCoordSystem."""
        r = self._h.vehicle_ops(self._angles[None], gcf=np.asarray(self.generalisedControlForces, np.float64)[None], want=("rpm",))
        self.controlVector = r["rpm"][0].astype(np.float64)
        return self.controlVector

    def forceModel(self, pos, angles, vel, rpms, retComp=False):
        """6DoF.py:253-404: (M, RHS) for the given attitude, body velocities and thruster rpm; with retComp=True the 6 x 5 breakdown
        [-Crb.vel, -Ca.vel, -D.vel, G, H] (:401-402) instead."""
        controlVector = rpms
        if retComp:
            return self._h.force_components(np.asarray(angles, np.float64)[None], np.asarray(vel, np.float64)[None],
                                            np.asarray(rpms, np.float64)[None])[0].astype(np.float64)
        r = self._h.vehicle_ops(np.asarray(angles, np.float64)[None], rpm=np.asarray(controlVector, np.float64)[None],
                                vel=np.asarray(vel, np.float64)[None], want=("rhs",))
        M = np.array(self._h.cfg.rov6.mass, dtype=np.float64).reshape(6, 6)
        return M, r["rhs"][0].astype(np.float64)


class BlueROV2Heavy3DoF(_Vehicle):
    _dof = 3

    def __init__(self, setPoint, precision="f64", **overrides):
        self.setPoint = np.asarray(setPoint, dtype=np.float64)
        self.eOld = None
        self.eInt = np.zeros(3)
        self.tOld = 0.
        self._open(precision, **overrides)

    def _memory(self):
        return self
