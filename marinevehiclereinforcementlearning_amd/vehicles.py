"""Vehicle-level mirrors of the reference's model classes for callers that drive `derivs(t, state)` themselves (e.g. with
their own `solve_ivp`, as dynamicsModel_BlueROV2_Heavy_6DoF.py:686-745 and example_trialTrajectories.py:66-146 do):

    BlueROV2Heavy6DoF_PID_controller(setPoint)   6DoF.py:27-73   (memory: eOld / eInt / tOld, reset())
    BlueROV2Heavy6DoF(controller)                6DoF.py:75-442  (derivs(t, state); generalisedControlForces, rpm)
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

    def _memory(self):
        return self.controller


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
