"""Host-side mirror of the reference's turbulence field class, backed by the HIP operators of libmvrl.so.

`ReconstructedFlow` keeps the interface of tag_00_Dec2023_simpleControlTurbulence/flowGenerator.py:13-159
(constructor from a data directory, `scale`, `interp`, `interpField`, attributes `dx dy dt time coords flowData`):
  * __init__ + scale  -> mvrl_flow_reconstruct : real(modes @ coeffs) + ltm fused with scale()'s affine map
  * interp            -> mvrl_flow_interp      : the trilinear (t, y, x) lookup, for one point or arrays of points
The environments consume the (u, v) part of the table through `table_uv()`.
"""
import os

import numpy as np

from . import _lib
from .synthetic import BASE_DT, synthetic_coords, synthetic_ltm, synthetic_spod


class ReconstructedFlow(object):
    def __init__(self, dataDir=None, *, modes=None, coeffs=None, lt_mean=None, coords=None, time_step=None, device=0):
        """dataDir: directory with coeffs.npy, modes_r.npy, ltm.npy, params_coeffs.yaml, turbulence_coords.npy
        (flowGenerator.py:16-29); or pass the arrays directly."""
        self.device = device
        if dataDir is not None:
            import yaml
            self.coeffs = np.load(os.path.join(dataDir, "coeffs.npy"), allow_pickle=False)
            self.modes = np.load(os.path.join(dataDir, "modes_r.npy"), allow_pickle=False)
            self.lt_mean = np.load(os.path.join(dataDir, "ltm.npy"), allow_pickle=False)
            with open(os.path.join(dataDir, "params_coeffs.yaml"), "r") as infile:
                params = yaml.safe_load(infile)
            self.baseDt = float(params["time_step"])
            self.baseCoords = np.load(os.path.join(dataDir, "turbulence_coords.npy"), allow_pickle=False)
        else:
            if modes is None or coeffs is None or lt_mean is None or coords is None:
                raise ValueError("give dataDir or modes/coeffs/lt_mean/coords")
            self.coeffs, self.modes, self.lt_mean = np.asarray(coeffs), np.asarray(modes), np.asarray(lt_mean)
            self.baseCoords = np.asarray(coords, dtype=np.float64)
            self.baseDt = float(BASE_DT if time_step is None else time_step)
        if self.modes.ndim != 4 or self.modes.shape[2] != 3 or self.coeffs.shape[0] != self.modes.shape[3]:
            raise ValueError("modes must be [Ny, Nx, 3, K] and coeffs [K, nT]")
        self.nT = int(self.coeffs.shape[1])
        self.baseTime = np.array([i * self.baseDt for i in range(self.nT)])
        # flowGenerator.py:35-42 - uniform spacing check
        self.baseDx = self.baseCoords[0, 1:, 0] - self.baseCoords[0, :-1, 0]
        self.baseDy = self.baseCoords[1:, 0, 1] - self.baseCoords[:-1, 0, 1]
        if not np.all(np.abs(self.baseDx - self.baseDx[0]) < 1e-6):
            raise ValueError("Non-uniform input grid spacing in the x-direction")
        if not np.all(np.abs(self.baseDy - self.baseDy[0]) < 1e-6):
            raise ValueError("Non-uniform input grid spacing in the y-direction")
        self.baseDx = float(self.baseDx[0])
        self.baseDy = float(self.baseDy[0])
        self._flowData = None
        self.scale(1., 1., 1.)

    @classmethod
    def synthetic(cls, n_modes=8, n_time=2000, ny=41, nx=61, ltm=None, device=0):
        """Seed-free synthetic SPOD data of the shipped grid size (the reference's coeffs/modes blobs are not
        distributed): the benchmark's and the tests' turbulence table."""
        modes, coeffs = synthetic_spod(n_modes, n_time, ny, nx)
        return cls(modes=modes, coeffs=coeffs, lt_mean=synthetic_ltm(ny, nx) if ltm is None else ltm,
                   coords=synthetic_coords(ny, nx), device=device)

    def scale(self, sizeScale, velocityScale, turbScale, translate=(0, 0)):
        """flowGenerator.py:53-95.  The table itself is (re)built lazily on the GPU with the affine map folded in:
        u' = (u V - V) T + V = (V T) u + V (1 - T) ;  v' = (V T) v ;  Cp' = Cp / max(1e-6, (V T)^2)."""
        self.coords = self.baseCoords.copy() * sizeScale + translate
        self.dx = self.baseDx * sizeScale
        self.dy = self.baseDy * sizeScale
        V, T = float(velocityScale), float(turbScale)
        self._mul = [V * T, V * T, 1.0 / max(1e-6, (V * T) ** 2.)]
        self._add = [V - V * T, 0.0, 0.0]
        self.dt = self.baseDt * sizeScale / max(1e-6, velocityScale)
        self.time = np.array([i * self.dt for i in range(self.nT)])
        self._flowData = None

    @property
    def flowData(self):
        """[nT, Ny, Nx, 3] float32, reconstructed + scaled on the GPU on first use."""
        if self._flowData is None:
            self._flowData = _lib.flow_reconstruct(self.modes, self.coeffs, self.lt_mean, self._mul, self._add,
                                                   device=self.device)
        return self._flowData

    def table_uv(self):
        return np.ascontiguousarray(self.flowData[..., :2])

    def interp(self, time, xy):
        """flowGenerator.py:97-136.  Scalars give one [3] vector; arrays (time[n], xy[n, 2]) give [n, 3]."""
        t = np.atleast_1d(np.asarray(time, dtype=np.float32))
        xy = np.asarray(xy, dtype=np.float32).reshape(-1, 2)
        out = _lib.flow_interp(self.flowData, self.dt, self.dx, self.dy, t, xy[:, 0], xy[:, 1], device=self.device)
        return out[0] if np.ndim(time) == 0 else out

    def interp_bounded(self, time, xy):
        """How the 3-/6-DoF + turbulence composition samples the table (no reference counterpart): `interp`
        inside it; outside it the boundary value held in space and time reflected over the table's duration (triangle wave) -
        `interp` at the clamped / reflected coordinates, exactly what the step kernels' flow_gather does with `bounded` set."""
        t = np.atleast_1d(np.asarray(time, dtype=np.float64))
        xy2 = np.asarray(xy, dtype=np.float64).reshape(-1, 2)
        per = (self.nT - 1) * self.dt
        m = np.mod(t, 2.0 * per)
        tr = per - np.abs(m - per)
        x = np.clip(xy2[:, 0], 0.0, (self.flowData.shape[2] - 1) * self.dx)
        y = np.clip(xy2[:, 1], 0.0, (self.flowData.shape[1] - 1) * self.dy)
        out = self.interp(tr, np.stack([x, y], axis=1))
        return out[0] if np.ndim(time) == 0 else out

    def interpField(self, time):
        """flowGenerator.py:138-159 (time-only interpolation of the whole plane; host arithmetic on the table)."""
        tt = time / self.dt
        kk = min(self.nT - 2, max(0, int(np.floor(tt))))
        w1 = tt - kk
        return self.flowData[kk] * (1. - w1) + self.flowData[kk + 1] * w1
