"""numpy restatement of ReconstructedFlow.__init__ / scale (tag/flowGenerator.py:14-95).

TEST INFRASTRUCTURE ONLY (see oracle/mvrl_oracle.c header).  Pinned by tests/golden/g12_flow_interp.npz.
"""
import numpy as np


def reconstruct(modes, coeffs, ltm):
    """baseFlowData[t] = real(modes[Ny,Nx,3,K] @ coeffs[K,t]) + ltm   (flowGenerator.py:19-23)."""
    base = np.real(np.einsum("jick,kt->tjic", modes, coeffs)) + ltm[None]
    return base


def scale(base, base_dx, base_dy, base_dt, size_scale, velocity_scale, turb_scale):
    """flowGenerator.py:76-95 -> (flowData[nT,Ny,Nx,3], dx, dy, dt).  `translate` only moves `coords`, which
    the interpolation never reads (flowGenerator.py:118-120), so it is not a parameter here."""
    fd = base.copy()
    fd[..., 0] *= velocity_scale
    fd[..., 1] *= velocity_scale
    fd[..., 0] = (fd[..., 0] - velocity_scale) * turb_scale + velocity_scale
    fd[..., 1] = (fd[..., 1] - 0.) * turb_scale
    fd[..., 2] /= max(1e-6, (velocity_scale * turb_scale) ** 2.)
    dt = base_dt * size_scale / max(1e-6, velocity_scale)
    return fd, base_dx * size_scale, base_dy * size_scale, dt


def grid_spacing(coords):
    """flowGenerator.py:35-42 incl. the uniformity check."""
    dx = coords[0, 1:, 0] - coords[0, :-1, 0]
    dy = coords[1:, 0, 1] - coords[:-1, 0, 1]
    if not np.all(np.abs(dx - dx[0]) < 1e-6):
        raise ValueError("Non-uniform input grid spacing in the x-direction")
    if not np.all(np.abs(dy - dy[0]) < 1e-6):
        raise ValueError("Non-uniform input grid spacing in the y-direction")
    return dx[0], dy[0]
