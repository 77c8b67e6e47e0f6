"""Closed-form synthetic SPOD basis for the turbulence table (numpy only, no RNG).

The reference's `tag/turbulenceData/coeffs.npy` and `modes_r.npy` are not shipped
(.MISSING_LARGE_BLOBS:67-68), so every consumer of a flow table here - golden generator,
parity tests, benchmark - builds `modes_r` / `coeffs` of the right rank from this formula
(`ReconstructedFlow.__init__`, tag/flowGenerator.py:16-23, reads `coeffs[K, nT]`,
`modes[Ny, Nx, 3, K]` and `ltm[Ny, Nx, 3]`).  A formula instead of a seeded RNG keeps the
table bit-reproducible across numpy versions and needs no storage.
"""
import numpy as np

# Geometry of the shipped grid (tag/turbulenceData/turbulence_coords.npy: 41 x 61 nodes, spacing 0.005).
BASE_NY, BASE_NX, BASE_DX, BASE_DT = 41, 61, 0.005, 0.002


def synthetic_coords(ny=BASE_NY, nx=BASE_NX, dx=BASE_DX):
    """Uniform (y, x)-ordered node coordinates, same layout as turbulence_coords.npy: [Ny, Nx, 2] = (x, y)."""
    x = np.arange(nx, dtype=np.float64) * dx
    y = np.arange(ny, dtype=np.float64) * dx
    c = np.zeros((ny, nx, 2))
    c[:, :, 0] = x[None, :]
    c[:, :, 1] = y[:, None]
    return c


def synthetic_ltm(ny=BASE_NY, nx=BASE_NX):
    """Long-time mean u/Uinf, v/Uinf, Cp close to (1, 0, 0), smooth; stand-in for ltm.npy where the
    shipped one (41 x 61) does not fit the requested grid."""
    j = np.arange(ny, dtype=np.float64)[:, None] / max(1, ny - 1)
    i = np.arange(nx, dtype=np.float64)[None, :] / max(1, nx - 1)
    ltm = np.zeros((ny, nx, 3))
    ltm[:, :, 0] = 1.0 + 0.02 * np.sin(2.1 * i + 0.3) * np.cos(1.7 * j)
    ltm[:, :, 1] = 0.008 * np.cos(2.9 * i) * np.sin(3.1 * j + 0.2)
    ltm[:, :, 2] = -0.01 * np.sin(1.3 * i + 2.2 * j)
    return ltm


def synthetic_spod(n_modes, n_time, ny=BASE_NY, nx=BASE_NX, amplitude=0.02):
    """Return (modes[Ny, Nx, 3, K] complex128, coeffs[K, nT] complex128)."""
    j = np.arange(ny, dtype=np.float64)[:, None, None, None] / max(1, ny - 1)
    i = np.arange(nx, dtype=np.float64)[None, :, None, None] / max(1, nx - 1)
    c = np.arange(3, dtype=np.float64)[None, None, :, None]
    k = np.arange(1, n_modes + 1, dtype=np.float64)[None, None, None, :]
    re = np.sin(np.pi * k * i * 1.5 + 0.9 * c) * np.cos(np.pi * k * j + 0.2 * k)
    im = np.cos(np.pi * k * i * 1.1 - 0.4 * c) * np.sin(np.pi * (k + 1.0) * j * 0.7 + 0.1)
    amp = amplitude * np.array([1.0, 0.8, 0.5])[None, None, :, None]
    modes = (re + 1j * im) * amp
    kk = np.arange(1, n_modes + 1, dtype=np.float64)[:, None]
    t = np.arange(n_time, dtype=np.float64)[None, :]
    coeffs = np.exp(1j * (2.0 * np.pi * t * kk / 97.0 + 0.7 * kk)) / kk
    return modes, coeffs
