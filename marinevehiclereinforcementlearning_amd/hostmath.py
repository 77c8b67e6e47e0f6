"""Tiny host-side helpers of the public API (pure functions of their arguments)."""
import numpy as np


def angle_error(psi_d, psi):
    """resources.angleError (resources.py:75-95) == tag/resources.headingError (tag/resources.py:26-46):
    signed difference in [-pi, pi)."""
    a = (psi_d - psi) % (2. * np.pi)
    b = (psi - psi_d) % (2. * np.pi)
    return a if a < b else -b


headingError = angle_error
