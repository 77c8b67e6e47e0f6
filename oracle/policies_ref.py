"""numpy restatement of the reference's two baseline policies (TEST INFRASTRUCTURE ONLY, pinned by
tests/golden/g17_pd_policy.npz and g18_los_policy.npz):
  pd_predict      PDController.predict            tag_00_Dec2023_simpleControlTurbulence/verySimpleAuv.py:22-50
  line_of_sight   lineOfSight / LOSNavigation     dynamicsModel_BlueROV2_Heavy_3DoF.py:517-607
"""
import numpy as np


def pd_predict(obs, old, dt, P, D):
    """One PDController.predict call for a batch: obs [n, >=3], old [n, 3] or None -> (actions [n, 3], new old)."""
    x = np.asarray(obs, dtype=np.float64)[:, :3]
    if old is None:
        old = x
    a = np.clip(x * P + (x - old) / dt * D, -1., 1.)
    return np.clip(a, -1., 1.), x.copy()


def line_of_sight(p0, p1, Rnav):
    p0, p1 = np.asarray(p0, float), np.asarray(p1, float)
    if np.sqrt(np.sum(p1 ** 2.)) < Rnav:
        return p1
    pathVec = p1 - p0
    dSegment = np.sqrt(np.sum(pathVec ** 2.))
    pHat = pathVec / dSegment
    det = p0[0] * p1[1] - p1[0] * p0[1]
    delta = Rnav ** 2. * dSegment ** 2. - det ** 2.
    if delta < 0:
        dAlong = np.dot(-p0, pHat)
        if dAlong > dSegment:
            return p1
        if dAlong < 0:
            return p0
        return p0 + dAlong * pHat
    sy = np.sign(pathVec[1])
    if np.abs(sy) < 1e-12:
        sy = 1.
    den = max(1e-6, dSegment) ** 2.
    pp0 = np.array([(det * pathVec[1] + sy * pathVec[0] * np.sqrt(delta)) / den,
                    (-det * pathVec[0] + np.abs(pathVec[1]) * np.sqrt(delta)) / den])
    pp1 = np.array([(det * pathVec[1] - sy * pathVec[0] * np.sqrt(delta)) / den,
                    (-det * pathVec[0] - np.abs(pathVec[1]) * np.sqrt(delta)) / den])
    s0 = np.dot(pHat, pp0 - p0) / max(1e-6, dSegment)
    s1 = np.dot(pHat, pp1 - p0) / max(1e-6, dSegment)
    if (s0 >= 0.) and (s0 <= 1.) and (s0 > s1):
        return pp0
    if (s1 >= 0.) and (s1 <= 1.):
        return pp1
    if np.linalg.norm(p1) < np.linalg.norm(p0):
        return p1
    return p0


def los_predict(obs, Rnav=0.5):
    obs = np.asarray(obs, dtype=np.float64)
    out = np.zeros((len(obs), 3))
    for i, o in enumerate(obs):
        out[i, :2] = line_of_sight(o[:2], o[2:4], Rnav)
        out[i, 2] = o[4]
    return out
