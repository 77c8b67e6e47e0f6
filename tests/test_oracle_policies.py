"""Baseline policies: numpy restatement (oracle side) vs goldens generated from the imported reference."""
import numpy as np

from .conftest import golden


def test_pd_policy_restatement():
    from oracle import policies_ref as R
    g = golden("g17_pd_policy.npz")
    for c in range(g["obs"].shape[0]):
        old = None
        for t in range(g["obs"].shape[1]):
            a, old = R.pd_predict(g["obs"][c, t][None], old, float(g["dt"]), g["P"][c], g["D"][c])
            assert np.max(np.abs(a[0] - g["actions"][c, t])) < 1e-13, (c, t)


def test_los_policy_restatement():
    from oracle import policies_ref as R
    g = golden("g18_los_policy.npz")
    a = R.los_predict(g["obs"], float(g["Rnav"]))
    assert np.max(np.abs(a - g["actions"])) < 1e-12
