#!/usr/bin/env python3
"""Where do the few fp64 envs that leave the reference trajectory come from? (round 5: 25 of 1 048 576 C4 envs beyond 1e-5 after a
250-step episode in precision = f64.)  A rare-branch bug - the full-sincos fall-back of the stage rotation, the wind-up reset, the yaw
wrap - would look exactly like chaos in the totals, so this looks at each departing env: the step at which it FIRST differs from the
fp64 oracle by more than 1e-10 (the two are independent fp64 implementations: a different operation order, 1e-16 per operation), and what
the oracle recorded in that step and the one before - the distance of its trajectory to each discontinuity of the reference's right-hand
side (OracleRovEnv.margins) and the largest angle increment of an RK stage (the stage rotation's fast path ends at 0.25 rad).

    python tests/audit/f64_outliers.py [c4|c3] [n] [steps]         (GPU box)
"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from marinevehiclereinforcementlearning_amd import _lib, params as P  # noqa: E402
from marinevehiclereinforcementlearning_amd.flow import ReconstructedFlow  # noqa: E402
from oracle import oracle as orc  # noqa: E402   (a measuring tool, like the tests: not a product path)
from tests.parity_util import NAMES  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "c4"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 250
    dof, use_flow = 6, which == "c4"
    h = _lib.Handle(P.make_config("rov6", n, auto_reset=False, max_steps=10 ** 9, use_flow=use_flow, seed=12345, precision="f64"))
    ft = None
    if use_flow:
        flow = ReconstructedFlow.synthetic(n_modes=8, n_time=2000)
        flow.scale(11., 1., 2., translate=(-1.65, -1.1))
        uv = flow.table_uv()
        h.set_flow(uv.astype(np.float64), flow.dt, flow.dx, flow.dy)
        ft = orc.FlowTable(uv.astype(np.float64), flow.dt, flow.dx, flow.dy)
    h.reset()
    st = h.get_state()
    init = np.concatenate([st[30:36].T, st[24:30].T[:, 3:]], axis=1)
    ref = orc.OracleRovEnv(dof, n, "f64", max_steps=10 ** 9, flow=ft)
    ref.reset(init, toffset=st[-2].copy())
    rng = np.random.default_rng(2024)
    first = np.full(n, -1)                       # step of the first departure beyond 1e-10
    m_at = np.full((n, 5), np.inf)               # smallest distance to each discontinuity in that step or the one before
    inc_at = np.zeros(n)                         # largest |angle rate| * h seen in that step
    err_at = np.zeros(n)
    worst = np.zeros(n)
    prev_m = np.full((n, 5), np.inf)
    hsub = 0.2 / 4
    t0 = time.time()
    for s in range(steps):
        a = rng.uniform(-1, 1, (n, dof)).astype(np.float32).astype(np.float64)
        y_before = ref.y.copy()
        ref.step(a)
        h.step(a, copy=False)
        y = h.get_state()[:12].T
        d = np.abs(y - ref.y)
        d[:, 3:6] = np.minimum(d[:, 3:6], np.abs(d[:, 3:6] - 2 * np.pi))
        e = (d / np.maximum(1.0, np.abs(ref.y))).max(axis=1)
        worst = np.maximum(worst, e)
        new = (first < 0) & (e > 1e-10)
        if new.any():
            first[new] = s + 1
            m_at[new] = np.minimum(ref.margins[new], prev_m[new])
            # Euler-angle rates ~ body rates (up to 1 / cos(theta)): the stage increments are at most h * |rate| of the faster end of the step
            rate = np.maximum(np.abs(y_before[new, 9:12]).max(axis=1), np.abs(ref.y[new, 9:12]).max(axis=1))
            inc_at[new] = hsub * rate / np.maximum(0.05, np.abs(np.cos(ref.y[new, 4])))
            err_at[new] = e[new]
        prev_m = ref.margins.copy()
        if (s + 1) % 50 == 0:
            print(f"# step {s + 1}: {int((first >= 0).sum())} envs have departed by > 1e-10, {int((worst > 1e-5).sum())} are beyond 1e-5, worst {worst.max():.1e}   [{time.time() - t0:.0f} s]", flush=True)
    dep = np.nonzero(first >= 0)[0]
    print(f"# {which}, precision f64, {n} envs x {steps} steps: {len(dep)} envs departed from the fp64 oracle by > 1e-10 at some step; {int((worst > 1e-5).sum())} ended beyond 1e-5")
    if len(dep) == 0:
        return
    bounds64 = np.array([1e-9, 1e-9, 1e-9, 1e-9, 1e-3])     # "within fp64 reach" of a discontinuity: distances are relative quantities of O(1)
    near = (m_at[dep] < bounds64).any(axis=1)
    kind = np.argmin(m_at[dep] / bounds64, axis=1)
    big_inc = inc_at[dep] > 0.25
    print(f"# of the departed: {int(near.sum())} were within 1e-9 of a discontinuity of the reference's RHS in that step or the one before "
          f"(" + ", ".join(f"{nm} {int(((kind == k) & near).sum())}" for k, nm in enumerate(NAMES)) + f"); {int((~near).sum())} were not")
    print(f"# stage rotation: {int(big_inc.sum())} of the departed had a stage angle increment beyond 0.25 rad (the fall-back path) in their first step; "
          f"all envs / steps with such an increment would depart if that path were wrong")
    order = dep[np.argsort(-worst[dep])][:30]
    print("# the 30 worst: env | first departure step | error there | final worst | smallest distances (pid sign, dead-band, wind-up, yaw branch, cos theta) | est. max stage increment [rad]")
    for i in order:
        print(f"{i:8d} | {first[i]:4d} | {err_at[i]:.1e} | {worst[i]:.1e} | " + " ".join(f"{v:.1e}" for v in m_at[i]) + f" | {inc_at[i]:.3f}")
    h.close()


if __name__ == "__main__":
    main()
