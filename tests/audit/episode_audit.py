#!/usr/bin/env python3
"""Whole-episode parity statement (VERDICT r3 "missing 4"): the fp32 HIP step kernel against the fp64 oracle over a full
250-step episode (6DoF.py:569-571) of the BENCH population - random device resets, uniform random actions - at 65 536 envs:
share of envs beyond 1e-5 after 25 / 100 / 250 steps, how many of them jumped next to a discontinuity of the reference's
right-hand side, how many merely drifted, how many jumped further from a discontinuity than the stated fp32 bounds.  Beside it the
same numbers for the fp32 BUILD OF THE ORACLE (the reference's formulation evaluated in fp32 without any of the kernel's
reformulations): the yardstick for what fp32 itself costs on this closed loop.  And a floor: the fp64 oracle whose state (pose,
velocities, controller memory, set-point) is rounded to fp32 ONCE PER ENV STEP - what a kernel with exact arithmetic but the
ABI's fp32 state planes would do; nothing that stores fp32 state between steps can be closer to the fp64 trajectory than that.

    python tests/audit/episode_audit.py c4|c3|c2|auv [n] [steps]        (GPU box; writes to stdout)
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
from tests.parity_util import NAMES, OutlierAudit, SMOOTH_TOL  # noqa: E402

CHECKPOINTS = (25, 50, 100, 150, 200, 250)


def circ_err(a, b, ang):
    d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
    d[..., ang] = np.minimum(d[..., ang], np.abs(d[..., ang] - 2 * np.pi))
    return d / np.maximum(1.0, np.abs(b))


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "c4"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 250
    n_sub = int(os.environ.get("MVRL_AUDIT_NSUB", "4"))                      # other parametrisations of the same audit
    mode = P.CTRL_ZOH if os.environ.get("MVRL_AUDIT_ZOH") else P.CTRL_FAITHFUL
    kw = dict(n_substeps=n_sub, control_mode=mode)
    prec = os.environ.get("MVRL_AUDIT_PRECISION", "f32")                     # f64: the exactness mode's column of the same audit
    dof = 3 if which == "c2" else 6
    use_flow = which == "c4"
    npos = 3 if dof == 6 else 2
    model = "rov6" if dof == 6 else "rov3"
    h = _lib.Handle(P.make_config(model, n, auto_reset=False, max_steps=10 ** 9, use_flow=use_flow, seed=12345, precision=prec, **kw))
    ft = None
    if use_flow:
        flow = ReconstructedFlow.synthetic(n_modes=8, n_time=2000)
        flow.scale(11., 1., 2., translate=(-1.65, -1.1))
        uv = flow.table_uv()
        h.set_flow(uv, flow.dt, flow.dx, flow.dy)
        ft = orc.FlowTable(uv.astype(np.float64), flow.dt, flow.dx, flow.dy)
    h.reset()                                             # the bench's population: Philox resets on the device
    st = h.get_state()
    sp = st[4 * dof:5 * dof].T                            # planes: y[2 dof] eOld[dof] eInt[dof] setPoint[dof] path[2 npos] ... (include/mvrl.h)
    path = st[5 * dof:5 * dof + 2 * npos].T
    init = np.concatenate([path, sp[:, npos:]], axis=1).astype(np.float64)
    toff = st[-2].copy()
    if prec == "f64":
        return main_f64(which, h, n, steps, dof, init, toff, ft, kw)
    ref = orc.OracleRovEnv(dof, n, "f64", max_steps=10 ** 9, flow=ft, **kw)
    low = orc.OracleRovEnv(dof, n, "f32", max_steps=10 ** 9, flow=ft, **kw)
    flo = orc.OracleRovEnv(dof, n, "f64", max_steps=10 ** 9, flow=ft, **kw)
    fla = orc.OracleRovEnv(dof, n, "f64", max_steps=10 ** 9, flow=ft, **kw)      # ... the same floor with the ANGLES left exact (what binary angles buy)
    ref.reset(init, toffset=toff)
    low.reset(init, toffset=toff)
    flo.reset(init, toffset=toff)
    fla.reset(init, toffset=toff)
    ang = [3, 4, 5] if dof == 6 else [2]
    a_gpu, a_low, a_flo, a_fla = (OutlierAudit(n, 1e-5, dof=dof) for _ in range(4))
    rng = np.random.default_rng(2024)
    print(f"# whole-episode audit {which}: {h.variant}, {n} envs x {steps} steps, dt 0.2, n_sub {n_sub}, {'ZOH' if mode == P.CTRL_ZOH else 'FAITHFUL'}, random resets + uniform actions, vs the fp64 oracle")
    print("# step | HIP fp32 kernel: beyond 1e-5 [%]  drifted  jumped(explained)  jumped(beyond the bounds)  median err  q99 of calm envs "
          "| fp32 build of the oracle: beyond 1e-5 [%]  drifted  jumped  beyond the bounds  median err "
          "| fp64 oracle with fp32 state between steps (floor): beyond 1e-5 [%]  median err "
          "| the same floor with exact angles: beyond 1e-5 [%]  median err")
    t0 = time.time()
    for s in range(steps):
        a = rng.uniform(-1, 1, (n, dof)).astype(np.float32)
        ref.step(a.astype(np.float64))
        low.step(a)
        h.step(a)
        flo.step(a.astype(np.float64))
        for arr in (flo.y, flo.eold, flo.eint, flo.sp):
            arr[:] = arr.astype(np.float32)
        efc = circ_err(flo.y, ref.y, ang)
        efl = efc.max(axis=1)
        a_flo.update(efl, ref.margins)
        fla.step(a.astype(np.float64))
        keep = fla.y[:, ang].copy()
        for arr in (fla.y, fla.eold, fla.eint, fla.sp):
            arr[:] = arr.astype(np.float32)
        fla.y[:, ang] = keep
        efa = circ_err(fla.y, ref.y, ang).max(axis=1)
        a_fla.update(efa, ref.margins)
        ec = circ_err(h.get_state()[:2 * dof].T, ref.y, ang)
        e = ec.max(axis=1)
        e32 = circ_err(low.y, ref.y, ang).max(axis=1)
        a_gpu.update(e, ref.margins)
        a_low.update(e32, ref.margins)
        if (s + 1) in CHECKPOINTS or s + 1 == steps:
            calm = e[~a_gpu.jumped]
            print(f"{s + 1:4d} | {100 * a_gpu.bad.mean():7.3f} {int(a_gpu.smooth().sum()):7d} {int(a_gpu.explained().sum()):7d} {int(a_gpu.unexplained().sum()):6d} "
                  f"{np.median(e):.2e} {np.quantile(calm, 0.99):.2e} | {100 * a_low.bad.mean():7.3f} {int(a_low.smooth().sum()):7d} {int(a_low.jumped.sum()):7d} "
                  f"{int(a_low.unexplained().sum()):6d} {np.median(e32):.2e} | {100 * a_flo.bad.mean():7.3f} {np.median(efl):.2e} | {100 * a_fla.bad.mean():7.3f} {np.median(efa):.2e}   [{time.time() - t0:.0f} s]", flush=True)
    assert np.isfinite(h.get_state()[:2 * dof]).all() and np.isfinite(ref.y).all()
    names = ["x", "y", "z", "phi", "theta", "psi", "u", "v", "w", "p", "q", "r"] if dof == 6 else ["x", "y", "psi", "u", "v", "r"]
    print("# last step, scaled error per state word, median / 90 % quantile over all envs - HIP kernel: " +
          "  ".join(f"{nm} {np.median(ec[:, k]):.1e}/{np.quantile(ec[:, k], 0.9):.1e}" for k, nm in enumerate(names)))
    print("#                                                                          storage floor: " +
          "  ".join(f"{nm} {np.median(efc[:, k]):.1e}/{np.quantile(efc[:, k], 0.9):.1e}" for k, nm in enumerate(names)))
    print("# median |state| at the last step: " + "  ".join(f"{nm} {np.median(np.abs(ref.y[:, k])):.2g}" for k, nm in enumerate(names)))
    # which discontinuity the explained jumps were next to
    ex = a_gpu.explained()
    kinds = np.argmin(a_gpu.margin_at_jump[ex] / a_gpu.bounds, axis=1) if ex.any() else np.zeros(0, int)
    print("# HIP kernel, jumps by nearest discontinuity: " + ", ".join(f"{nm} {int((kinds == k).sum())}" for k, nm in enumerate(NAMES)))
    un = np.nonzero(a_gpu.unexplained())[0]
    if len(un):
        r = a_gpu.margin_at_jump[un] / a_gpu.bounds
        print(f"# HIP kernel, {len(un)} jumps beyond the distance bounds: smallest distance / bound quantiles (50/90/max) "
              f"{np.quantile(r.min(axis=1), 0.5):.1f} {np.quantile(r.min(axis=1), 0.9):.1f} {r.min(axis=1).max():.1f}; "
              f"first-jump step quantiles (10/50/90) {np.quantile(a_gpu.first_jump[un], [0.1, 0.5, 0.9]).astype(int).tolist()}")
    print(f"# per step {100 * a_gpu.near_share_per_step():.3f} % of all envs are within the fp32 bounds of a discontinuity; an env that jumped never comes back "
          f"(the closed loop is chaotic under random actions), so the share beyond 1e-5 can only grow with episode length")
    print(f"# jump threshold {SMOOTH_TOL:g}; bounds {a_gpu.bounds.tolist()}")
    h.close()


def main_f64(which, h, n, steps, dof, init, toff, ft, kw):
    """precision = f64 (the mode that follows the reference for whole episodes): worst / median error and the count beyond 1e-5."""
    ref = orc.OracleRovEnv(dof, n, "f64", max_steps=10 ** 9, flow=ft, **kw)
    ref.reset(init, toffset=toff)
    ang = [3, 4, 5] if dof == 6 else [2]
    rng = np.random.default_rng(2024)
    worst = np.zeros(n)
    print(f"# whole-episode audit {which}, precision f64: {h.variant}, {n} envs x {steps} steps vs the fp64 oracle")
    print("# step | envs beyond 1e-5 | worst env so far | median of the per-env worst | 99.99 % quantile")
    t0 = time.time()
    for s in range(steps):
        a = rng.uniform(-1, 1, (n, dof)).astype(np.float32).astype(np.float64)
        ref.step(a)
        h.step(a, copy=False)
        worst = np.maximum(worst, circ_err(h.get_state()[:2 * dof].T, ref.y, ang).max(axis=1))
        if (s + 1) in CHECKPOINTS or s + 1 == steps:
            print(f"{s + 1:4d} | {int((worst > 1e-5).sum()):6d} | {worst.max():.2e} | {np.median(worst):.2e} | {np.quantile(worst, 0.9999):.2e}   [{time.time() - t0:.0f} s]", flush=True)
    h.close()


def main_auv(n, steps):
    """AuvEnv + turbulence (tag/verySimpleAuv.py): episodes end on the time limit (250 steps of 0.02 s) or when the vehicle leaves the
    +-1 m box; an env counts while the fp64 oracle still runs it.  Reports the share of such envs whose pose left 1e-5, the envs that
    terminated at a different step, and the same for the fp32 build of the oracle."""
    flow = ReconstructedFlow.synthetic(n_modes=8, n_time=2000)
    flow.scale(11., 1., 2., translate=(-1.65, -1.1))
    uv = flow.table_uv()
    auv = P.auv_params(noiseMagCoeffs=0.1, noiseMagActuation=0.1)
    h = _lib.Handle(P.make_config("auv", n, dt=0.02, auto_reset=False, max_steps=steps, use_flow=True, seed=12345, auv=auv))
    h.set_flow(uv, flow.dt, flow.dx, flow.dy)
    h.reset()
    st = h.get_state()
    pl = P.STATE_PLANES[P.MODEL_AUV]
    init = np.zeros((n, 16))
    init[:, :3] = st[pl["pose"]][:3].T
    init[:, 3] = st[pl["heading_target"]]
    init[:, 4] = st[pl["toffset"]]
    init[:, 5:] = st[pl["mult"]].T
    ft = orc.FlowTable(uv.astype(np.float64), flow.dt, flow.dx, flow.dy)
    ref = orc.OracleAuvEnv(n, "f64", dt=0.02, max_steps=steps, flow=ft, auv=auv)
    low = orc.OracleAuvEnv(n, "f32", dt=0.02, max_steps=steps, flow=ft, auv=auv)
    ref.reset(init)
    low.reset(init)
    rng = np.random.default_rng(2024)
    alive = np.ones(n, bool)            # the oracle has not finished the env yet
    bad, bad32 = np.zeros(n, bool), np.zeros(n, bool)
    split, split32 = np.zeros(n, bool), np.zeros(n, bool)
    worst_rew = 0.0
    print(f"# whole-episode audit auv: {h.variant}, {n} envs x up to {steps} steps, dt 0.02, random resets + uniform actions, vs the fp64 oracle")
    print("# step | envs still running | HIP fp32 kernel: % of them with the pose beyond 1e-5 (cumulative), terminated at another step, median pose err "
          "| fp32 build of the oracle: % beyond, terminated at another step, median")
    for s in range(steps):
        a = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
        o_ref, r_ref, d_ref = ref.step(a.astype(np.float64))
        o32, r32, d32 = low.step(a)
        o, r, d = h.step(a)
        pose = h.get_state()[:6].T.astype(np.float64)

        def perr(p_):
            dd = np.abs(p_ - ref.pose)
            dd[:, 2] = np.minimum(dd[:, 2], np.abs(dd[:, 2] - 2 * np.pi))
            return (dd / np.maximum(1.0, np.abs(ref.pose))).max(axis=1)
        e, e32 = perr(pose), perr(low.pose.astype(np.float64))
        bad |= alive & (e > 1e-5)
        bad32 |= alive & (e32 > 1e-5)
        split |= alive & ((d != 0) != (d_ref != 0))
        split32 |= alive & ((d32 != 0) != (d_ref != 0))
        ok = alive & ~bad & ~split
        if ok.any():
            worst_rew = max(worst_rew, float((np.abs(r - r_ref) / np.maximum(1.0, np.abs(r_ref)))[ok].max()))
        if (s + 1) in CHECKPOINTS or s + 1 == steps:
            print(f"{s + 1:4d} | {int(alive.sum()):6d} | {100 * bad.mean():7.3f} {int(split.sum()):5d} {np.median(e[alive]) if alive.any() else 0:.2e} "
                  f"| {100 * bad32.mean():7.3f} {int(split32.sum()):5d} {np.median(e32[alive]) if alive.any() else 0:.2e}", flush=True)
        alive &= (d_ref == 0) & (d == 0)
    print(f"# envs whose pose left 1e-5 while the oracle ran them: {int(bad.sum())} of {n} ({100 * bad.mean():.3f} %); terminated at another step: {int(split.sum())}; "
          f"worst reward difference among the others {worst_rew:.1e}")
    h.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "auv":
        main_auv(int(sys.argv[2]) if len(sys.argv) > 2 else 65536, int(sys.argv[3]) if len(sys.argv) > 3 else 250)
    else:
        main()
