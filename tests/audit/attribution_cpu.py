#!/usr/bin/env python3
"""Which fp32 rounding loses the envs over a whole episode - CPU only (VERDICT r4 "next 1a": attribute first).

The fp64 oracle steps a population of the C4 kind (6-DoF + turbulence, random paths / target attitudes / time offsets, uniform random
actions, 250 steps = one episode, 6DoF.py:569-571) once exactly and once per SWITCH with exactly one class of fp32 rounding
committed (mvrl_oracle.c `orc_set_emulate`, plus per-step rounding of word classes of the stored state done here on the arrays).
Printed per switch: share of envs beyond 1e-5 of the exact trajectory at steps 25 / 100 / 250 and the median error at step 250.
Nothing here touches the GPU or the product library: it answers what ANY implementation with that rounding could reach at best.

    python tests/audit/attribution_cpu.py [c4|c3] [n_envs] [steps]
"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from marinevehiclereinforcementlearning_amd import params as P  # noqa: E402   (constants only)
from marinevehiclereinforcementlearning_amd.synthetic import BASE_DT, BASE_DX, synthetic_ltm, synthetic_spod  # noqa: E402
from oracle import flow_ref, oracle as orc  # noqa: E402

CHECK = (25, 100, 250)


def flow_table():
    modes, coeffs = synthetic_spod(8, 2000)
    base = flow_ref.reconstruct(modes, coeffs, synthetic_ltm())
    fd, dx, dy, dt = flow_ref.scale(base, BASE_DX, BASE_DX, BASE_DT, 11., 1., 2.)
    uv = np.ascontiguousarray(fd[..., :2]).astype(np.float32).astype(np.float64)     # the product's table is fp32
    return orc.FlowTable(uv, dt, dx, dy)


def circ_err(a, b, ang):
    d = np.abs(a - b)
    d[:, ang] = np.minimum(d[:, ang], np.abs(d[:, ang] - 2 * np.pi))
    return (d / np.maximum(1.0, np.abs(b))).max(axis=1)


# name -> (emulate mask, word classes of the stored state rounded to fp32 once per env step)
SWITCHES = [
    ("all state words fp32 once per step (plain fp32 storage)", 0, ("pose", "ang", "vel", "pid", "sp")),
    ("... with the three angle words exact (what binary angles buy at best)", 0, ("pose", "vel", "pid", "sp")),
    ("only the 3 position words", 0, ("pose",)),
    ("only the 6 velocity words", 0, ("vel",)),
    ("only the controller memory (eOld, eInt)", 0, ("pid",)),
    ("velocities + controller memory", 0, ("vel", "pid")),
    ("only the turbulence SAMPLE TIME formed in fp32 (round-4 kernels)", 2, ()),
    ("only the sampled current rounded to fp32", 4, ()),
    ("only every RHS OUTPUT rounded to fp32 (state exact: the least an fp32 right-hand side commits)", 1, ()),
    ("forceModel + M^-1 + J in fp32 ARITHMETIC (fp32 build of the oracle), state / controller / allocation exact", 16, ()),
    ("allocateThrust in fp32 arithmetic, everything else exact", 32, ()),
    ("allocation + forceModel + M^-1 + J in fp32 arithmetic, state and controller exact", 16 | 32, ()),
    ("RHS outputs fp32 + state fp32 after every sub-step (an fp32 integrator at its best)", 1 | 8, ()),
    ("RHS outputs + sample time + current + plain fp32 storage", 1 | 2 | 4, ("pose", "ang", "vel", "pid", "sp")),
]


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "c4"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 250
    dof = 6
    ft = flow_table() if which == "c4" else None
    rng = np.random.default_rng(12345)
    path = (rng.random((n, 6)) - 0.5) * 10.0                    # 3DoF.py:423-424 extended to three coordinates (random_init6)
    ang0 = rng.random((n, 3)) * 2 * np.pi
    init = np.concatenate([path, ang0], axis=1)
    toff = (rng.random(n) * (2000 // 4) * ft.dt).astype(np.float32) if ft is not None else None
    kw = dict(n_substeps=4, control_mode=P.CTRL_FAITHFUL, max_steps=10 ** 9, flow=ft)
    setemu = orc.lib().orc_set_emulate_f64
    setemu.restype, setemu.argtypes = None, [orc.C.c_int]
    ref = orc.OracleRovEnv(dof, n, "f64", **kw)
    ref.reset(init, toffset=toff)
    envs = []
    for _ in SWITCHES:
        e = orc.OracleRovEnv(dof, n, "f64", **kw)
        e.reset(init, toffset=toff)
        envs.append(e)
    bad = [np.zeros(n, bool) for _ in SWITCHES]
    rows = {c: [] for c in CHECK}
    med = [0.0] * len(SWITCHES)
    ang = [3, 4, 5]
    arng = np.random.default_rng(2024)
    print(f"# fp32-rounding attribution on the CPU: fp64 oracle, {which} population, {n} envs x {steps} steps, n_sub 4, FAITHFUL; "
          f"share of envs beyond 1e-5 of the exact fp64 trajectory", flush=True)
    t0 = time.time()
    f32 = lambda a: a.astype(np.float32).astype(np.float64)  # noqa: E731
    for s in range(steps):
        a = arng.uniform(-1, 1, (n, dof)).astype(np.float32).astype(np.float64)
        setemu(0)
        ref.step(a)
        for k, ((name, mask, classes), e) in enumerate(zip(SWITCHES, envs)):
            setemu(mask)
            e.step(a)
            setemu(0)
            if "pose" in classes:
                e.y[:, 0:3] = f32(e.y[:, 0:3])
            if "ang" in classes:
                e.y[:, 3:6] = f32(e.y[:, 3:6])
            if "vel" in classes:
                e.y[:, 6:12] = f32(e.y[:, 6:12])
            if "pid" in classes:
                e.eold[:] = f32(e.eold)
                e.eint[:] = f32(e.eint)
            if "sp" in classes:
                e.sp[:] = f32(e.sp)
            err = circ_err(e.y, ref.y, ang)
            bad[k] |= err > 1e-5
            med[k] = float(np.median(err))
        if (s + 1) in CHECK:
            for k in range(len(SWITCHES)):
                rows[s + 1].append(100.0 * bad[k].mean())
            print(f"#   step {s + 1}: " + " ".join(f"{v:.2f}" for v in rows[s + 1]) + f"   [{time.time() - t0:.0f} s]", flush=True)
    print("switch | % of envs beyond 1e-5 at step " + " / ".join(str(c) for c in CHECK if c <= steps) + " | median err at the last step")
    for k, (name, mask, classes) in enumerate(SWITCHES):
        print(f"{name:112s} | " + " / ".join(f"{rows[c][k]:6.2f}" for c in CHECK if c <= steps) + f" | {med[k]:.2e}")


if __name__ == "__main__":
    main()
