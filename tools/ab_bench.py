#!/usr/bin/env python3
"""A/B timing on ONE box in ONE call: every arm is a (label, library, bench.py arguments) triple; the arms are run
round-robin for several rounds so that box-to-box and clock-drift differences cancel.  Prints one line per run and a
summary (median of the rounds per arm).

    python tools/ab_bench.py --rounds 3 --arm base:: --arm full:fulltrig: --arm c1::"--chains 1"
        arm = label:variant:extra-args   (variant '' = the default libmvrl.so, else variants_build/libmvrl_<variant>.so)
"""
import argparse
import json
import os
import shlex
import statistics
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ap = argparse.ArgumentParser()
ap.add_argument("--arm", action="append", required=True)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--common", default="--workload c4 --no-cpu-baseline --steps 2000 --warmup 100 --repeats 3")
args = ap.parse_args()

arms = []
for a in args.arm:
    label, variant, extra = a.split(":", 2)
    lib = os.path.join(REPO, "variants_build", f"libmvrl_{variant}.so") if variant else None
    arms.append((label, lib, shlex.split(extra)))
res = {a[0]: [] for a in arms}
for rnd in range(args.rounds):
    for label, lib, extra in arms:
        env = dict(os.environ)
        if lib:
            env["MVRL_LIB"] = lib
        while extra and "=" in extra[0] and not extra[0].startswith("-"):     # leading KEY=value words of an arm: environment
            k_, v_ = extra[0].split("=", 1)
            env[k_] = v_
            extra = extra[1:]
        r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + shlex.split(args.common) + extra, env=env,
                           capture_output=True, text=True)
        try:
            j = json.loads(r.stdout.strip().splitlines()[-1])
            us = j["ms_per_step"] * 1e3
            res[label].append(us)
            print(f"round {rnd} {label:12s} {us:8.2f} us/step  value {j['value']:.3e}  kernel_us/step {j['roofline']['kernel_us_per_step']:.2f} "
                  f"frac {j['roofline']['frac']:.3f} reps {[round(x * 1e3, 1) for x in j['timing']['ms_per_step_repeats']]}", flush=True)
        except Exception:  # noqa: BLE001
            print(label, "FAILED", r.stdout[-300:], r.stderr[-600:], flush=True)
print("---- summary (median over rounds, us per step)")
for label, v in res.items():
    if v:
        print(f"{label:12s} median {statistics.median(v):8.2f}  min {min(v):8.2f}  max {max(v):8.2f}  n={len(v)}")
