#!/usr/bin/env python3
"""Phase timing of the 6-DoF step kernel from in-kernel s_memtime stamps (build variant `stamp`, tools/variants.py):
    python tools/variants.py build stamp && MVRL_LIB=variants_build/libmvrl_stamp.so python tools/stamp_probe.py
Per wave: t0 start, t1 state/actions/flow arrived, t2 RK4 loop done, t3 stores issued, t4 stores acknowledged."""
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from marinevehiclereinforcementlearning_amd import _lib  # noqa: E402
from marinevehiclereinforcementlearning_amd.flow import ReconstructedFlow  # noqa: E402
from marinevehiclereinforcementlearning_amd.vec_env import MarineVecEnv  # noqa: E402

n = int(os.environ.get("N", 1048576))
flow = ReconstructedFlow.synthetic(n_modes=8, n_time=2000, device=0)
flow.scale(11., 1., 2., translate=(-1.65, -1.1))
env = MarineVecEnv("rov6", n, seed=1, flow=flow, infos="lean")
env.reset_tensors()
act = torch.rand((4, n, 6), device="cuda") * 2 - 1
for k in range(int(os.environ.get('WARM', 3000))):
    env.step_tensors(act[k % 4])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
env.step_tensors(act[0])
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
lib = _lib.load()
W = 32768
buf = np.zeros(5 * W, np.uint64)
lib.mvrl_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
rc = lib.mvrl_debug_stamps(buf.ctypes.data, buf.size)
assert rc == 0, rc
nw = min(W, (n + 63) // 64)
t = buf.reshape(5, W)[:, :nw].astype(np.int64)
if hasattr(lib, "mvrl_debug_stamps_rt"):
    rt = np.zeros(5 * W, np.uint64)
    lib.mvrl_debug_stamps_rt.argtypes = [C.c_void_p, C.c_size_t]
    assert lib.mvrl_debug_stamps_rt(rt.ctypes.data, rt.size) == 0
    rt = rt.reshape(5, W)[:, :nw].astype(np.int64)
    clk = (t[2] - t[1]) / np.maximum(1, rt[2] - rt[1]) * 100.0   # MHz: shader cycles per 10 ns tick of s_memrealtime
    print(f"in-kernel shader clock over the RK4 loop (s_memtime / s_memrealtime): median {np.median(clk):.0f} MHz  p10 {np.percentile(clk, 10):.0f}  p90 {np.percentile(clk, 90):.0f}")
    life = (rt[4] - rt[0]) * 10e-3
    print(f"wave life in real time: median {np.median(life):.2f} us, loop {np.median((rt[2] - rt[1]) * 10e-3):.2f} us, load {np.median((rt[1] - rt[0]) * 10e-3):.2f} us")
for k in range(5):
    print("slot", k, "zeros", int((t[k] == 0).sum()), "min", int(t[k].min()), "max", int(t[k].max()))
# s_memtime counters of different XCDs are not aligned: cluster the waves by counter domain (gaps >> kernel length)
order = np.argsort(t[0])
gaps = np.nonzero(np.diff(t[0][order]) > 10_000_000)[0]
groups = np.split(order, gaps + 1)
print(f"launch {ms * 1e3:.1f} us by events; {len(groups)} counter domains (XCDs) with {[len(g) for g in groups]} waves")
spans = []
for g in groups:
    t[:, g] -= t[0, g].min()
    spans.append(t[4, g].max())
span = float(np.median(spans))
tick_us = span / (ms * 1e3)
print(f"kernel span per domain {min(spans)}..{max(spans)} ticks -> ~{tick_us:.0f} ticks/us if the span is the launch")
def stat(name, d):
    print(f"  {name:28s} mean {d.mean():9.0f} ticks  p10 {np.percentile(d, 10):9.0f}  p50 {np.percentile(d, 50):9.0f}  p90 {np.percentile(d, 90):9.0f}  max {d.max():9.0f}   (mean {d.mean() / tick_us:6.2f} us)")
stat("load phase   t1-t0", t[1] - t[0])
stat("RK4 loop     t2-t1", t[2] - t[1])
stat("epilogue     t3-t2", t[3] - t[2])
stat("store drain  t4-t3", t[4] - t[3])
stat("wave life    t4-t0", t[4] - t[0])
edges = np.linspace(0, span, 21)
mid = 0.5 * (edges[:-1] + edges[1:])
print("  waves alive per 5% slice :", [int(np.sum((t[0] <= m) & (t[4] > m))) for m in mid])
print("  waves in RK4 loop        :", [int(np.sum((t[1] <= m) & (t[2] > m))) for m in mid])
print("  waves in load phase      :", [int(np.sum((t[0] <= m) & (t[1] > m))) for m in mid])
first_round = t[0] < 0.02 * span
print(f"  first-round waves ({int(first_round.sum())}): load phase mean {np.mean((t[1] - t[0])[first_round]):.0f} ticks, loop {np.mean((t[2] - t[1])[first_round]):.0f}")
late = t[0] > 0.5 * span
print(f"  waves started in the 2nd half ({int(late.sum())}): load phase mean {np.mean((t[1] - t[0])[late]):.0f} ticks, loop {np.mean((t[2] - t[1])[late]):.0f}")
print(f"  last start at {t[0].max() / tick_us:.1f} us, 50% of waves ended by {np.percentile(t[4], 50) / tick_us:.1f} us, 99% by {np.percentile(t[4], 99) / tick_us:.1f} us, all by {t[4].max() / tick_us:.1f} us")
