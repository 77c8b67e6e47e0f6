#!/usr/bin/env python3
"""Achievable HBM bandwidth of this box in bursts and sustained (device-to-device copy and read-only reduction), to put the
HBM-bound AuvEnv kernel's GB/s in context."""
import time
import torch
n = 1 << 28                      # 1 GiB of fp32
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)
def run(fn, reps, bytes_per_rep):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return bytes_per_rep * reps / dt / 1e12
for name, fn, bpr in (("copy (read + write)", lambda: b.copy_(a), 2 * 4 * n), ("sum (read only)", lambda: a.sum(), 4 * n),
                      ("fill (write only)", lambda: b.fill_(1.0), 4 * n)):
    fn(); fn()
    burst = run(fn, 20, bpr)
    sustained = run(fn, 3000, bpr)
    print(f"{name:22s}: burst (20 reps) {burst:.2f} TB/s, sustained (3000 reps) {sustained:.2f} TB/s")
