#!/bin/bash
# Round-3 evidence session (GPU box): power / clock under the final kernels (bounded vs extrapolating composition), large-sample
# error audits of the other parametrisations, and a 100 000-step soak of C4 with a finiteness check.
cd $GRAFT_REPO_ROOT
{
  echo "# rocm-smi sclk / package power while bench.py runs 12 000-step regions (tools/clock_watch.sh), round-3 final kernels"
  echo "## c4, two chains, bounded composition (every lane finite)"; bash tools/clock_watch.sh "" 12000
  echo "## c4, two chains, MVRL_FLOW_EXTRAPOLATE=1 (rounds 1-2: lanes turn non-finite in the second half of every episode)"
  MVRL_FLOW_EXTRAPOLATE=1 bash tools/clock_watch.sh "" 12000
  echo "## c4, one launch per step, bounded"; bash tools/clock_watch.sh "" 12000 --chains 1 --launch single
  echo "## c3 at 1 048 576 envs (no turbulence)"; bash tools/clock_watch.sh "" 12000 --workload c3 --envs-per-gpu 1048576
} > gpurun_out/r3_power_clock.txt 2>&1
cat gpurun_out/r3_power_clock.txt | grep -v "^$" | tail -34
export MVRL_CPU_THREADS=16 OMP_NUM_THREADS=16
{
  echo "# tests/audit/err_quantiles.py <n> <steps> <n_sub> <mode> <dof>, round-3 final kernels"
  timeout -k 10 400 python tests/audit/err_quantiles.py 1048576 25 4 0 3 2>&1 | grep -v amdgpu.ids
  timeout -k 10 400 python tests/audit/err_quantiles.py 262144 40 4 1 6 2>&1 | grep -v amdgpu.ids
  timeout -k 10 400 python tests/audit/err_quantiles.py 262144 25 8 0 6 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r3_error_audit_other.txt 2>&1
grep -E "lib=|beyond 1e-5|control" gpurun_out/r3_error_audit_other.txt
timeout -k 10 300 python - > gpurun_out/r3_soak.txt 2>&1 <<'PY'
import time, torch, sys
sys.path.insert(0, ".")
from marinevehiclereinforcementlearning_amd.vec_env import MarineVecEnv
from marinevehiclereinforcementlearning_amd.flow import ReconstructedFlow
n = 1048576
flow = ReconstructedFlow.synthetic(n_modes=8, n_time=2000, device=0); flow.scale(11., 1., 2., translate=(-1.65, -1.1))
env = MarineVecEnv("rov6", n, seed=12345, flow=flow, infos="lean")
ring = torch.empty((8, n, 6), device="cuda")
for r in range(8):
    env.handle.fill_uniform_dev(ring[r].data_ptr(), n * 6, 12345, r, -1.0, 1.0, torch.cuda.current_stream().cuda_stream)
env.reset_tensors()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 100000
for k in range(K):
    env.step_tensors(ring[k & 7])
    if k % 20000 == 19999:
        st = torch.from_numpy(env.get_state()[:36])
        print(k + 1, "steps: all planes finite", bool(torch.isfinite(st).all()), "max |uvw| %.2f" % float(st[6:9].abs().max()), flush=True)
torch.cuda.synchronize()
print("soak: %d steps x %d envs = %.2e env-steps (400 episodes per env), one launch per step" % (K, n, K * n))
PY
cat gpurun_out/r3_soak.txt | grep -v amdgpu
