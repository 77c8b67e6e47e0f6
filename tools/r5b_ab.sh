#!/bin/bash
# second session of round 5: the 3-DoF kernel after exp2 replaced the library expf (default library) against the library without it
# (variant incsel = this session's flags only) - C2 timing, and the 3-DoF parity tests on the new library
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/ -x -q -m gpu -k "rov3 or 3dof or dof3 or c2 or configs0 or three" > gpurun_out/r5b_rov3_tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5b_rov3_tests.log
timeout -k 10 600 python tools/ab_bench.py --rounds 3 --common "--no-cpu-baseline --warmup 100 --repeats 3" \
  --arm "c2::--workload c2 --steps 4000" --arm "c2old:incsel:--workload c2 --steps 4000" \
  --arm "c2graph::--workload c2 --steps 4000 --warmup 96 --graph" --arm "c2roll::--workload c2 --steps 4000 --warmup 96 --rollout" \
  > gpurun_out/r5b_ab4.log 2>&1
echo "ab rc=$?"
tail -6 gpurun_out/r5b_ab4.log
