#!/bin/bash
# second session of round 5: fp64 flavours (literal constants = pairs of s_mov_b32 per use vs constants through scalar loads)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 900 python tools/ab_bench.py --rounds 2 --common "--workload c4 --no-cpu-baseline --steps 1000 --warmup 100 --repeats 3 --precision f64" \
  --arm "baked::" --arm "ctrl::--flavour ctrl" --arm "sym::--flavour sym" --arm "fin:f64fin:" --arm "finsym:f64fin:--flavour sym" > gpurun_out/r5b_ab2.log 2>&1
echo "ab rc=$?"
tail -8 gpurun_out/r5b_ab2.log
