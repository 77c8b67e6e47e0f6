#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py tests/test_gpu_f64.py -x -q -m gpu -s -k "reference_rk4 or derivs or golden or facade or components" > gpurun_out/r3_s4_parity.log 2>&1; echo "parity rc=$?"
grep -E "g09_|derivs[36] fp32|passed|failed|Error" gpurun_out/r3_s4_parity.log | tail -30
timeout -k 10 600 python tools/ab_bench.py --rounds 2 --common "--workload c2 --no-cpu-baseline --steps 4000 --warmup 200 --repeats 3" --arm c2:: --arm c2blk32:blk32: > gpurun_out/r3_s4_ab.log 2>&1; echo "ab rc=$?"
tail -4 gpurun_out/r3_s4_ab.log
