#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py tests/test_gpu_edge.py tests/test_gpu_chains.py tests/test_gpu_f64.py tests/test_gpu_api.py -x -q -m gpu -s > gpurun_out/r3_s7_parity.log 2>&1; echo "parity rc=$?"
grep -E "passed|failed|Error|resolver|dof=3" gpurun_out/r3_s7_parity.log | cut -c1-260 | tail -14
timeout -k 10 600 python tools/ab_bench.py --rounds 2 --common "--workload c2 --no-cpu-baseline --steps 4000 --warmup 200 --repeats 3" --arm c2new:: --arm c2prev:prevz: > gpurun_out/r3_s7_ab.log 2>&1; echo "ab rc=$?"
tail -4 gpurun_out/r3_s7_ab.log
bash tools/valu_count.sh c2 default prevz 2>&1 | tee gpurun_out/r3_s7_valu.log
