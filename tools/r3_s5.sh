#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py tests/test_gpu_edge.py tests/test_gpu_chains.py tests/test_gpu_f64.py -x -q -m gpu -s > gpurun_out/r3_s5_parity.log 2>&1; echo "parity rc=$?"
grep -E "g09_|passed|failed|Error|resolver" gpurun_out/r3_s5_parity.log | tail -24
bash tools/valu_count.sh c4 default prevz 2>&1 | tee gpurun_out/r3_s5_valu.log
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python tools/ab_bench.py --rounds 2 --arm new:: --arm prev:prevz: --arm new1::"--chains 1 --launch single" --arm prev1:prevz:"--chains 1 --launch single" > gpurun_out/r3_s5_ab.log 2>&1; echo "ab rc=$?"
tail -6 gpurun_out/r3_s5_ab.log
