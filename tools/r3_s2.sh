#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py tests/test_gpu_edge.py tests/test_gpu_chains.py -x -q -m gpu > gpurun_out/r3_s2_parity.log 2>&1; echo "parity rc=$?"
tail -5 gpurun_out/r3_s2_parity.log
bash tools/valu_count.sh c4 default noyaw nofb 2>&1 | tee gpurun_out/r3_s2_valu.log
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python tools/ab_bench.py --rounds 2 --arm base:: --arm noyaw:noyaw: --arm base1::"--chains 1 --launch single" --arm noyaw1:noyaw:"--chains 1 --launch single" > gpurun_out/r3_s2_ab.log 2>&1; echo "ab rc=$?"
tail -8 gpurun_out/r3_s2_ab.log
