#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py tests/test_gpu_chains.py tests/test_gpu_f64.py tests/test_gpu_fullsize.py -q -m gpu -x > gpurun_out/r3_s12_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r3_s12_tests.log
bash tools/valu_count.sh c4 default novote 2>&1 | tee gpurun_out/r3_s12_valu.log
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python tools/ab_bench.py --rounds 3 --arm vote:: --arm novote:novote: --arm vote1::"--chains 1 --launch single" --arm novote1:novote:"--chains 1 --launch single" > gpurun_out/r3_s12_ab.log 2>&1; echo "ab rc=$?"
tail -6 gpurun_out/r3_s12_ab.log
