#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py tests/test_gpu_chains.py tests/test_gpu_f64.py tests/test_gpu_fullsize.py -q -m gpu -x -s > gpurun_out/r3_s14_tests.log 2>&1; echo "tests rc=$?"
tail -2 gpurun_out/r3_s14_tests.log; grep -E "g09_rk4_6dof_faithful_nsub4_x64|dof=6 mode=0 n_sub=4" gpurun_out/r3_s14_tests.log | cut -c1-200
bash tools/valu_count.sh c4 default nos3 2>&1 | tee gpurun_out/r3_s14_valu.log
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python tools/ab_bench.py --rounds 3 --arm s3:: --arm nos3:nos3: > gpurun_out/r3_s14_ab.log 2>&1; echo "ab rc=$?"
tail -3 gpurun_out/r3_s14_ab.log
export MVRL_CPU_THREADS=16 OMP_NUM_THREADS=16
timeout -k 10 400 python tests/audit/err_quantiles.py 1048576 25 4 0 6 2>&1 | grep -v amdgpu.ids | head -4
