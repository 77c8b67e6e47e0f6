#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_bench_contract.py tests/test_gpu_chains.py -m gpu -q > $OUT/r2_t13.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -6 $OUT/r2_t13.log | cut -c1-300
if [ $rc -ge 124 ]; then exit $rc; fi
bash tools/profile_round.sh c4 r02_c4 > $OUT/r2_prof_c4.log 2>&1; rc=$?; echo "profile c4 rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
bash tools/profile_round.sh auv r02_auv > $OUT/r2_prof_auv.log 2>&1; rc=$?; echo "profile auv rc=$rc"
