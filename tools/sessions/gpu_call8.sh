#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
ROOT=${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fuzz or random_batch or turbulence" > $OUT/r2_t8.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 $OUT/r2_t8.log | cut -c1-300
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 600 python tools/ab_bench.py --rounds 3 --arm lane:: --arm wave:trigwave: > $OUT/r2_ab8.log 2>&1
rc=$?; echo "ab rc=$rc"; tail -4 $OUT/r2_ab8.log; if [ $rc -ge 124 ]; then exit $rc; fi
bash tools/profile_round.sh c4 r02_c4 > $OUT/r2_prof_c4.log 2>&1; rc=$?; echo "profile c4 rc=$rc"; tail -12 $OUT/r2_prof_c4.log | cut -c1-200; if [ $rc -ge 124 ]; then exit $rc; fi
bash tools/profile_round.sh auv r02_auv > $OUT/r2_prof_auv.log 2>&1; rc=$?; echo "profile auv rc=$rc"; tail -3 $OUT/r2_prof_auv.log | cut -c1-200
