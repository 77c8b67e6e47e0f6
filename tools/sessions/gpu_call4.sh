#!/bin/bash
# round-2 GPU call 4: discontinuity margins of the outlier lanes (calibration of tests/parity_util.F32_BOUNDS), a focused
# re-run of the tests that failed in call 3, and the power / clock picture of the 3- and 4-wave builds
OUT=gpurun_out
mkdir -p $OUT
ROOT=${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 600 python tests/audit/margin_probe.py > $OUT/r2_margins.log 2>&1; rc=$?; echo "margins rc=$rc"; tail -3 $OUT/r2_margins.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 600 python -m pytest tests/test_gpu_edge.py tests/test_gpu_chains.py "tests/test_gpu_api.py::test_rollout_equals_k_steps" -m gpu -q > $OUT/r2_t4.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 $OUT/r2_t4.log
if [ $rc -ge 124 ]; then exit $rc; fi
( bash tools/clock_watch.sh "" 12000; bash tools/clock_watch.sh $ROOT/variants_build/libmvrl_nopark.so 12000; bash tools/clock_watch.sh "" 12000 --chains 1; bash tools/clock_watch.sh "" 12000 --control-mode zoh; bash tools/clock_watch.sh "" 12000 --workload auv ) > $OUT/r2_power.log 2>&1
echo "power rc=$?"; cat $OUT/r2_power.log
