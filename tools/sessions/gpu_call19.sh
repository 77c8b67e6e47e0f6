#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
for a in "65536 25 4 0" "65536 25 8 0" "65536 40 2 0"; do
timeout -k 10 300 python tests/audit/err_quantiles.py $a >> $OUT/r2_errq19.log 2>&1; rc=$?
if [ $rc -ge 124 ]; then exit $rc; fi
done
grep -v amdgpu.ids $OUT/r2_errq19.log
