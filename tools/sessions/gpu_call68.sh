#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 800 python tests/audit/err_quantiles.py 1048576 25 4 0 > $OUT/r2_errq68.log 2>&1; rc=$?
grep -v "amdgpu.ids\|^   sensitive" $OUT/r2_errq68.log | cut -c1-300
exit $rc
