#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
rm -f $OUT/r2_errq46.log
for a in "1048576 25 4 0 3" "262144 25 8 0 6" "262144 40 4 1 6"; do
timeout -k 10 500 python tools/err_quantiles.py $a >> $OUT/r2_errq46.log 2>&1; rc=$?
if [ $rc -ge 124 ]; then exit $rc; fi
done
grep -v "amdgpu.ids\|^   sensitive" $OUT/r2_errq46.log | cut -c1-260
