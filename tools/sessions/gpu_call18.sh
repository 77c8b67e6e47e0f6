#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
for lib in "" variants_build/libmvrl_prev.so; do
for a in "65536 25 4 0" "65536 25 8 0" "65536 25 4 1"; do
MVRL_LIB=${lib:+$PWD/$lib} timeout -k 10 300 python tests/audit/err_quantiles.py $a >> $OUT/r2_errq18.log 2>&1; rc=$?
if [ $rc -ge 124 ]; then exit $rc; fi
done; done
cat $OUT/r2_errq18.log
