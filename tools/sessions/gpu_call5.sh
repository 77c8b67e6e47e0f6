#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 600 python tests/audit/margin_probe.py > $OUT/r2_margins2.log 2>&1; rc=$?; echo "margins rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
N=16384 STEPS=60 timeout -k 10 900 python tests/audit/margin_probe.py > $OUT/r2_margins3.log 2>&1; rc=$?; echo "margins3 rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/r2_bench_driver2.log 2>&1; echo "bench rc=$?"; tail -c 900 $OUT/r2_bench_driver2.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/r2_bench_default2.log 2>&1; echo "bench rc=$?"; tail -c 600 $OUT/r2_bench_default2.log
