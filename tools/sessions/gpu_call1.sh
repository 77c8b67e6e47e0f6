#!/bin/bash
# round-2 GPU call 1: parity suite on the new build, then A/B of chains x stage-trig, then the driver's own bench arguments
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/r2_t1.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 $OUT/r2_t1.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 600 python tools/ab_bench.py --rounds 2 --arm trig_c2:: --arm trig_c1::"--chains 1" --arm full_c2:fulltrig: --arm full_c1:fulltrig:"--chains 1" --arm trig_c2_nostag::"--no-stagger" --arm trig_c3::"--chains 3" > $OUT/r2_ab1.log 2>&1
rc=$?; echo "ab rc=$rc"; tail -8 $OUT/r2_ab1.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/r2_bench_driver.log 2>&1
echo "bench rc=$?"; tail -c 1500 $OUT/r2_bench_driver.log
