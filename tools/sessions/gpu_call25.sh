#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 1000 bash tools/bench_table.sh > $OUT/r2_table25.log 2>&1; rc=$?; echo "table rc=$rc"; cat $OUT/r2_table25.log | cut -c1-200
if [ $rc -ge 124 ]; then exit $rc; fi
( for a in "" "--chains 1" "--control-mode zoh" "--workload auv"; do timeout -k 10 120 bash tools/clock_watch.sh "" 12000 $a; done ) > $OUT/r2_power25.log 2>&1; echo "power rc=$?"; cat $OUT/r2_power25.log | cut -c1-200
