#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 600 python tools/ab_bench.py --rounds 2 --common "--workload c2 --envs-per-gpu 1048576 --no-cpu-baseline --steps 1000 --warmup 50 --repeats 3" --arm lds4:: --arm nolds:rov3nolds: --arm lds4_zoh::"--control-mode zoh" --arm nolds_zoh:rov3nolds:"--control-mode zoh" > $OUT/r2_ab14.log 2>&1
rc=$?; echo "ab rc=$rc"; tail -6 $OUT/r2_ab14.log
timeout -k 10 300 python tools/ab_bench.py --rounds 2 --common "--workload c2 --no-cpu-baseline --steps 2000 --warmup 50 --repeats 3" --arm lds4:: --arm nolds:rov3nolds: > $OUT/r2_ab14b.log 2>&1
tail -3 $OUT/r2_ab14b.log
