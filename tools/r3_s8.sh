#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3_s8_all.log 2>&1; echo "all gpu tests rc=$?"
tail -5 gpurun_out/r3_s8_all.log
timeout -k 10 600 python tools/ab_bench.py --rounds 2 --arm new:: --arm prev:prevz: --arm new1::"--chains 1 --launch single" --arm prev1:prevz:"--chains 1 --launch single" > gpurun_out/r3_s8_ab.log 2>&1; echo "ab rc=$?"
tail -6 gpurun_out/r3_s8_ab.log
bash tools/valu_count.sh c4 default 2>&1 | tee gpurun_out/r3_s8_valu.log
