#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/r3_s10_all.log 2>&1; echo "all gpu tests rc=$?"
tail -6 gpurun_out/r3_s10_all.log
timeout -k 10 600 python tools/ab_bench.py --rounds 2 --arm bounded:: --arm prev:prevz: --arm bounded1::"--chains 1 --launch single" --arm prev1:prevz:"--chains 1 --launch single" --arm c3at1M::"--workload c3 --envs-per-gpu 1048576" > gpurun_out/r3_s10_ab.log 2>&1; echo "ab rc=$?"
tail -7 gpurun_out/r3_s10_ab.log
bash tools/valu_count.sh c4 default 2>&1 | tee gpurun_out/r3_s10_valu.log
