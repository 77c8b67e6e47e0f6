#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/r3_s11_all.log 2>&1; echo "all gpu tests rc=$?"
tail -6 gpurun_out/r3_s11_all.log
grep -n "520 steps\|hiprtc build\|jit generic\|jit sym" gpurun_out/r3_s11_all.log | head
