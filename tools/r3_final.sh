#!/bin/bash
# Round-3 closing session (GPU box): the full GPU test-suite, smoke(), the profile session, the bench table and the driver's command
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/r3_final_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 gpurun_out/r3_final_tests.log
timeout -k 10 600 python __graft_entry__.py smoke > gpurun_out/r3_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r3_smoke.log
bash tools/r3_profiles.sh
cd $GRAFT_REPO_ROOT
bash tools/r3_table.sh
