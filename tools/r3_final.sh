#!/bin/bash
# smoke + profile session + bench table + driver-args line with the final kernels
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python __graft_entry__.py smoke > gpurun_out/r3_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r3_smoke.log
bash tools/r3_profiles.sh
cd $GRAFT_REPO_ROOT
bash tools/r3_table.sh
