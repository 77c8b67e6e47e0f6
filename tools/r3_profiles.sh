#!/bin/bash
# Round-3 profile session (GPU box): rocprofv3 kernel-trace stats + one --pmc pass per counter group for every BASELINE config
# and the AuvEnv workloads; summaries are made afterwards from gpurun_out/ by tools/summarize_counters.py (see tools/r3_summarize.sh).
cd $GRAFT_REPO_ROOT
for spec in "c4 r03_c4" "c3 r03_c3" "c2 r03_c2" "auv r03_auv"; do
  set -- $spec
  bash tools/profile_round.sh $1 $2 > gpurun_out/prof_$2.log 2>&1; rc=$?; echo "$2 rc=$rc"; cd $GRAFT_REPO_ROOT
  if [ $rc -ge 124 ]; then exit $rc; fi
done
bash tools/profile_round.sh auv r03_auv4m --envs-per-gpu 4194304 > gpurun_out/prof_r03_auv4m.log 2>&1; rc=$?; echo "auv4m rc=$rc"; cd $GRAFT_REPO_ROOT
if [ $rc -ge 124 ]; then exit $rc; fi
MVRL_PROFILE_PMC=0 bash tools/profile_round.sh c4 r03_c4zoh --control-mode zoh > gpurun_out/prof_r03_c4zoh.log 2>&1; echo "c4zoh rc=$?"; cd $GRAFT_REPO_ROOT
MVRL_PROFILE_PMC=0 bash tools/profile_round.sh c4 r03_c4gen --flavour generic --specialize > gpurun_out/prof_r03_c4gen.log 2>&1; echo "c4gen rc=$?"; cd $GRAFT_REPO_ROOT
du -sh gpurun_out | tail -1
