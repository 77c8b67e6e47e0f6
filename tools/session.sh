#!/bin/bash
# The measurement sessions behind profiles/<round>_* (round = r04, r05 ...).  Parts that need the GPU run on the box through gpurun;
# `summarize` runs in the build container afterwards, on what gpurun merged back into gpurun_out/.
#   tools/session.sh profiles    r04   (GPU)  rocprofv3 kernel-trace stats + one --pmc pass per counter group, every BASELINE config + AuvEnv
#   tools/session.sh summarize   r04   (CPU)  -> profiles/r04_counters.json, profiles/r04_*_kernel_stats.csv
#   tools/session.sh table       r04   (GPU)  -> gpurun_out/r04_bench_table.txt (tools/bench_table.sh) + the driver's command
#   tools/session.sh driver_runs r04   (GPU)  -> gpurun_out/r04_driver_args_runs.txt: eight runs of the driver's command on one box
#   tools/session.sh evidence    r04   (GPU)  -> gpurun_out/r04_power_clock.txt, r04_error_audit_other.txt, r04_soak.txt
PART=$1; R=${2:-r04}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; ROOT=$PWD
case "$PART" in
profiles|profiles_a|profiles_b)
  # one gpurun call is limited to 20 minutes: part a = the four BASELINE-sized workloads, part b = the rest
  if [ "$PART" != "profiles_b" ]; then
  for spec in "c4 ${R}_c4" "c3 ${R}_c3" "c2 ${R}_c2" "auv ${R}_auv"; do
    set -- $spec
    bash tools/profile_round.sh $1 $2 > gpurun_out/prof_$2.log 2>&1; rc=$?; echo "$2 rc=$rc"; cd $ROOT
    if [ $rc -ge 124 ]; then exit $rc; fi
  done
  fi
  if [ "$PART" != "profiles_a" ]; then
  bash tools/profile_round.sh auv ${R}_auv4m --envs-per-gpu 4194304 > gpurun_out/prof_${R}_auv4m.log 2>&1; rc=$?; echo "auv4m rc=$rc"; cd $ROOT
  if [ $rc -ge 124 ]; then exit $rc; fi
  bash tools/profile_round.sh c4 ${R}_c4f64 --precision f64 --steps 300 --warmup 30 > gpurun_out/prof_${R}_c4f64.log 2>&1; rc=$?; echo "c4f64 rc=$rc"; cd $ROOT
  if [ $rc -ge 124 ]; then exit $rc; fi
  MVRL_PROFILE_PMC=0 bash tools/profile_round.sh c4in ${R}_c4in > gpurun_out/prof_${R}_c4in.log 2>&1; echo "c4in rc=$?"; cd $ROOT
  MVRL_PROFILE_PMC=0 bash tools/profile_round.sh c4 ${R}_c4zoh --control-mode zoh > gpurun_out/prof_${R}_c4zoh.log 2>&1; echo "c4zoh rc=$?"; cd $ROOT
  MVRL_PROFILE_PMC=0 bash tools/profile_round.sh c4 ${R}_c4gen --flavour generic --specialize > gpurun_out/prof_${R}_c4gen.log 2>&1; echo "c4gen rc=$?"; cd $ROOT
  fi
  du -sh gpurun_out | tail -1 ;;
summarize)
  python tools/summarize_counters.py ${R}_c4 c4 rov6_step_kernel 1048576 216 365 > /dev/null
  python tools/summarize_counters.py ${R}_c3 c3 rov6_step_kernel 262144 152 297 > /dev/null
  python tools/summarize_counters.py ${R}_c2 c2 rov3_step_kernel 65536 84 165 > /dev/null
  python tools/summarize_counters.py ${R}_auv auv auv_step_kernel 1048576 292 393 > /dev/null
  python tools/summarize_counters.py ${R}_auv4m auv_4194304 auv_step_kernel 4194304 292 393 > /dev/null
  python tools/summarize_counters.py ${R}_c4f64 c4_f64 rov6_step_kernel 1048576 432 730 > /dev/null
  python tools/summarize_counters.py ${R}_c4in c4_in_table rov6_step_kernel 1048576 216 341 > /dev/null
  python tools/summarize_counters.py ${R}_c4zoh c4_zoh rov6_step_kernel 1048576 216 365 > /dev/null
  python tools/summarize_counters.py ${R}_c4gen c4_generic_specialised rov6_step_kernel 1048576 216 365 > /dev/null
  python - $R <<'PY'
import json, sys
j = json.load(open("profiles/%s_counters.json" % sys.argv[1]))
print("hash", j["kernel_source_hash"], "commit", j["commit"])
for k, v in j["workloads"].items():
    print("%-24s chains %s us  single %s us  traffic/env %s B (alg %s)  valu/wave-step %s" % (
        k, v.get("bench_command_kernel_avg_us"), v.get("chains1_kernel_avg_us"), v.get("hbm_bytes_per_env"), v.get("algorithmic_bytes_per_env"),
        (v.get("valu") or {}).get("wave_instr_per_env_step")))
PY
  ;;
table)
  bash tools/bench_table.sh > gpurun_out/${R}_bench_table.txt 2>&1
  cat gpurun_out/${R}_bench_table.txt
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${R}_driver_args.json 2> gpurun_out/${R}_driver_args.err; echo "driver-args rc=$?" ;;
driver_runs)
  for i in 1 2 3 4 5 6 7 8; do
    timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /tmp/da_$i.json 2>/dev/null
    python - $i <<'PY'
import json, sys
j = json.loads(open("/tmp/da_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
r = j["roofline"]; s = r["single_launch"]
reps = j["timing"]["launch_plans"]
print("run %s: value %.3e  chains %.1f us/step (frac %.3f; %d repeats %.1f..%.1f)  single %.1f us (frac %.3f; repeats %.1f..%.1f)" % (
    sys.argv[1], j["value"], j["ms_per_step"] * 1e3, r["frac"], j["timing"]["repeats"], min(reps["chains"]["ms_per_step_repeats"]) * 1e3, max(reps["chains"]["ms_per_step_repeats"]) * 1e3,
    s["kernel_us_per_launch"], s["frac"], min(reps["single"]["ms_per_step_repeats"]) * 1e3, max(reps["single"]["ms_per_step_repeats"]) * 1e3), flush=True)
PY
  done | tee gpurun_out/${R}_driver_args_runs.txt ;;
evidence)
  {
    echo "# rocm-smi sclk / package power while bench.py runs 12 000-step regions (tools/clock_watch.sh), $R kernels"
    echo "## c4, two chains"; bash tools/clock_watch.sh "" 12000
    echo "## c4, one launch per step"; bash tools/clock_watch.sh "" 12000 --chains 1 --launch single
    echo "## c3 at 1 048 576 envs (no turbulence)"; bash tools/clock_watch.sh "" 12000 --workload c3 --envs-per-gpu 1048576
    echo "## c4, precision f64, two chains"; bash tools/clock_watch.sh "" 5000 --precision f64
    echo "## auv at 4 194 304 envs (HBM-bound)"; bash tools/clock_watch.sh "" 4000 --workload auv --envs-per-gpu 4194304
  } > gpurun_out/${R}_power_clock.txt 2>&1
  grep -v "^$" gpurun_out/${R}_power_clock.txt | tail -24
  export MVRL_CPU_THREADS=16 OMP_NUM_THREADS=16
  {
    echo "# tests/audit/err_quantiles.py <n> <steps> <n_sub> <mode> <dof>, $R kernels"
    timeout -k 10 400 python tests/audit/err_quantiles.py 1048576 25 4 0 6 2>&1 | grep -v amdgpu.ids
    timeout -k 10 400 python tests/audit/err_quantiles.py 1048576 25 4 0 3 2>&1 | grep -v amdgpu.ids
    timeout -k 10 400 python tests/audit/err_quantiles.py 262144 40 4 1 6 2>&1 | grep -v amdgpu.ids
    timeout -k 10 400 python tests/audit/err_quantiles.py 262144 25 8 0 6 2>&1 | grep -v amdgpu.ids
  } > gpurun_out/${R}_error_audit_25.txt 2>&1
  grep -E "lib=|beyond 1e-5|control" gpurun_out/${R}_error_audit_25.txt
  timeout -k 10 300 python tools/soak.py 100000 > gpurun_out/${R}_soak.txt 2>&1; tail -2 gpurun_out/${R}_soak.txt ;;
audit_f64_1m)
  # the fp64 mode at BASELINE configs[3]'s full size for a whole episode: 1 048 576 envs x 250 steps against the fp64 oracle (the oracle is
  # the slow side: ~6 min on the box's 16 cores)
  export MVRL_CPU_THREADS=16 OMP_NUM_THREADS=16
  MVRL_AUDIT_PRECISION=f64 timeout -k 10 900 python tests/audit/episode_audit.py c4 1048576 250 > gpurun_out/${R}_audit_c4_f64_1m.txt 2>&1
  tail -8 gpurun_out/${R}_audit_c4_f64_1m.txt ;;
audits)
  # whole-episode parity audits (65 536 envs x 250 steps each against the fp64 oracle; tests/audit/episode_audit.py) -> gpurun_out/<round>_audit_*.txt
  export MVRL_CPU_THREADS=16 OMP_NUM_THREADS=16
  bash tools/gpu_steps.sh \
    "${R}_audit_c4|400|python tests/audit/episode_audit.py c4 65536 250" \
    "${R}_audit_c4_t32|400|if [ -f variants_build/libmvrl_flowt32.so ]; then MVRL_LIB=variants_build/libmvrl_flowt32.so python tests/audit/episode_audit.py c4 65536 250; else echo 'A/B arm skipped: build it first in the container: python tools/variants.py build flowt32'; fi" \
    "${R}_audit_c3|400|python tests/audit/episode_audit.py c3 65536 250" \
    "${R}_audit_c2|400|python tests/audit/episode_audit.py c2 65536 250" \
    "${R}_audit_c4_f64|400|MVRL_AUDIT_PRECISION=f64 python tests/audit/episode_audit.py c4 65536 250" \
    "${R}_audit_c3_f64|400|MVRL_AUDIT_PRECISION=f64 python tests/audit/episode_audit.py c3 65536 250" \
    "${R}_audit_c2_f64|400|MVRL_AUDIT_PRECISION=f64 python tests/audit/episode_audit.py c2 65536 250" ;;
*) echo "usage: tools/session.sh profiles_a|profiles_b|summarize|table|driver_runs|evidence|audits <round>"; exit 2 ;;
esac
