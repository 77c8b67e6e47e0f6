#!/bin/bash
# After tools/r3_profiles.sh: condense gpurun_out/prof_r03_* into profiles/r03_counters.json and profiles/r03_*_kernel_stats.csv
cd "$(dirname "$0")/.."
python tools/summarize_counters.py r03_c4 c4 rov6_step_kernel 1048576 216 365 > /dev/null
python tools/summarize_counters.py r03_c3 c3 rov6_step_kernel 262144 152 297 > /dev/null
python tools/summarize_counters.py r03_c2 c2 rov3_step_kernel 65536 84 165 > /dev/null
python tools/summarize_counters.py r03_auv auv auv_step_kernel 1048576 292 393 > /dev/null
python tools/summarize_counters.py r03_auv4m auv_4194304 auv_step_kernel 4194304 292 393 > /dev/null
python tools/summarize_counters.py r03_c4zoh c4_zoh rov6_step_kernel 1048576 216 365 > /dev/null
python tools/summarize_counters.py r03_c4gen c4_generic_specialised rov6_step_kernel 1048576 216 365 > /dev/null
python - <<'PY'
import json
j = json.load(open("profiles/r03_counters.json"))
print("hash", j["kernel_source_hash"], "commit", j["commit"])
for k, v in j["workloads"].items():
    print("%-24s chains %s us  single %s us  traffic/env %s B (alg %s)  valu/wave-step %s" % (
        k, v.get("bench_command_kernel_avg_us"), v.get("chains1_kernel_avg_us"), v.get("hbm_bytes_per_env"), v.get("algorithmic_bytes_per_env"),
        (v.get("valu") or {}).get("wave_instr_per_env_step")))
PY
