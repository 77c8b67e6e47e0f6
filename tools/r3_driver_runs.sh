#!/bin/bash
# Eight runs of the driver's command on one box: how much its 2.2-ms regions swing (profiles/r03_driver_args_runs.txt)
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8; do
  timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /tmp/da_$i.json 2>/dev/null
  python - $i <<'PY'
import json, sys
j = json.loads(open("/tmp/da_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
r = j["roofline"]; s = r["single_launch"]
reps = j["timing"]["launch_plans"]
print("run %s: value %.3e  chains %.1f us/step (frac %.3f; repeats %.1f..%.1f)  single %.1f us (frac %.3f; repeats %.1f..%.1f)" % (
    sys.argv[1], j["value"], j["ms_per_step"] * 1e3, r["frac"], min(reps["chains"]["ms_per_step_repeats"]) * 1e3, max(reps["chains"]["ms_per_step_repeats"]) * 1e3,
    s["kernel_us_per_launch"], s["frac"], min(reps["single"]["ms_per_step_repeats"]) * 1e3, max(reps["single"]["ms_per_step_repeats"]) * 1e3), flush=True)
PY
done
