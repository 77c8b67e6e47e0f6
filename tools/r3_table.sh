#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/bench_table.sh > gpurun_out/r3_bench_table.txt 2>&1
cat gpurun_out/r3_bench_table.txt
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_driver_args.json 2> gpurun_out/r3_driver_args.err; echo "driver-args rc=$?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r3_driver_args.json").read().strip().splitlines()[-1])
r = j["roofline"]
print("driver args: value %.3e  ms/step %.4f  frac %.3f  single %s  plans %s  traffic %s  valu %s" % (
    j["value"], j["ms_per_step"], r["frac"], (r.get("single_launch") or {}).get("kernel_us_per_launch"),
    {k: round(v["ms_per_step"] * 1e3, 1) for k, v in j["timing"]["launch_plans"].items()}, r["traffic"], (r.get("valu") or {}).get("frac")))
print(j["cpu_baseline"])
PY
