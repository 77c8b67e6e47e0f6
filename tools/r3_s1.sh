#!/bin/bash
# round 3 call 1: launcher test + baseline numbers on today's box
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_bench_contract.py -x -q -m gpu > gpurun_out/r3_s1_contract.log 2>&1; echo "contract rc=$?"
tail -3 gpurun_out/r3_s1_contract.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3_s1_driver.json 2>gpurun_out/r3_s1_driver.err; echo "driver rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline --repeats 5 > gpurun_out/r3_s1_c4.json 2>gpurun_out/r3_s1_c4.err; echo "c4 rc=$?"
timeout -k 10 300 python bench.py --workload c2 --no-cpu-baseline --repeats 5 > gpurun_out/r3_s1_c2.json 2>gpurun_out/r3_s1_c2.err; echo "c2 rc=$?"
timeout -k 10 300 python bench.py --workload c3 --no-cpu-baseline --repeats 5 > gpurun_out/r3_s1_c3.json 2>gpurun_out/r3_s1_c3.err; echo "c3 rc=$?"
python - <<'PY'
import json
for n in ("driver","c4","c2","c3"):
    try:
        j=json.loads(open(f"gpurun_out/r3_s1_{n}.json").read().strip().splitlines()[-1])
        r=j["roofline"]; s=r.get("single_launch") or {}
        print(n, "%.3e"%j["value"], "us/step %.1f"%(j["ms_per_step"]*1e3), "frac %.3f"%r["frac"], "single", s.get("kernel_us_per_launch"), j.get("launch_plan"))
    except Exception as e: print(n,"FAILED",e)
PY
