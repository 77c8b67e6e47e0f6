B="python bench.py --workload c4 --precision f64 --steps 300 --warmup 30 --no-cpu-baseline"
bash tools/gpu_steps.sh \
 "r5_s2_gputests|1000|python -m pytest tests/ -x -q -m gpu" \
 "r5_c4in|200|python bench.py --workload c4in --steps 1000 --warmup 100 --no-cpu-baseline" \
 "r5_c4|200|python bench.py --workload c4 --steps 1000 --warmup 100 --no-cpu-baseline" \
 "r5_f64_ilp|200|MVRL_LIB=variants_build/libmvrl_f64ilp.so $B" \
 "r5_f64_occ|200|MVRL_LIB=variants_build/libmvrl_f64occ.so $B" \
 "r5_c3_f64|200|python bench.py --workload c3 --precision f64 --steps 300 --warmup 30 --no-cpu-baseline" \
 "r5_c2_f64|200|python bench.py --workload c2 --precision f64 --steps 300 --warmup 30 --no-cpu-baseline"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5_c4*.log")+glob.glob("gpurun_out/r5_f64_*.log")+glob.glob("gpurun_out/r5_c[23]_f64.log")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "%.3e"%j["value"], "us/step %.1f"%(j["ms_per_step"]*1e3), "frac %.3f"%j["roofline"]["frac"], "single %.1f"%j["roofline"]["single_launch"]["kernel_us_per_launch"])
    except Exception as e: print(f, "ERR", e)
PY
