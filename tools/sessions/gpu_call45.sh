#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
run() {
timeout -k 10 300 python bench.py --no-cpu-baseline --repeats 5 "$@" > $OUT/r2_bench45.log 2>&1; rc=$?
python - "$*" <<'PY'
import json,sys,os
try:
    j=json.loads(open('gpurun_out/r2_bench45.log').read().strip().splitlines()[-1])
    print(sys.argv[1], '| us/step %.1f'%(j['ms_per_step']*1e3), j['config']['launch'][:50])
except Exception as e:
    print('FAILED', open('gpurun_out/r2_bench45.log').read()[-300:])
PY
return 0
}
run --chains 1
run --graph --warmup 48
run
run --workload auv --graph --warmup 48
run --workload auv --chains 1
