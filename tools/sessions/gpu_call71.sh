#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
for rep in 1 2 3; do
for wl in c4 auv c3; do
for st in "" "--no-stagger"; do
timeout -k 10 300 python bench.py --no-cpu-baseline --repeats 7 --launch chains --workload $wl $st > $OUT/r2_bench71.log 2>&1; rc=$?
python - "$wl $st" <<'PY'
import json,sys
j=json.loads(open('gpurun_out/r2_bench71.log').read().strip().splitlines()[-1])
print(sys.argv[1], '| us/step %.2f'%(j['ms_per_step']*1e3))
PY
if [ $rc -ge 124 ]; then exit $rc; fi
done; done; done
