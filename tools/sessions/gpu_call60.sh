#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
for a in "--workload c4" "--workload auv" "--workload c2"; do
timeout -k 10 600 python bench.py --no-cpu-baseline --repeats 1 --steps 100000 --warmup 100 $a > $OUT/r2_soak60.log 2>&1; rc=$?
python - "$a" <<'PY'
import json,sys
j=json.loads(open('gpurun_out/r2_soak60.log').read().strip().splitlines()[-1])
print(sys.argv[1], '| steps', j['steps'], 'us/step %.1f'%(j['ms_per_step']*1e3), 'finite', j.get('outputs_finite'), 'episodes/env ~', j['steps']//250)
PY
if [ $rc -ge 124 ]; then exit $rc; fi
done
