#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $OUT/r2_t53.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 $OUT/r2_t53.log | cut -c1-300
if [ $rc -ge 124 ]; then exit $rc; fi
for a in "--workload auv" "--workload auv --chains 1" "" "--chains 1" "--workload auvcyl" "--workload loop --steps 1000 --warmup 50"; do
timeout -k 10 300 python bench.py --no-cpu-baseline --repeats 5 $a > $OUT/r2_bench53.log 2>&1; rc=$?
python - "$a" <<'PY'
import json,sys
try:
    j=json.loads(open('gpurun_out/r2_bench53.log').read().strip().splitlines()[-1])
    print(sys.argv[1], '| %.3e'%j['value'], 'us/step %.1f'%(j['ms_per_step']*1e3), 'frac %.3f'%j['roofline']['frac'], j['config']['kernel'])
except Exception as e:
    print('FAILED', open('gpurun_out/r2_bench53.log').read()[-500:])
PY
if [ $rc -ge 124 ]; then exit $rc; fi
done
