#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $OUT/r2_t17.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 $OUT/r2_t17.log | cut -c1-300
if [ $rc -ge 124 ]; then exit $rc; fi
for a in "" "--chains 1" "--workload c3" "--control-mode zoh"; do
timeout -k 10 300 python bench.py --no-cpu-baseline $a > $OUT/r2_bench17.log 2>&1; rc=$?
python - "$a" <<'PY'
import json,sys
j=json.loads(open('gpurun_out/r2_bench17.log').read().strip().splitlines()[-1])
print(sys.argv[1], '| %.3e'%j['value'], 'us/step %.1f'%(j['ms_per_step']*1e3), 'frac %.3f'%j['roofline']['frac'], j['config']['launch'][-60:])
PY
if [ $rc -ge 124 ]; then exit $rc; fi
done
