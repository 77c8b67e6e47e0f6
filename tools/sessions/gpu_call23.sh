#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
for a in "--launch chains --chains 2" "--launch chains --chains 3" "--launch chains --chains 4" "--launch chains --chains 2 --no-stagger" "--launch chains --chains 2 --steps 100 --warmup 20"; do
timeout -k 10 300 python bench.py --no-cpu-baseline $a > $OUT/r2_bench23.log 2>&1; rc=$?
python - "$a" <<'PY'
import json,sys
j=json.loads(open('gpurun_out/r2_bench23.log').read().strip().splitlines()[-1])
print(sys.argv[1], '| %.3e'%j['value'], 'us/step %.1f'%(j['ms_per_step']*1e3), 'frac %.3f'%j['roofline']['frac'])
PY
if [ $rc -ge 124 ]; then exit $rc; fi
done
