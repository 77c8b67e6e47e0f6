#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
for w in "" 4 3 2; do
if [ -n "$w" ]; then export MVRL_JIT_MIN_WAVES=$w; else unset MVRL_JIT_MIN_WAVES; fi; timeout -k 10 300 python bench.py --no-cpu-baseline --repeats 5 --flavour generic --specialize > $OUT/r2_bench70.log 2>&1; rc=$?
python - "$w" <<'PY'
import json,sys
j=json.loads(open('gpurun_out/r2_bench70.log').read().strip().splitlines()[-1])
print('min waves', sys.argv[1] or 'auto', '| us/step %.1f'%(j['ms_per_step']*1e3), j['config']['kernel'])
PY
if [ $rc -ge 124 ]; then exit $rc; fi
done
