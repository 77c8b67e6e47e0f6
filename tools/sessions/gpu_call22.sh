#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
rm -f $OUT/r2_errq22.log
for a in "65536 25 4 0" "65536 25 8 0" "65536 40 2 0"; do
timeout -k 10 300 python tests/audit/err_quantiles.py $a >> $OUT/r2_errq22.log 2>&1; rc=$?
if [ $rc -ge 124 ]; then exit $rc; fi
done
grep -v "amdgpu.ids\|^   sensitive" $OUT/r2_errq22.log | cut -c1-250
for a in "" "--chains 1" "--workload c3"; do
timeout -k 10 300 python bench.py --no-cpu-baseline $a > $OUT/r2_bench22.log 2>&1; rc=$?
python - "$a" <<'PY'
import json,sys
j=json.loads(open('gpurun_out/r2_bench22.log').read().strip().splitlines()[-1])
print(sys.argv[1], '| %.3e'%j['value'], 'us/step %.1f'%(j['ms_per_step']*1e3), 'frac %.3f'%j['roofline']['frac'], j['config']['launch'][-60:])
PY
if [ $rc -ge 124 ]; then exit $rc; fi
done
