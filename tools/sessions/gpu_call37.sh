#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -q > $OUT/r2_t37.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -6 $OUT/r2_t37.log | cut -c1-300
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 200 python __graft_entry__.py smoke > $OUT/r2_smoke37.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/r2_smoke37.log
for a in "" "--workload c2" "--flavour generic --specialize"; do
timeout -k 10 300 python bench.py --no-cpu-baseline --repeats 5 $a > $OUT/r2_bench37.log 2>&1; rc=$?
python - "$a" <<'PY'
import json,sys
j=json.loads(open('gpurun_out/r2_bench37.log').read().strip().splitlines()[-1])
print(sys.argv[1], '| %.3e'%j['value'], 'us/step %.1f'%(j['ms_per_step']*1e3), 'frac %.3f'%j['roofline']['frac'], j['config']['kernel'])
PY
if [ $rc -ge 124 ]; then exit $rc; fi
done
