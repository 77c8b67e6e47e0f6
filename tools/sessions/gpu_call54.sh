#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -q > $OUT/r2_t54.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 $OUT/r2_t54.log | cut -c1-300
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 200 python __graft_entry__.py smoke > $OUT/r2_smoke54.log 2>&1; echo "smoke rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/r2_bench_driver54.log 2>&1; echo "bench rc=$?"; python - <<'PY'
import json
j=json.loads(open('gpurun_out/r2_bench_driver54.log').read().strip().splitlines()[-1])
print('driver args:', '%.3e'%j['value'], '%.1f'%(j['ms_per_step']*1e3), 'frac %.3f'%j['roofline']['frac'], j['config']['launch'][:60], 'cpu', '%.2e'%j['cpu_baseline']['value'])
PY
bash tools/profile_round.sh c4 r02_c4 > $OUT/r2_prof_c4.log 2>&1; rc=$?; echo "profile c4 rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
bash tools/profile_round.sh auv r02_auv > $OUT/r2_prof_auv.log 2>&1; rc=$?; echo "profile auv rc=$rc"
timeout -k 10 1000 bash tools/bench_table.sh > $OUT/r2_table54.log 2>&1; echo "table rc=$?"; cat $OUT/r2_table54.log | cut -c1-190
