#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -q > $OUT/r2_t61.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 $OUT/r2_t61.log | cut -c1-300
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 200 python __graft_entry__.py smoke > $OUT/r2_smoke61.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/r2_smoke61.log
timeout -k 10 300 python bench.py > $OUT/r2_bench_default61.log 2>&1; echo "bench rc=$?"; python - <<'PY'
import json
j=json.loads(open('gpurun_out/r2_bench_default61.log').read().strip().splitlines()[-1])
r=j['roofline']
print('default:', '%.3e'%j['value'], '%.1f us'%(j['ms_per_step']*1e3), 'frac %.3f'%r['frac'], 'valu', r['valu'] and round(r['valu']['frac'],3), 'traffic', r['traffic'] and round(r['traffic']/1e6,1), 'hash', r['kernel_source_hash'], 'cpu', '%.2e'%j['cpu_baseline']['value'])
PY
