#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
for a in "--steps 1 --warmup 0" "--steps 3 --warmup 1" "--steps 50 --warmup 10" "--gpus 1 --steps 20 --warmup 5"; do
timeout -k 10 300 python bench.py --no-cpu-baseline $a > $OUT/r2_bench62.log 2>&1; rc=$?
python - "$a" $rc <<'PY'
import json,sys
try:
    lines=[l for l in open('gpurun_out/r2_bench62.log').read().strip().splitlines() if l.startswith('{')]
    j=json.loads(lines[-1])
    print(sys.argv[1], 'rc', sys.argv[2], '| json lines', len(lines), 'steps', j['steps'], 'warmup', j['warmup'], 'us/step %.1f'%(j['ms_per_step']*1e3), 'frac %.3f'%j['roofline']['frac'], 'n_gpus', j['n_gpus'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e, open('gpurun_out/r2_bench62.log').read()[-400:])
PY
done
