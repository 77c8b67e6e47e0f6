#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
for lib in "" variants_build/libmvrl_fullcells.so "" variants_build/libmvrl_fullcells.so; do
for a in "" "--rollout --steps 2000 --warmup 96"; do
MVRL_LIB=${lib:+$PWD/$lib} timeout -k 10 300 python bench.py --no-cpu-baseline --repeats 5 $a > $OUT/r2_bench55.log 2>&1; rc=$?
python - "$lib $a" <<'PY'
import json,sys
j=json.loads(open('gpurun_out/r2_bench55.log').read().strip().splitlines()[-1])
print(sys.argv[1], '| us/step %.1f'%(j['ms_per_step']*1e3), j['config']['kernel'])
PY
if [ $rc -ge 124 ]; then exit $rc; fi
done; done
