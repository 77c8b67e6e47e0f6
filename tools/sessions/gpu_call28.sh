#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "specialised" > $OUT/r2_t28.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -12 $OUT/r2_t28.log | cut -c1-300
if [ $rc -ge 124 ]; then exit $rc; fi
for a in "--flavour generic --specialize" "--flavour sym --specialize" "--flavour ctrl --specialize" "--flavour generic"; do
timeout -k 10 300 python bench.py --no-cpu-baseline --repeats 5 $a > $OUT/r2_bench28.log 2>&1; rc=$?
python - "$a" <<'PY'
import json,sys
try:
    j=json.loads(open('gpurun_out/r2_bench28.log').read().strip().splitlines()[-1])
    print(sys.argv[1], '| %.3e'%j['value'], 'us/step %.1f'%(j['ms_per_step']*1e3), 'frac %.3f'%j['roofline']['frac'], j['config']['kernel'])
except Exception as e:
    print(sys.argv[1], 'FAILED', open('gpurun_out/r2_bench28.log').read()[-600:])
PY
if [ $rc -ge 124 ]; then exit $rc; fi
done
MVRL_JIT_MIN_WAVES=3 timeout -k 10 300 python bench.py --no-cpu-baseline --repeats 5 --flavour generic --specialize > $OUT/r2_bench28.log 2>&1; python - <<'PY'
import json
j=json.loads(open('gpurun_out/r2_bench28.log').read().strip().splitlines()[-1])
print('generic jit 3 waves | %.3e'%j['value'], 'us/step %.1f'%(j['ms_per_step']*1e3), j['config']['kernel'])
PY
