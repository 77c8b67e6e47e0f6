#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
run() {
timeout -k 10 300 python bench.py --no-cpu-baseline --repeats 5 "$@" > $OUT/r2_bench50.log 2>&1; rc=$?
python - "$*" <<'PY'
import json,sys,os
try:
    j=json.loads(open('gpurun_out/r2_bench50.log').read().strip().splitlines()[-1])
    print('force', os.environ.get('MVRL_JIT_FORCE','-'), 'compiler', os.environ.get('MVRL_JIT_COMPILER','auto'), sys.argv[1], '| us/step %.1f'%(j['ms_per_step']*1e3), j['config']['kernel'])
except Exception as e:
    print('FAILED', open('gpurun_out/r2_bench50.log').read()[-700:])
PY
return 0
}
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "specialised" 2>&1 | tail -3
run
MVRL_JIT_FORCE=1 run --specialize
MVRL_JIT_FORCE=1 MVRL_JIT_COMPILER=hiprtc run --specialize
run --flavour sym
run --flavour sym --specialize
run --flavour ctrl --specialize
run --flavour generic --specialize
