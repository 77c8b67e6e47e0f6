#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
run() {
timeout -k 10 300 python bench.py --no-cpu-baseline --repeats 5 --chains 1 "$@" > $OUT/r2_bench49.log 2>&1; rc=$?
python - "$*" <<'PY'
import json,sys,os
try:
    j=json.loads(open('gpurun_out/r2_bench49.log').read().strip().splitlines()[-1])
    print('file', os.path.basename(os.environ.get('MVRL_JIT_CODE_FILE','-')), 'loader', os.environ.get('MVRL_JIT_LOADER','static'), sys.argv[1], '| us/step %.1f'%(j['ms_per_step']*1e3), j['config']['kernel'])
except Exception as e:
    print('FAILED', open('gpurun_out/r2_bench49.log').read()[-700:])
PY
return 0
}
run
export MVRL_JIT_FORCE=1 MVRL_JIT_LOADER=module
run --specialize
MVRL_JIT_CODE_FILE=$PWD/variants_build/rov6_genco.co run --specialize
MVRL_JIT_CODE_FILE=$PWD/variants_build/rov6_genco_jit.co run --specialize
