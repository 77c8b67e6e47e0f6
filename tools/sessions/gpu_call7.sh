#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
ROOT=${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 600 python tests/audit/margin_probe.py > $OUT/r2_margins4.log 2>&1; rc=$?; echo "margins rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
N=16384 STEPS=60 timeout -k 10 900 python tests/audit/margin_probe.py > $OUT/r2_margins5.log 2>&1; rc=$?; echo "margins5 rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 1100 python -m pytest tests -m gpu -q -s > $OUT/r2_t7.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -8 $OUT/r2_t7.log | cut -c1-300
if [ $rc -ge 124 ]; then exit $rc; fi
for a in "--workload auv" "--workload auvcyl" "--flavour sym" "--flavour generic" "--flavour generic --chains 1"; do
  timeout -k 10 200 python bench.py $a --no-cpu-baseline --repeats 3 > $OUT/r2_b7.json 2>$OUT/r2_b7.err; python -c "
import json
j=json.loads(open('$OUT/r2_b7.json').read().strip().splitlines()[-1]); print('$a', '%.3e'%j['value'], 'us/step %.1f'%(j['ms_per_step']*1e3), 'frac %.3f'%j['roofline']['frac'], j['config']['kernel'])"
done
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload auv --no-cpu-baseline --chains 1 --steps 200 --warmup 20 --repeats 1 --prewarm-s 0.2"
for c in WRITE_SIZE FETCH_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $ROOT/$OUT/r2_auvpmc2_$c -- python3 $ARGS > $ROOT/$OUT/r2_auvpmc2_$c.json 2> $ROOT/$OUT/r2_auvpmc2_$c.err
  rc=$?; echo "pmc $c rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
  find $ROOT/$OUT/r2_auvpmc2_$c -name "*_kernel_trace.csv" -delete
done
