#!/bin/bash
# round-2 GPU call 3: LDS parking -> 4 waves/SIMD: parity suite, then A/B against the 3-wave build
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/r2_t3.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -8 $OUT/r2_t3.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 700 python tools/ab_bench.py --rounds 2 --arm park:: --arm nopark:nopark: --arm nopark_w4:nopark_w4: --arm park6k:park6k: --arm park_c1::"--chains 1" --arm nopark_c1:nopark:"--chains 1" --arm park_c4::"--chains 4" > $OUT/r2_ab3.log 2>&1
rc=$?; echo "ab rc=$rc"; tail -9 $OUT/r2_ab3.log
if [ $rc -ge 124 ]; then exit $rc; fi
for wl in c3 c2 auv; do timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --repeats 3 > $OUT/r2_b3_$wl.json 2>$OUT/r2_b3_$wl.err; echo "$wl rc=$?"; python -c "
import json,sys
j=json.loads(open('$OUT/r2_b3_$wl.json').read().strip().splitlines()[-1]); print('$wl', j['value'], j['ms_per_step']*1e3, j['roofline']['frac'])"; done
timeout -k 10 200 python bench.py --workload c4 --control-mode zoh --no-cpu-baseline --repeats 3 > $OUT/r2_b3_zoh.json 2>$OUT/r2_b3_zoh.err; python -c "
import json,sys
j=json.loads(open('$OUT/r2_b3_zoh.json').read().strip().splitlines()[-1]); print('zoh', j['value'], j['ms_per_step']*1e3, j['roofline']['frac'])"
