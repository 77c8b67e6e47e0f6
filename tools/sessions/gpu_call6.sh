#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
ROOT=${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 1100 python -m pytest tests -m gpu -q > $OUT/r2_t6.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -12 $OUT/r2_t6.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 200 python __graft_entry__.py smoke > $OUT/r2_smoke6.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/r2_smoke6.log
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload auv --no-cpu-baseline --chains 1 --steps 200 --warmup 20 --repeats 1 --prewarm-s 0.2"
for v in default auvskip1 auvskip2 auvskip4 auvskip8; do
  for c in WRITE_SIZE FETCH_SIZE; do
    if [ $v = default ]; then unset MVRL_LIB; else export MVRL_LIB=$ROOT/variants_build/libmvrl_$v.so; fi
    timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $ROOT/$OUT/r2_auvpmc_${v}_$c -- python3 $ARGS > $ROOT/$OUT/r2_auvpmc_${v}_$c.json 2> $ROOT/$OUT/r2_auvpmc_${v}_$c.err
    rc=$?; echo "pmc $v $c rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
    find $ROOT/$OUT/r2_auvpmc_${v}_$c -name "*_kernel_trace.csv" -delete
  done
done
