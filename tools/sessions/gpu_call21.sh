#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -q > $OUT/r2_t21.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -6 $OUT/r2_t21.log | cut -c1-300
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 200 python __graft_entry__.py smoke > $OUT/r2_smoke21.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/r2_smoke21.log | cut -c1-400
