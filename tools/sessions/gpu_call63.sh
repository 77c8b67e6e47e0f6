#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
MVRL_JIT_COMPILER=hiprtc timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "specialised or flavours" > $OUT/r2_t63.log 2>&1
rc=$?; echo "hiprtc fallback pytest rc=$rc"; tail -4 $OUT/r2_t63.log | cut -c1-300
MVRL_HIPCC=/nonexistent/hipcc timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "specialised" > $OUT/r2_t63b.log 2>&1
rc=$?; echo "missing hipcc -> fallback pytest rc=$rc"; tail -3 $OUT/r2_t63b.log | cut -c1-300
