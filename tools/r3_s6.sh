#!/bin/bash
cd $GRAFT_REPO_ROOT
export MVRL_CPU_THREADS=16 OMP_NUM_THREADS=16
MVRL_LIB=$PWD/variants_build/libmvrl_prevz.so timeout -k 10 500 python tests/audit/err_quantiles.py 1048576 25 4 0 6 > gpurun_out/r3_s6_audit_prev.log 2>&1; echo "prev rc=$?"
timeout -k 10 500 python tests/audit/err_quantiles.py 1048576 25 4 0 6 > gpurun_out/r3_s6_audit_new.log 2>&1; echo "new rc=$?"
head -12 gpurun_out/r3_s6_audit_prev.log; head -12 gpurun_out/r3_s6_audit_new.log
