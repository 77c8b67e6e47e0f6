#!/bin/bash
# round-2 GPU call 2: what bounds the 6-DoF step kernel?  micro-benchmarks, scheduler variants A/B, SQ counters, wave stamps
OUT=gpurun_out
mkdir -p $OUT
ROOT=${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 120 ./tools/valu_dep > $OUT/r2_valu_dep.log 2>&1; rc=$?; echo "valu_dep rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 120 rocprofv3 --list-avail > $OUT/r2_counters_avail.txt 2>&1; echo "list rc=$?"
timeout -k 10 600 python tools/ab_bench.py --rounds 2 --arm base:: --arm ilp:ilp: --arm ilpw3:ilpw3: --arm base_c1::"--chains 1" --arm ilpw3_c1:ilpw3:"--chains 1" > $OUT/r2_ab2.log 2>&1
rc=$?; echo "ab rc=$rc"; tail -7 $OUT/r2_ab2.log; if [ $rc -ge 124 ]; then exit $rc; fi
MVRL_LIB=$ROOT/variants_build/libmvrl_stamp.so timeout -k 10 200 python tools/stamp_probe.py > $OUT/r2_stamp.log 2>&1; rc=$?; echo "stamp rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload c4 --no-cpu-baseline --chains 1 --steps 200 --warmup 20 --repeats 1 --prewarm-s 0.2"
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_WAVES_EQ_64" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $ROOT/$OUT/r2_sq_$i -- python3 $ARGS > $ROOT/$OUT/r2_sq_$i.json 2> $ROOT/$OUT/r2_sq_$i.err
  rc=$?; echo "pmc group $i ($grp) rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
  find $ROOT/$OUT/r2_sq_$i -name "*_kernel_trace.csv" -delete
done
