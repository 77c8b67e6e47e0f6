#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: the rocprofv3 evidence behind bench.py's roofline objects.
#   bash tools/profile_round.sh <workload> <tag> [extra bench args]
# Pass 1 (--kernel-trace --stats) profiles the bench command under its `chains` launch plan (defaults: repeats, K), pass 1b
# under the `single` plan (--chains 1); bench.py's default (--launch auto) times both plans and reports the faster.
# The counter passes (one rocprofv3 run per group, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE separately)
# need per-launch values only: 200 steps, one launch per step (--chains 1), so that one dispatch = one env step of the batch.
# Raw output lands under gpurun_out/prof_<tag>_*; tools/summarize_counters.py condenses it into profiles/.
# `--specialize` arguments are fine under the profiler: the hipcc child of mvrl_specialize gets a scrubbed environment (no
# LD_PRELOAD / ROCP_* / HSA_TOOLS_*: mvrl_abi.hip child_environment), so the tool library does not follow into the compiler.
WL=${1:-c4}
TAG=${2:-r04_$WL}
shift 2
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload $WL --no-cpu-baseline $*"
PMC_ARGS="$ARGS --chains 1 --steps 200 --warmup 20 --repeats 1 --prewarm-s 0.2"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_kt -- python3 $ARGS --launch chains > $OUT/prof_${TAG}_kt.json 2> $OUT/prof_${TAG}_kt.err
rc=$?; echo "kt rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
python3 $ROOT/tools/trace_median.py $OUT/prof_${TAG}_kt      # per-kernel median / p10 / p90 before the trace goes
find $OUT/prof_${TAG}_kt -name "*_kernel_trace.csv" -delete     # the per-dispatch trace of a 10k-launch run is large: keep the stats tables
# the same bench command with one launch per step: its per-kernel average is what bench.py's roofline.single_launch reports
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_kt1 -- python3 $ARGS --chains 1 > $OUT/prof_${TAG}_kt1.json 2> $OUT/prof_${TAG}_kt1.err
rc=$?; echo "kt1 rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
python3 $ROOT/tools/trace_median.py $OUT/prof_${TAG}_kt1
find $OUT/prof_${TAG}_kt1 -name "*_kernel_trace.csv" -delete
if [ "${MVRL_PROFILE_PMC:-1}" = "0" ]; then tail -c 300 $OUT/prof_${TAG}_kt.json; exit 0; fi   # kernel-trace passes only
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/prof_${TAG}_pmc$i -- python3 $PMC_ARGS > $OUT/prof_${TAG}_pmc$i.json 2> $OUT/prof_${TAG}_pmc$i.err
  rc=$?; echo "pmc group $i ($grp) rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
  find $OUT/prof_${TAG}_pmc$i -name "*_kernel_trace.csv" > $OUT/prof_${TAG}_pmc$i.tracefiles
done
tail -c 400 $OUT/prof_${TAG}_kt.json
