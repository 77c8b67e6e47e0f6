#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: kernel-trace stats + HBM traffic counters of the bench command.
#   bash tools/profile_round.sh <workload> <tag>
# Pass 1 (--kernel-trace --stats) profiles the SAME command the bench line comes from (default steps / warm-up);
# the two counter passes (FETCH_SIZE, WRITE_SIZE - separately, as MI355X_MICROARCH.md prescribes) only need per-launch
# byte counts, so they run 300 steps.  Raw output lands under gpurun_out/prof_<tag>_*; tools/summarize_profile.py
# condenses it into profiles/.
set -e
WL=${1:-c4}
TAG=${2:-r01_$WL}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload $WL --no-cpu-baseline $EXTRA"
PMC_ARGS="$ARGS --steps 300 --warmup 30"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_kt -- python3 $ARGS > $OUT/prof_${TAG}_kt.json 2> $OUT/prof_${TAG}_kt.err
# the per-dispatch trace of a 5500-launch run is large: keep the stats tables only
find $OUT/prof_${TAG}_kt -name "*_kernel_trace.csv" -delete
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_fetch -- python3 $PMC_ARGS > $OUT/prof_${TAG}_fetch.json 2> $OUT/prof_${TAG}_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_write -- python3 $PMC_ARGS > $OUT/prof_${TAG}_write.json 2> $OUT/prof_${TAG}_write.err
find $OUT/prof_${TAG}_fetch $OUT/prof_${TAG}_write -name "*_kernel_trace.csv" -delete
tail -c 600 $OUT/prof_${TAG}_kt.json
