#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: kernel-trace stats + HBM traffic counters of the bench command.
#   bash tools/profile_round.sh <workload> <tag>
# Writes raw rocprofv3 output under gpurun_out/prof_<tag>_*; tools/summarize_profile.py condenses it into profiles/.
set -e
WL=${1:-c4}
TAG=${2:-r01_$WL}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload $WL --steps 100 --warmup 10 --no-cpu-baseline $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_kt -- python3 $ARGS > $OUT/prof_${TAG}_kt.json 2> $OUT/prof_${TAG}_kt.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_fetch -- python3 $ARGS > $OUT/prof_${TAG}_fetch.json 2> $OUT/prof_${TAG}_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_write -- python3 $ARGS > $OUT/prof_${TAG}_write.json 2> $OUT/prof_${TAG}_write.err
ls -R $OUT/prof_${TAG}_kt | head -20
