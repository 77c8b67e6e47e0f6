#!/bin/bash
# Where does the one-launch-per-step plan lose its ~11 us per step?  Per-dispatch begin / end timestamps of the step kernel under
# `--launch single` (rocprofv3 --kernel-trace), condensed by tools/trace_series.py: kernel duration vs the gap to the next dispatch.
#   bash tools/gap_probe.sh <tag> [MVRL_LIB path] [extra bench args]
TAG=${1:-gap}; LIB=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
if [ -n "$LIB" ]; then export MVRL_LIB=$ROOT/$LIB; fi
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/gap_$TAG -- python3 $ROOT/bench.py --workload c4 --no-cpu-baseline --launch single \
   --steps 400 --warmup 50 --repeats 3 --prewarm-s 0.2 "$@" > $OUT/gap_$TAG.json 2> $OUT/gap_$TAG.err
rc=$?; echo "rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
python3 $ROOT/tools/trace_series.py $OUT/gap_$TAG
python3 -c "
import json,sys
j=json.loads(open('$OUT/gap_$TAG.json').read().strip().splitlines()[-1])
print('bench under the profiler: us/step', j['ms_per_step']*1e3, 'kernel_us_per_step', j['roofline']['kernel_us_per_step'])"
find $OUT/gap_$TAG -name "*_kernel_trace.csv" -delete
