#!/bin/bash
# Executed VALU instructions per wave and env step (SQ_INSTS_VALU / SQ_WAVES) of library variants, on the GPU box:
#   bash tools/valu_count.sh <workload> <variant> [<variant> ...]      ('' or "default" = the in-tree libmvrl.so)
# One rocprofv3 --pmc pass per variant (60 single-launch steps); prints one line per variant.
WL=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = "default" ]; then unset MVRL_LIB; else export MVRL_LIB=$ROOT/variants_build/libmvrl_$v.so; fi
  d=$OUT/valu_${WL}_$v; rm -rf $d
  timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $d -- python3 $ROOT/bench.py --workload $WL --no-cpu-baseline --chains 1 --launch single --steps 60 --warmup 10 --repeats 1 --prewarm-s 0.1 > $d.json 2> $d.err
  rc=$?; if [ $rc -ge 124 ]; then echo "$v: timeout"; exit $rc; fi
  python3 - "$d" "$v" <<'PY'
import csv, glob, sys
acc = {}
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "step_kernel" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
if "SQ_WAVES" in acc:
    w = sum(acc["SQ_WAVES"]) / len(acc["SQ_WAVES"]); i = sum(acc["SQ_INSTS_VALU"]) / len(acc["SQ_INSTS_VALU"])
    print("%-12s SQ_INSTS_VALU / SQ_WAVES = %.1f  (%d launches, %d waves)" % (sys.argv[2], i / w, len(acc["SQ_WAVES"]), w), flush=True)
else:
    print(sys.argv[2], "no counters", flush=True)
PY
  find $d -name "*.csv" -delete 2>/dev/null
done
