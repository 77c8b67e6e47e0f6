#!/bin/bash
# Executed instruction mix per wave and env step of the step kernel, on the GPU box: one rocprofv3 --pmc pass per counter group.
#   bash tools/valu_mix.sh <workload> "<bench args>" <variant> [<variant> ...]     ('default' = the in-tree libmvrl.so)
WL=$1; ARGS=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
GROUPS_=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64" "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU")
for v in "$@"; do
  if [ "$v" = "default" ]; then unset MVRL_LIB; else export MVRL_LIB=$ROOT/variants_build/libmvrl_$v.so; fi
  i=0
  for grp in "${GROUPS_[@]}"; do
    d=$OUT/mix_${WL}_${v}_$i; rm -rf $d
    timeout -k 10 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python3 $ROOT/bench.py --workload $WL $ARGS --no-cpu-baseline --chains 1 --launch single --steps 40 --warmup 10 --repeats 1 --prewarm-s 0.1 > $d.json 2> $d.err
    rc=$?; if [ $rc -ge 124 ]; then echo "$v group $i: timeout"; exit $rc; fi
    python3 - "$d" "$v" <<'PY'
import csv, glob, sys
acc = {}
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "step_kernel" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
n_waves = 16384.0
print(sys.argv[2], " ".join("%s=%.1f" % (k, sum(v) / len(v) / n_waves) for k, v in sorted(acc.items())), "(per wave of a 1 048 576-env launch)", flush=True)
PY
    find $d -name "*.csv" -delete 2>/dev/null
    i=$((i+1))
  done
done
