#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python tools/ab_bench.py --rounds 3 --arm bounded:: --arm extrap::"MVRL_FLOW_EXTRAPOLATE=1" --arm bounded1::"--chains 1 --launch single" --arm extrap1::"MVRL_FLOW_EXTRAPOLATE=1 --chains 1 --launch single" > gpurun_out/r3_s13_ab.log 2>&1; echo "ab rc=$?"
tail -6 gpurun_out/r3_s13_ab.log
cd /tmp && export TMPDIR=/tmp
export MVRL_FLOW_EXTRAPOLATE=1
timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/valu_extrap -- python3 $GRAFT_REPO_ROOT/bench.py --workload c4 --no-cpu-baseline --chains 1 --launch single --steps 60 --warmup 10 --repeats 1 --prewarm-s 0.1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os
acc = {}
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/valu_extrap/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "step_kernel" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print("extrapolating composition: SQ_INSTS_VALU / SQ_WAVES = %.1f" % (sum(acc["SQ_INSTS_VALU"]) / sum(acc["SQ_WAVES"])))
PY
