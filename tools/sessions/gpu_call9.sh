#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
ROOT=${GRAFT_REPO_ROOT:-$PWD}
# N = 2 rehearsal of the multi-rank bench path on ONE card: gloo instead of RCCL (RCCL refuses two ranks on one device)
MVRL_BENCH_BACKEND=gloo MVRL_BENCH_SAME_DEVICE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --envs-per-gpu 262144 --repeats 3 > $OUT/r2_n2_rehearsal.log 2>&1
rc=$?; echo "n2 rehearsal rc=$rc"; tail -c 1500 $OUT/r2_n2_rehearsal.log; if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/r2_bench_driver3.log 2>&1; echo "bench driver-args rc=$?"; python - <<'PY'
import json
j=json.loads(open('gpurun_out/r2_bench_driver3.log').read().strip().splitlines()[-1])
print('driver args:', j['value'], j['ms_per_step'], j['timing']['ms_per_step_repeats'], j['roofline']['frac'], j['roofline']['single_launch'], j['roofline']['valu'])
PY
bash tools/bench_table.sh > $OUT/r2_table.log 2>&1; rc=$?; echo "table rc=$rc"; cat $OUT/r2_table.log; if [ $rc -ge 124 ]; then exit $rc; fi
bash tools/profile_round.sh c4 r02_c4 > $OUT/r2_prof_c4.log 2>&1; rc=$?; echo "profile c4 rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
bash tools/profile_round.sh auv r02_auv > $OUT/r2_prof_auv.log 2>&1; rc=$?; echo "profile auv rc=$rc"
