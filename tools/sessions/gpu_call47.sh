#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
MVRL_BENCH_BACKEND=gloo MVRL_BENCH_SAME_DEVICE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --envs-per-gpu 262144 --repeats 3 > $OUT/r2_n2_rehearsal47.log 2>&1
rc=$?; echo "n2 rehearsal rc=$rc"; tail -c 1800 $OUT/r2_n2_rehearsal47.log; if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/r2_bench_driver47.log 2>&1; echo "bench driver-args rc=$?"; python - <<'PY'
import json
j=json.loads(open('gpurun_out/r2_bench_driver47.log').read().strip().splitlines()[-1])
print('driver args:', '%.3e'%j['value'], '%.1f us'%(j['ms_per_step']*1e3), 'frac %.3f'%j['roofline']['frac'], 'valu', (j['roofline']['valu'] or {}).get('frac'), 'traffic', j['roofline']['traffic'], 'cpu', j['cpu_baseline'])
PY
