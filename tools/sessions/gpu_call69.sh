#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
MVRL_BENCH_BACKEND=gloo MVRL_BENCH_SAME_DEVICE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --envs-per-gpu 262144 --repeats 3 > $OUT/r2_n2_rehearsal69.log 2>&1
rc=$?; echo "n2 rehearsal rc=$rc"; python - <<'PY'
import json
lines=[l for l in open('gpurun_out/r2_n2_rehearsal69.log').read().splitlines() if l.startswith('{')]
j=json.loads(lines[-1])
print('json lines', len(lines), 'n_gpus', j['n_gpus'], 'value %.3e'%j['value'], 'scaling', j['scaling'], 'with_gather', {k:(round(v,3) if isinstance(v,float) else v) for k,v in j.get('with_gather',{}).items() if k in ('value','ms_per_step','error')}, 'rccl', j.get('rccl'))
PY
