#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
for i in 1 2; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r2_bench_driver6.log 2>&1; echo "bench driver-args rc=$?"; python - <<'PY'
import json
j=json.loads(open('gpurun_out/r2_bench_driver6.log').read().strip().splitlines()[-1])
print('driver args:', '%.3e'%j['value'], '%.1f'%(j['ms_per_step']*1e3), 'frac %.3f'%j['roofline']['frac'], j['config']['launch'])
for pl,v in j['timing']['launch_plans'].items(): print('   ', pl, '%.1f'%(v['ms_per_step']*1e3), [round(x*1e3,1) for x in v['ms_per_step_repeats']])
PY
done
timeout -k 10 300 python bench.py > $OUT/r2_bench_default7.log 2>&1; python - <<'PY'
import json
j=json.loads(open('gpurun_out/r2_bench_default7.log').read().strip().splitlines()[-1])
print('default:', '%.3e'%j['value'], '%.1f'%(j['ms_per_step']*1e3), 'frac %.3f'%j['roofline']['frac'], j['config']['launch'], j['cpu_baseline'])
PY
MVRL_BENCH_BACKEND=gloo MVRL_BENCH_SAME_DEVICE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --envs-per-gpu 262144 --repeats 3 > $OUT/r2_n2_rehearsal2.log 2>&1
rc=$?; echo "n2 rehearsal rc=$rc"; tail -c 700 $OUT/r2_n2_rehearsal2.log
