#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
MVRL_BENCH_BACKEND=gloo MVRL_BENCH_SAME_DEVICE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --envs-per-gpu 262144 --repeats 3 > $OUT/r2_n2_rehearsal.log 2>&1
rc=$?; echo "n2 rehearsal rc=$rc"; tail -c 2500 $OUT/r2_n2_rehearsal.log; if [ $rc -ge 124 ]; then exit $rc; fi
for i in 1 2; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r2_bench_driver4.log 2>&1; echo "bench driver-args rc=$?"; python - <<'PY'
import json
j=json.loads(open('gpurun_out/r2_bench_driver4.log').read().strip().splitlines()[-1])
print('driver args:', '%.3e'%j['value'], j['ms_per_step'], [round(x*1e3,1) for x in j['timing']['ms_per_step_repeats']], 'frac %.3f'%j['roofline']['frac'], 'single %.1f'%j['roofline']['single_launch']['kernel_us_per_launch'], 'valu', (j['roofline']['valu'] or {}).get('frac'), 'traffic', j['roofline']['traffic'])
PY
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stagger > $OUT/r2_bench_driver5.log 2>&1; python - <<'PY'
import json
j=json.loads(open('gpurun_out/r2_bench_driver5.log').read().strip().splitlines()[-1])
print('driver args no-stagger:', '%.3e'%j['value'], j['ms_per_step'], [round(x*1e3,1) for x in j['timing']['ms_per_step_repeats']], 'frac %.3f'%j['roofline']['frac'])
PY
timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/r2_bench_default6.log 2>&1; python - <<'PY'
import json
j=json.loads(open('gpurun_out/r2_bench_default6.log').read().strip().splitlines()[-1])
print('default:', '%.3e'%j['value'], j['ms_per_step'], 'frac %.3f'%j['roofline']['frac'], 'single %.1f'%j['roofline']['single_launch']['kernel_us_per_launch'], j['roofline']['valu'])
PY
timeout -k 10 600 python -m pytest tests/test_gpu_chains.py tests/test_gpu_api.py -m gpu -q > $OUT/r2_t10.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r2_t10.log
