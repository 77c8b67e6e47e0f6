#!/bin/bash
# One line per workload / option of DESIGN.md section 5 (run on the GPU box): bash tools/bench_table.sh > gpurun_out/table.log
run() {
  timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --repeats 3 > /tmp/bt.json 2>/tmp/bt.err
  python - "$@" <<'PY'
import json, sys
try:
    j = json.loads(open('/tmp/bt.json').read().strip().splitlines()[-1])
    r = j['roofline']
    s = r.get('single_launch') or {}
    print('%-44s %.3e env-steps/s  %7.1f us/step  hbm frac %.3f  single-launch %s us  kernel %s' % (
        ' '.join(sys.argv[1:]), j['value'], j['ms_per_step'] * 1e3, r['frac'], ('%.1f' % s['kernel_us_per_launch']) if s else '-', j['config']['kernel']), flush=True)
except Exception as e:
    print(' '.join(sys.argv[1:]), 'FAILED', e, open('/tmp/bt.err').read()[-300:], flush=True)
PY
}
run --workload c4
run --workload c4 --launch chains
run --workload c4 --chains 1
run --workload c4 --steps 20 --warmup 5
run --workload c4 --control-mode zoh
run --workload c4 --n-substeps 8
run --workload c4 --precision f64 --steps 400 --warmup 40
run --workload c3 --precision f64 --steps 400 --warmup 40
run --workload c2 --precision f64 --steps 400 --warmup 40
run --workload c4in
run --workload c4in --chains 1
run --workload c4 --flavour ctrl
run --workload c4 --flavour sym
run --workload c4 --flavour generic
run --workload c4 --flavour sym --specialize
run --workload c4 --flavour generic --specialize
run --workload c4 --rollout --steps 2000 --warmup 96
run --workload c3
run --workload c3 --rollout --steps 2000 --warmup 96
run --workload c2
run --workload c2 --graph --steps 2000 --warmup 96
run --workload c2 --rollout --steps 2000 --warmup 96
run --workload auv
run --workload auv --envs-per-gpu 4194304 --steps 500 --warmup 50
run --workload auv --envs-per-gpu 4194304 --chains 1 --steps 500 --warmup 50
run --workload auv --rollout --steps 2000 --warmup 96
run --workload auvcyl
run --workload loop --steps 1000 --warmup 50
run --workload pdeval --steps 5000 --warmup 250
