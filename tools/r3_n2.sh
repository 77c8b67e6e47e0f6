#!/bin/bash
# the plain N = 2 command on the one-GPU box: two ranks share the card over gloo (rehearsal knobs); everything else as the driver's N > 1 runs
cd $GRAFT_REPO_ROOT
MVRL_BENCH_BACKEND=gloo MVRL_BENCH_SAME_DEVICE=1 timeout -k 10 600 python bench.py --gpus 2 --steps 20 --warmup 5 --envs-per-gpu 524288 > gpurun_out/r3_n2_rehearsal.json 2> gpurun_out/r3_n2_rehearsal.err; echo "rc=$?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r3_n2_rehearsal.json").read().strip().splitlines()[-1])
print({k: j[k] for k in ("n_gpus", "value", "ms_per_step", "launch_plan", "scaling")})
print(j["rccl"]); print(j["with_gather"]); print(j["scaling_claim"])
PY
