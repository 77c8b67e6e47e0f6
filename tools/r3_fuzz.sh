#!/bin/bash
# exploratory: the configuration sweep with other seeds (the committed test runs seed 2024)
cd $GRAFT_REPO_ROOT
for seed in 2024 1 2 3 4 5 6 7 8; do
  MVRL_FUZZ_SEED=$seed timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k config_fuzz -s > gpurun_out/r3_fuzz_$seed.log 2>&1; rc=$?
  echo "seed $seed rc=$rc: $(grep -c 'fuzz case' gpurun_out/r3_fuzz_$seed.log) case lines; $(grep -h -A1 'AssertionError' gpurun_out/r3_fuzz_$seed.log | head -2 | tr '\n' ' ' | cut -c1-200)"
done
