#!/bin/bash
# Round 5, second sitting: the A/B runs behind profiles/r05_ab_second_session.txt - one experiment per call, every arm round-robin on ONE box
# (tools/ab_bench.py).  Variant libraries: python tools/variants.py build <names> in the build container first.
#   bash tools/r5b_ab.sh flags     PID increment branch vs select (incsel), fp64 twins with NaN / Inf / signed-zero bookkeeping (f64nofin), machine LICM off
#   bash tools/r5b_ab.sh flavours  fp64: literal constants (two s_mov_b32 per use) vs constants through scalar loads (ctrl / sym flavours)
#   bash tools/r5b_ab.sh vote      fall-back guards as plain exec guards (notrigvote) vs behind a wave vote (default), every config
#   bash tools/r5b_ab.sh exp2      the 3-DoF stage with the library expf (-DMVRL_LIB_EXP: variant libexp) vs exp2 (default); --graph / --rollout
#   bash tools/r5b_ab.sh blocks    one-wave workgroups (default) vs two / four waves per workgroup (blk128 / blk256)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
EXP=${1:-vote}
COMMON="--no-cpu-baseline --warmup 100 --repeats 3"
case "$EXP" in
flags)
  ARMS=(--arm "base64::--workload c4 --steps 1000 --precision f64" --arm "incsel64:incsel:--workload c4 --steps 1000 --precision f64"
        --arm "nofin64:f64nofin:--workload c4 --steps 1000 --precision f64" --arm "nolicm64:f64nolicm:--workload c4 --steps 1000 --precision f64"
        --arm "base32::--workload c4 --steps 1000" --arm "incsel32:incsel:--workload c4 --steps 1000") ;;
flavours)
  ARMS=(--arm "baked::--workload c4 --steps 1000 --precision f64" --arm "ctrl::--workload c4 --steps 1000 --precision f64 --flavour ctrl"
        --arm "sym::--workload c4 --steps 1000 --precision f64 --flavour sym") ;;
vote)
  ARMS=(--arm "c4::--workload c4 --steps 1000" --arm "c4guard:notrigvote:--workload c4 --steps 1000"
        --arm "c4f64::--workload c4 --steps 500 --precision f64" --arm "c4f64guard:notrigvote:--workload c4 --steps 500 --precision f64"
        --arm "c3::--workload c3 --steps 2000" --arm "c3guard:notrigvote:--workload c3 --steps 2000"
        --arm "c2::--workload c2 --steps 4000" --arm "c2guard:notrigvote:--workload c2 --steps 4000") ;;
exp2)
  ARMS=(--arm "c2::--workload c2 --steps 4000" --arm "c2libexp:libexp:--workload c2 --steps 4000"
        --arm "c2graph::--workload c2 --steps 4000 --warmup 96 --graph" --arm "c2roll::--workload c2 --steps 4000 --warmup 96 --rollout") ;;
blocks)
  ARMS=(--arm "c4::--workload c4 --steps 1000" --arm "c4b128:blk128:--workload c4 --steps 1000" --arm "c4b256:blk256:--workload c4 --steps 1000"
        --arm "c3::--workload c3 --steps 2000" --arm "c3b128:blk128:--workload c3 --steps 2000" --arm "c3b256:blk256:--workload c3 --steps 2000"
        --arm "c2::--workload c2 --steps 4000" --arm "c2b128:blk128:--workload c2 --steps 4000" --arm "c2b256:blk256:--workload c2 --steps 4000"
        --arm "c4f64::--workload c4 --steps 500 --precision f64" --arm "c4f64b128:blk128:--workload c4 --steps 500 --precision f64") ;;
*) echo "usage: tools/r5b_ab.sh flags|flavours|vote|exp2|blocks"; exit 2 ;;
esac
timeout -k 10 1100 python tools/ab_bench.py --rounds 3 --common "$COMMON" "${ARMS[@]}" > gpurun_out/r5b_ab_$EXP.log 2>&1
echo "ab rc=$?"
tail -14 gpurun_out/r5b_ab_$EXP.log
