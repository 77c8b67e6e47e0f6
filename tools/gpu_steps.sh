#!/bin/bash
# Run a list of GPU steps one after another on the GPU box, each under its own `timeout -k 10`, logging to gpurun_out/<name>.log.
# A step that FAILS lets the next one run; a step that TIMES OUT (or is killed) stops the list - no GPU step is started after a hang.
#   tools/gpu_steps.sh "name|seconds|command" ...
cd "${GRAFT_REPO_ROOT:-.}" && mkdir -p gpurun_out
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; secs="${rest%%|*}"; cmd="${rest#*|}"
  echo "== $name (limit ${secs}s): $cmd"
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1; rc=$?
  echo "== $name rc=$rc; tail:"; tail -n 6 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== $name timed out: stopping"; exit $rc; fi
done
