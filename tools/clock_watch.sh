#!/bin/bash
# Sample shader clock / power with rocm-smi while a long bench run is in flight (DVFS diagnosis, read-only).
#   tools/clock_watch.sh <lib.so|""> <steps> [extra bench args]
lib=$1; steps=$2; shift 2
if [ -n "$lib" ]; then export MVRL_LIB=$lib; fi
python bench.py --no-cpu-baseline --steps $steps --warmup 50 --repeats 5 "$@" > /tmp/cw_bench.json 2>/dev/null &
pid=$!
sleep 5
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | sed 's/.*: *//' | tr '\n' ' '; echo
  sleep 1
done
wait $pid
python -c "
import json
d=json.loads(open('/tmp/cw_bench.json').read().strip().splitlines()[-1])
print('lib=$lib', 'steps', d['steps'], 'value %.3e' % d['value'], 'us/step %.1f' % d['roofline']['kernel_us_per_step'], 'args $*')
"
