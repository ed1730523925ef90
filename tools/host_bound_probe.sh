#!/bin/bash
# Is the in-flight loop bound by the host's launch loop (one core under the GIL)?  CPU seconds of the process per timed step next to the
# step time, for several interpreter switch intervals and scenes in flight: tools/host_bound_probe.sh > gpurun_out/host_bound.txt
for cfg in "0.5 3" "0.1 3" "5 3" "0.5 2" "0.5 4"; do
  set -- $cfg
  PASCO_BENCH_SWITCH_MS=$1 timeout 200 python bench.py --steps 48 --in-flight $2 --no-cpu-baseline --no-exact --no-configs --no-profile 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('switch interval $1 ms, $2 in flight: %.2f scenes/s, %.2f ms per step, host CPU %.2f ms per step, one at a time %s ms' % (d['value'], d['ms_per_step'], d.get('host_cpu_ms_per_step', -1), d.get('in_flight_1', {}).get('ms_per_step')))"
done
