#!/bin/bash
# which part of the delivery costs the pipelined step?  KM_DEBUG_DELIVER: 1 = no copy, 2 = no kernels
# (timing ablations; nothing valid arrives); and the number of hardware queues the streams share
export KM_LIBRARY=$(python3 tools/_diag.py)     # KM_DEBUG_DELIVER exists in the diagnostics build only
run() {
  env "$@" python3 bench.py --steps 40 --warmup 4 --no-cpu --only-step --check 0 --cache /tmp/kmc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*: %.3f ms/step delivered, %.3f kernel-only, %.3f full delivery' % (d['ms_per_step'], d['kernel_only']['ms_per_step'], d['full_delivery']['ms_per_step']))
"
}
if [ "$1" = "queues" ]; then
  for q in 4 8 16; do run GPU_MAX_HW_QUEUES=$q; run GPU_MAX_HW_QUEUES=$q; done
else
  for d in 0 1 2 3; do run KM_DEBUG_DELIVER=$d; done
fi
