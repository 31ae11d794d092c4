#!/bin/bash
# pipelined step time against the number of batches in flight (diagnostics)
for n in ${@:-1 2 3 4 6 8}; do
  python3 bench.py --steps 40 --warmup 4 --no-cpu --only-step --check 0 --inflight $n --cache /tmp/kmc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('inflight %d: %.3f ms/step delivered, %.3f kernel-only, %.3f full delivery' % (d['batches_in_flight'], d['ms_per_step'], d['kernel_only']['ms_per_step'], d['full_delivery']['ms_per_step']))
"
done
