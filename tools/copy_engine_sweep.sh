#!/bin/bash
# delivered step time under different copy-engine settings of the HIP / HSA runtimes (diagnostics)
run() {
  echo "== $*"
  env "$@" python3 bench.py --steps 40 --warmup 4 --no-cpu --only-step --check 0 --cache /tmp/kmc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   %.3f ms/step delivered, %.3f kernel-only, %.3f full delivery; d2h alone %.3f ms lean / %.3f full' % (d['ms_per_step'], d['kernel_only']['ms_per_step'], d['full_delivery']['ms_per_step'], d['kernel_ms']['d2h_copy'], d['full_delivery']['d2h_copy_ms']))
"
}
for i in 1 2 3; do
run X=1
run HSA_ENABLE_SDMA=1
run HSA_ENABLE_SDMA_GANG=0
done
