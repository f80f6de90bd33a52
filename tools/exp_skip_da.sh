#!/bin/bash
# how much of the step hangs on the side stream's length: the step with the dA = U^T x launches left out (wrong gradients; timing only)
run() {
  env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-retrieval --no-parity --no-second-flavor 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('  value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2), 'gemm frac', round(r['frac'], 4), 'gemm ms/step', round(r['kernel_ms_per_step'], 2))"
}
for spec in "X=0" "REID_EXP_SKIP_DA=1" "X=0" "REID_EXP_SKIP_DA=1" "REID_TN_STREAM=0"; do
  echo "$spec"; run $spec
done
