#!/bin/bash
# upper bound of what one pass over dY per LoRA linear less would buy: the step with the U = dY.B launches left out (wrong gradients; timing only)
run() {
  env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-retrieval --no-parity --no-second-flavor 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('  value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2), 'gemm frac', round(r['frac'], 4), 'gemm ms/step', round(r['kernel_ms_per_step'], 2))"
}
for spec in "REID_EXP_SKIP_U=0" "REID_EXP_SKIP_U=1" "REID_EXP_SKIP_U=0" "REID_EXP_SKIP_U=1"; do
  echo "$spec"; run $spec
done
