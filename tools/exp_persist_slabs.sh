#!/bin/bash
# persistent multiply-by-derivative GEMM (REID_GEMM_PERSIST=13) with more row slabs for the side stream's fused kernels
run() {
  env $1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-retrieval --no-parity --no-second-flavor 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('  value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2), 'gemm frac', round(r['frac'], 4), 'gemm ms/step', round(r['kernel_ms_per_step'], 2))"
}
for spec in "X=0" "REID_GEMM_PERSIST=13 REID_TN_BLOCKS=768" "REID_GEMM_PERSIST=13 REID_TN_BLOCKS=1536" "REID_GEMM_PERSIST=13 REID_TN_BLOCKS=576" "X=0"; do
  echo "$spec"; run "$spec"
done
