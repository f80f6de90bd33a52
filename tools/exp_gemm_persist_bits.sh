#!/bin/bash
# which epilogues run on the persistent ping-pong GEMM (REID_GEMM_PERSIST bits: 1 plain/residual, 2 GELU, 4 multiply-by-derivative, 8 out-projection shapes; default 9)
run() {
  env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-retrieval --no-parity --no-second-flavor 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('  value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2), 'gemm frac', round(r['frac'], 4), 'gemm ms/step', round(r['kernel_ms_per_step'], 2))"
}
for spec in ${SPECS:-"X=0" "REID_GEMM_PERSIST=13" "REID_GEMM_PERSIST=11" "REID_GEMM_PERSIST=15" "X=0" "REID_GEMM_PERSIST=1"}; do
  echo "$spec"; run $spec
done
