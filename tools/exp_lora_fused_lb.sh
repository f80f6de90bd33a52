#!/bin/bash
# fused adapter-gradient kernel variants built with tools/build_variant.sh <name> (REID_LIB_BF16 selects the library), interleaved on one box
run() {
  env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-retrieval --no-parity --no-second-flavor 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('  value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2), 'gemm frac', round(r['frac'], 4), 'gemm ms/step', round(r['kernel_ms_per_step'], 2))"
}
V=$PWD/prcv2025reid_amd/csrc/libreid_hip_${1:-lb3}.so
for spec in "X=0" "REID_LIB_BF16=$V" "X=0" "REID_LIB_BF16=$V"; do
  echo "$spec"; run $spec
done
