#!/bin/bash
# fused adapter-gradient kernel: register budget (2 waves/SIMD = 194 VGPRs vs 3 = 168) and the two-launch path, interleaved
run() {
  env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-retrieval --no-parity --no-second-flavor 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('  value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2), 'gemm frac', round(r['frac'], 4), 'gemm ms/step', round(r['kernel_ms_per_step'], 2))"
}
V=$PWD/prcv2025reid_amd/csrc/libreid_hip_lb3.so
for spec in "X=0" "REID_LIB_BF16=$V" "REID_LORA_FUSED=0" "X=0" "REID_LIB_BF16=$V" "REID_LORA_FUSED=0"; do
  echo "$spec"; run $spec
done
