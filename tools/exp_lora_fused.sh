#!/bin/bash
# adapter gradients: U and dB from one pass over dY (default) vs two launches (REID_LORA_FUSED=0); interleaved on one box
run() {
  env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-retrieval --no-parity --no-second-flavor 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('  value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2), 'gemm frac', round(r['frac'], 4), 'gemm ms/step', round(r['kernel_ms_per_step'], 2))"
}
for spec in "REID_LORA_FUSED=1" "REID_LORA_FUSED=768" "REID_LORA_FUSED=0" "REID_LORA_FUSED=1" "REID_LORA_FUSED=768" "REID_LORA_FUSED=0"; do
  echo "$spec"; run $spec
done
