#!/bin/bash
# A/B of the side stream's priority (engine._side_stream) on the benchmark step: prints ms/step per setting
python -c "import torch; print('priority_range', torch.cuda.Stream.priority_range())"
for p in 0 1 2 -1; do
  echo "REID_SIDE_PRIORITY=$p"
  REID_SIDE_PRIORITY=$p python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-retrieval --no-parity --no-second-flavor 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('  value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2), 'gemm frac', round(r['frac'], 4), 'gemm ms/step', round(r['kernel_ms_per_step'], 2))"
done
