#!/bin/bash
# A/B an environment knob inside one box: tools/exp_ab.sh VAR  (values 0 and 1, alternating)
for r in 0 1 0 1; do
  echo -n "$1=$r "
  env "$1=$r" python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-retrieval --no-kernel-events 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'
done
