#!/bin/bash
# r04 in-step A/B on ONE box, alternating rounds: tools/exp_r04_ab.sh "NAME=ENV=VAL,ENV=VAL" "NAME2=..." ...   (NAME=none: defaults)
QUICK="--steps 20 --warmup 6 --no-cpu-baseline --no-retrieval --no-kernel-events --no-second-flavor --no-parity"
for round in 1 2; do
  for spec in "$@"; do
    name="${spec%%=*}"; envs="${spec#*=}"
    if [ "$envs" = "none" ]; then envs=""; fi
    line=$(env $(echo "$envs" | tr ',' ' ') python bench.py $QUICK 2>/dev/null | tail -1)
    echo "$name round $round: $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f inst/s  %.2f ms/step" % (d["value"], d["ms_per_step"]))')"
  done
done
