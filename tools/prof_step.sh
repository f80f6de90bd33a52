#!/bin/bash
# rocprofv3 kernel trace of the benchmark step; writes gpurun_out/<tag>/*kernel_stats.csv and a per-step summary
TAG=${1:-prof}
shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG -o $TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 7 --warmup 0 --no-cpu-baseline --no-retrieval --no-parity --no-second-flavor --no-kernel-events "$@" > $GRAFT_REPO_ROOT/gpurun_out/$TAG.json 2> $GRAFT_REPO_ROOT/gpurun_out/$TAG.log
cd $GRAFT_REPO_ROOT
python3 tools/step_summary.py $(find gpurun_out/$TAG -name "*kernel_stats.csv" | head -1) 7 | tee gpurun_out/$TAG.summary.txt
