#!/bin/bash
# rocprofv3 kernel stats of the mid-range retrieval call (tools/bench_topk_mid.py NQ): tools/prof_topk_mid.sh TAG NQ [ENV=VAL]
TAG=$1; NQ=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG -o $TAG -- python3 $GRAFT_REPO_ROOT/tools/bench_topk_mid.py $NQ > $GRAFT_REPO_ROOT/gpurun_out/$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/$TAG/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print('%-90s calls %5s avg %9.1f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
