#!/bin/bash
# HBM traffic per kernel of one benchmark step: two SEPARATE rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> profiles/<tag>_pmc_traffic.{md,json}
# usage (GPU box): bash tools/pmc_collect.sh r03
TAG=${1:-r03}
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-retrieval --no-kernel-events --no-parity --no-second-flavor"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_fetch -o f -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_write -o w -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write $TAG "bench.py $ARGS"
mkdir -p gpurun_out/profiles_out && cp profiles/${TAG}_pmc_traffic.* gpurun_out/profiles_out/
