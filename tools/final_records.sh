#!/bin/bash
# round-end records on one GPU box: kernel stats of the bench step, PMC traffic, the default bench line, the per-rank workloads of configs 3 and 5
set -o pipefail
TAG=${1:-r03}
timeout -k 10 400 bash tools/prof_step.sh ${TAG}_final > gpurun_out/${TAG}_final.out 2>&1 || exit 1
timeout -k 10 400 bash tools/pmc_collect.sh ${TAG} > gpurun_out/${TAG}_pmc.out 2>&1 || exit 2
timeout -k 10 600 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.log || exit 3
timeout -k 10 300 python bench.py --P 32 --K 4 --steps 6 --warmup 2 --no-cpu-baseline --no-retrieval --no-parity --no-second-flavor > gpurun_out/${TAG}_bench_P32K4.json 2> gpurun_out/${TAG}_bench_P32K4.log || exit 4
timeout -k 10 300 python bench.py --P 64 --K 4 --rank 16 --mask-drop 0.3 --accum 2 --steps 4 --warmup 2 --no-cpu-baseline --no-retrieval --no-parity --no-second-flavor > gpurun_out/${TAG}_bench_P64K4_r16_accum2.json 2> gpurun_out/${TAG}_bench_P64K4_r16_accum2.log || exit 5
echo all done
