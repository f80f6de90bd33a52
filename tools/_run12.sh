set -e
mkdir -p gpurun_out/r2e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-retrieval --no-kernel-events --no-parity --no-second-flavor"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2e/prof -- python3 $CMD > gpurun_out/r2e/prof.log 2>&1
cp $(find gpurun_out/r2e/prof -name "*kernel_stats.csv" | head -1) gpurun_out/r2e/kernel_stats.csv
rm -rf gpurun_out/r2e/prof
CMD1="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-retrieval --no-kernel-events --no-parity --no-second-flavor"
timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2e/pmc_f -- python3 $CMD1 > gpurun_out/r2e/pmc_f.log 2>&1
timeout -k 10 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r2e/pmc_w -- python3 $CMD1 > gpurun_out/r2e/pmc_w.log 2>&1
python tools/pmc_summary.py gpurun_out/r2e/pmc_f gpurun_out/r2e/pmc_w r02 "$CMD1"
cp profiles/r02_pmc_traffic.* gpurun_out/r2e/
rm -rf gpurun_out/r2e/pmc_f gpurun_out/r2e/pmc_w
head -12 gpurun_out/r2e/r02_pmc_traffic.md | cut -c1-200
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2e/prof_sdm -- python3 tools/bench_sdm.py > gpurun_out/r2e/sdm.log 2>&1
cp $(find gpurun_out/r2e/prof_sdm -name "*kernel_stats.csv" | head -1) gpurun_out/r2e/sdm_kernel_stats.csv; rm -rf gpurun_out/r2e/prof_sdm
head -8 gpurun_out/r2e/sdm_kernel_stats.csv | cut -c1-160
