mkdir -p gpurun_out/r2f
timeout -k 10 180 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/r2f/t_epi.log 2>&1; rc=$?; echo "test rc=$rc"; tail -3 gpurun_out/r2f/t_epi.log | cut -c1-200
if [ $rc -ne 0 ]; then grep -E "^E  " gpurun_out/r2f/t_epi.log | head -10 | cut -c1-250; exit 0; fi
REID_GEMM_TILE=12 timeout -k 10 180 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -2
VARIANTS="generic:GEMM_EPI=0;lean:GEMM_EPI=-1;pp:GEMM_TILE=12;ppgen:GEMM_TILE=12,GEMM_EPI=0;e4gen:GEMM_DBG=4,GEMM_EPI=0;e4lean:GEMM_DBG=4;e4pp:GEMM_DBG=4,GEMM_TILE=12" timeout -k 10 300 python tools/bench_gemm_variants.py > gpurun_out/r2f/gemm_epi.log 2>&1
cat gpurun_out/r2f/gemm_epi.log
