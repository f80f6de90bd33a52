set -e
mkdir -p gpurun_out/r2a
REID_GEMM_TILE=9 REID_GEMM_PERSIST=0 timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/r2a/t_gemm_ring9.log 2>&1 || (tail -30 gpurun_out/r2a/t_gemm_ring9.log; exit 1)
REID_GEMM_TILE=10 REID_GEMM_PERSIST=0 timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/r2a/t_gemm_ring10.log 2>&1 || (tail -30 gpurun_out/r2a/t_gemm_ring10.log; exit 1)
VARIANTS="base:GEMM_PERSIST=0;r9:GEMM_PERSIST=0,GEMM_TILE=9;r10:GEMM_PERSIST=0,GEMM_TILE=10;r11:GEMM_PERSIST=0,GEMM_TILE=11;kbase:GEMM_PERSIST=0,GEMM_DBG=1;kr9:GEMM_PERSIST=0,GEMM_TILE=9,GEMM_DBG=1;kr10:GEMM_PERSIST=0,GEMM_TILE=10,GEMM_DBG=1;kr11:GEMM_PERSIST=0,GEMM_TILE=11,GEMM_DBG=1" timeout -k 10 900 python tools/bench_gemm_variants.py > gpurun_out/r2a/gemm_variants3.log 2>&1
cat gpurun_out/r2a/gemm_variants3.log
