set -e
mkdir -p gpurun_out/r2a
VARIANTS="base:GEMM_PERSIST=0;t2:GEMM_PERSIST=0,GEMM_TILE=2;t8:GEMM_PERSIST=0,GEMM_TILE=8;kbase:GEMM_PERSIST=0,GEMM_DBG=1;kt2:GEMM_PERSIST=0,GEMM_TILE=2,GEMM_DBG=1;kt8:GEMM_PERSIST=0,GEMM_TILE=8,GEMM_DBG=1" timeout -k 10 900 python tools/bench_gemm_variants.py > gpurun_out/r2a/gemm_variants2.log 2>&1
cat gpurun_out/r2a/gemm_variants2.log
