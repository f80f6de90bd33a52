mkdir -p gpurun_out/r2a
VARIANTS="base:GEMM_EPI=0;epiT:GEMM_EPI=1;e4:GEMM_DBG=4;e4T:GEMM_DBG=4,GEMM_EPI=1" timeout -k 10 900 python tools/bench_gemm_variants.py > gpurun_out/r2a/gemm_variants6.log 2>&1
cat gpurun_out/r2a/gemm_variants6.log
