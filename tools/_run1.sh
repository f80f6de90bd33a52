set -e
mkdir -p gpurun_out/r2a
hipcc --offload-arch=gfx950 -O2 -Wno-unused-value tools/hwinfo.hip -o /tmp/hwinfo 2>/dev/null && /tmp/hwinfo > gpurun_out/r2a/hwinfo.log 2>&1
REID_GEMM_PERSIST=1 timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/r2a/t_gemm_persist.log 2>&1
VARIANTS="base:GEMM_PERSIST=0;pers0:GEMM_PERSIST=1,GEMM_STAGGER=0;s50:GEMM_PERSIST=1,GEMM_STAGGER=50;s100:GEMM_PERSIST=1,GEMM_STAGGER=100;s150:GEMM_PERSIST=1,GEMM_STAGGER=150;s250:GEMM_PERSIST=1,GEMM_STAGGER=250;m2:GEMM_PERSIST=1,GEMM_STAGGER=100,GEMM_STAGGER_MODE=2;m3:GEMM_PERSIST=1,GEMM_STAGGER=100,GEMM_STAGGER_MODE=3;kbase:GEMM_PERSIST=0,GEMM_DBG=1;kpers:GEMM_PERSIST=1,GEMM_STAGGER=0,GEMM_DBG=1" timeout -k 10 900 python tools/bench_gemm_variants.py > gpurun_out/r2a/gemm_variants.log 2>&1
tail -5 gpurun_out/r2a/t_gemm_persist.log
cat gpurun_out/r2a/gemm_variants.log
