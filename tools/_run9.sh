mkdir -p gpurun_out/r2c
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "gemm" 2>&1 | tail -4
timeout -k 10 300 python tools/bench_skinny.py 2>&1 | tail -5
