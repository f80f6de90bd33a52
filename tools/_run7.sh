mkdir -p gpurun_out/r2c
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "sdm" > gpurun_out/r2c/t_sdm.log 2>&1; echo "rc=$?"; grep -E "^E  |passed|failed|FAILED" gpurun_out/r2c/t_sdm.log | cut -c1-240 | head -30
timeout -k 10 300 python tools/bench_sdm.py > gpurun_out/r2c/bench_sdm.log 2>&1; cat gpurun_out/r2c/bench_sdm.log | cut -c1-700
timeout -k 10 600 python -m pytest tests/test_evaluate_gpu.py -q -m gpu -s -k "config4" > gpurun_out/r2c/t_cfg4.log 2>&1; echo "rc=$?"; grep -E "^E  |passed|failed|FAILED|10k x" gpurun_out/r2c/t_cfg4.log | cut -c1-240 | head
