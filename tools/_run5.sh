mkdir -p gpurun_out/r2b
timeout -k 10 1100 python -m pytest tests/test_model_gpu.py tests/test_parallel_gpu.py -q -m gpu -s -k "edge or moddrop or modality_dropout or two_ranks or device_masks or learnable or constructor" > gpurun_out/r2b/t_model2.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2b/t_model2.log
grep -n "^E  .*Error\|^E  .*assert\|FAILED\|passed\|failed\|DP(2\|\] edge\|tiny_moddrop" gpurun_out/r2b/t_model2.log | cut -c1-220 | tail -50
