mkdir -p gpurun_out/r2c
timeout -k 10 1150 python -m pytest tests -q -m gpu > gpurun_out/r2c/t_all.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/r2c/t_all.log | cut -c1-300
