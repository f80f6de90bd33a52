mkdir -p gpurun_out/r2c
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "sdm" 2>&1 | tail -2
timeout -k 10 300 python tools/bench_sdm.py > gpurun_out/r2c/bench_sdm2.log 2>&1; python - <<'PY'
import json
for l in open('gpurun_out/r2c/bench_sdm2.log'):
    if l.startswith('{'):
        d=json.loads(l); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in('P','N','fwd_us','bwd_us','fwd_tflops_fp32','bwd_tflops_fp32','ws_MB')})
PY
