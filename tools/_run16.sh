mkdir -p gpurun_out/r2f
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/r2f/t_all.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r2f/t_all.log | cut -c1-300
timeout -k 10 300 python bench.py --steps 15 --warmup 4 --no-retrieval --no-cpu-baseline --no-parity --no-second-flavor 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline'])"
