mkdir -p gpurun_out/r2d
for i in 1 2; do
for flag in "" "--hi-prio"; do
timeout -k 10 300 python bench.py --steps 15 --warmup 4 $flag --no-retrieval --no-cpu-baseline --no-parity --no-second-flavor --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$flag', d['value'], d['ms_per_step'])"
done; done
