mkdir -p gpurun_out/r2b
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2b/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r2b/smoke.log; grep -v Warning gpurun_out/r2b/smoke.log | tail -4
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2b/bench1.json 2> gpurun_out/r2b/bench1.err; echo "bench rc=$?"; tail -3 gpurun_out/r2b/bench1.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2b/bench1.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','dtype','n_gpus')}); print('roofline',{k:d['roofline'][k] for k in ('achieved','frac','avg_launch_us','kernel_ms_per_step')})
print('flavors',d['flavors']); print('parity',{k:(v if not isinstance(v,dict) else {kk:v[kk] for kk in ('bn_features_unit_maxabs','worst','meets_1e-3','loss_abs')}) for k,v in d['parity'].items() if k in('bf16','f16','oracle_seconds')})
print('cpu',d['cpu_baseline']); print('retr',{k:d['retrieval'][k] for k in ('queries_per_s','ms','mfma_frac')}, d['retrieval']['single_query'])
PY
timeout -k 10 600 python bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --P 8 --no-kernel-events > gpurun_out/r2b/bench_gloo2.json 2> gpurun_out/r2b/bench_gloo2.err; echo "gloo2 rc=$?"; tail -2 gpurun_out/r2b/bench_gloo2.err; cat gpurun_out/r2b/bench_gloo2.json | cut -c1-1500
