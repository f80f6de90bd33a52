mkdir -p gpurun_out/r2d
timeout -k 10 600 python bench.py --steps 10 --warmup 3 --graph on --no-retrieval --no-cpu-baseline --no-parity --no-second-flavor --no-kernel-events > gpurun_out/r2d/bench_graph.json 2> gpurun_out/r2d/bench_graph.err; echo "graph rc=$?"; tail -3 gpurun_out/r2d/bench_graph.err; cut -c1-400 gpurun_out/r2d/bench_graph.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2d/prof -- python3 bench.py --steps 5 --warmup 2 --no-retrieval --no-cpu-baseline --no-parity --no-second-flavor --no-kernel-events > gpurun_out/r2d/prof.log 2>&1; echo "prof rc=$?"
f=$(find gpurun_out/r2d/prof -name "*kernel_stats.csv" | head -1); echo $f; head -30 $f | cut -c1-180
