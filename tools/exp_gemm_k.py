import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from tools.bench_gemm import bench
M = 64*4*197
for K in (64, 128, 256, 768):
    bench(M, 2304, K, K2=0)
