"""GPU micro-benchmark of reid_mer_gemm on the shapes of the training step (random data)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prcv2025reid_amd import ops

def bench(M, N, K, K2=32, reps=20, **kw):
    g = torch.Generator(device='cuda').manual_seed(0)
    A = torch.randn(M, K, device='cuda', generator=g).to(torch.bfloat16)
    B = (torch.randn(N, K, device='cuda', generator=g) * 0.05).to(torch.bfloat16)
    A2 = torch.randn(M, K2, device='cuda', generator=g).to(torch.bfloat16) if K2 else None
    B2 = torch.randn(N, K2, device='cuda', generator=g).to(torch.bfloat16) if K2 else None
    bias = torch.randn(N, device='cuda', generator=g)
    C = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(A, B, C, A2=A2, B2=B2, K2=K2, bias=bias, **kw)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm(A, B, C, A2=A2, B2=B2, K2=K2, bias=bias, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tf = 2.0 * M * N * (K + K2) / ms / 1e9
    print(f'M={M:6d} N={N:5d} K={K:5d}+{K2:3d}: {ms*1e3:8.1f} us  {tf:7.1f} TFLOP/s  ({tf/25:.1f}% of 2.5 PF)', flush=True)
    return tf

if __name__ == '__main__':
    M = 64 * 4 * 197
    for (N, K) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
        bench(M, N, K)
    bench(8192, 8192, 8192, K2=0, reps=5)
    bench(4096, 4096, 4096, K2=0, reps=10)
