"""Stand-alone times of the reduce-over-rows GEMMs (LoRA dA / dB) at training size, against the bytes they must read."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prcv2025reid_amd import ops, _lib
T16 = _lib.t16()
M = 64 * 4 * 197
g = torch.Generator(device='cuda').manual_seed(0)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
tot = 0.0; totb = 0.0
for name, P, Q, count in (('dB fc1  (3072 x 32)', 3072, 32, 1), ('dB fc2/out/qkv-group (768 x 32)', 768, 32, 5), ('dA fc2 (32 x 3072)', 32, 3072, 1),
                          ('dA fc1/out/qkv-group (32 x 768)', 32, 768, 5), ('dB qkv as one (2304 x 96)', 2304, 96, 0), ('dA qkv as one (96 x 768)', 96, 768, 0)):
    X = torch.randn(M, P, device='cuda', generator=g).to(T16); Y = torch.randn(M, Q, device='cuda', generator=g).to(T16)
    C = torch.zeros(P, Q, device='cuda')
    us = min(timeit(lambda: ops.gemm_tn(X, Y, C, beta=1.0)) for _ in range(3))
    nbytes = M * (P + Q) * 2
    print(f'{name:34s}: {us:7.1f} us  {nbytes / 1e6:6.1f} MB  {nbytes / us / 1e6:5.2f} TB/s   x{count} per layer', flush=True)
    tot += us * count; totb += nbytes * count
print(f'per layer: {tot:.0f} us for {totb / 1e6:.0f} MB = {totb / tot / 1e6:.2f} TB/s; 12 layers: {tot * 12 / 1e3:.2f} ms')
