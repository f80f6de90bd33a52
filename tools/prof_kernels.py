"""Small fixed workload for PMC collection: QKV-shaped GEMM, attention fwd/bwd, LN, TN, retrieval filter pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prcv2025reid_amd import ops, _lib
n_img, S, d, heads = 256, 197, 768, 12
M = n_img * S
T16 = _lib.t16()
g = torch.Generator(device='cuda').manual_seed(0)
h = torch.randn(M, d, device='cuda', generator=g).to(T16)
W = (torch.randn(3 * d, d, device='cuda', generator=g) * 0.03).to(T16)
bias = torch.randn(3 * d, device='cuda', generator=g)
qkv = torch.empty(M, 3 * d, device='cuda', dtype=T16)
o = torch.empty(M, d, device='cuda', dtype=T16); lse = torch.empty(n_img, heads, S, device='cuda')
do = torch.randn(M, d, device='cuda', generator=g).to(T16); dqkv = torch.empty(M, 3 * d, device='cuda', dtype=T16); delta = torch.empty_like(lse)
for _ in range(3):
    ops.gemm(h, W, qkv, bias=bias)
    ops.attn_fwd(qkv, o, lse, n_img, S, heads)
    ops.attn_bwd(qkv, o, do, lse, dqkv, delta, n_img, S, heads)
torch.cuda.synchronize()
print('done')
