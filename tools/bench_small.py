"""GPU micro-benchmarks of the non-GEMM kernels on the training-step shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prcv2025reid_amd import ops, _lib

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

n_img, S, d, ff, Rp, heads = 256, 197, 768, 3072, 32, 12
M = n_img * S
T16 = _lib.t16()
g = torch.Generator(device='cuda').manual_seed(0)
h = torch.randn(M, d, device='cuda', generator=g).to(T16)
gg = torch.randn(M, ff, device='cuda', generator=g).to(T16)
A = (torch.randn(Rp, d, device='cuda', generator=g) * 0.05).to(T16)
A3 = (torch.randn(3 * Rp, d, device='cuda', generator=g) * 0.05).to(T16)
Aff = (torch.randn(Rp, ff, device='cuda', generator=g) * 0.05).to(T16)
img_mod = torch.randint(0, 4, (n_img,), device='cuda', dtype=torch.int32)
T = torch.empty(M, Rp, device='cuda', dtype=T16); T3 = torch.empty(M, 3 * Rp, device='cuda', dtype=T16)
mk = dict(img_mod=img_mod, mask_r=8, mask_period=Rp, rows_per_img=S, alpha=0.125)
us = timeit(lambda: ops.gemm(h, A, T, **mk)); print(f'lora-down K=768 N=32 : {us:7.1f} us  {M*d*2/us/1e6:6.2f} TB/s')
us = timeit(lambda: ops.gemm(h, A3, T3, **mk)); print(f'lora-down K=768 N=96 : {us:7.1f} us  {M*d*2/us/1e6:6.2f} TB/s')
us = timeit(lambda: ops.gemm(gg, Aff, T, **mk)); print(f'lora-down K=3072 N=32: {us:7.1f} us  {M*ff*2/us/1e6:6.2f} TB/s')
gB = torch.empty(d, Rp, device='cuda'); gA = torch.empty(Rp, d, device='cuda'); gBf = torch.empty(ff, Rp, device='cuda'); gAf = torch.empty(Rp, ff, device='cuda')
us = timeit(lambda: ops.gemm_tn(h, T, gB)); print(f'tn dB [768,32]       : {us:7.1f} us  {M*(d+Rp)*2/us/1e6:6.2f} TB/s')
us = timeit(lambda: ops.gemm_tn(T, h, gA)); print(f'tn dA [32,768]       : {us:7.1f} us  {M*(d+Rp)*2/us/1e6:6.2f} TB/s')
us = timeit(lambda: ops.gemm_tn(gg, T, gBf)); print(f'tn dB [3072,32]      : {us:7.1f} us  {M*(ff+Rp)*2/us/1e6:6.2f} TB/s')
us = timeit(lambda: ops.gemm_tn(T, gg, gAf)); print(f'tn dA [32,3072]      : {us:7.1f} us  {M*(ff+Rp)*2/us/1e6:6.2f} TB/s')
qkv = torch.randn(M, 3 * d, device='cuda', generator=g).to(T16)
o = torch.empty(M, d, device='cuda', dtype=T16); lse = torch.empty(n_img, heads, S, device='cuda')
us = timeit(lambda: ops.attn_fwd(qkv, o, lse, n_img, S, heads)); fl = 4.0 * S * S * 64 * n_img * heads
print(f'attn fwd             : {us:7.1f} us  {fl/us/1e6:6.1f} TFLOP/s (useful)')
do = torch.randn(M, d, device='cuda', generator=g).to(T16); dqkv = torch.empty(M, 3 * d, device='cuda', dtype=T16); delta = torch.empty_like(lse)
us = timeit(lambda: ops.attn_bwd(qkv, o, do, lse, dqkv, delta, n_img, S, heads)); print(f'attn bwd (3 kernels) : {us:7.1f} us  {2.5*fl/us/1e6:6.1f} TFLOP/s (useful)')
x = torch.randn(M, d, device='cuda', generator=g); gam = torch.ones(d, device='cuda'); bet = torch.zeros(d, device='cuda')
mean = torch.empty(M, device='cuda'); rstd = torch.empty(M, device='cuda')
us = timeit(lambda: ops.layernorm_fwd(x, gam, bet, y_bf16=o, mean=mean, rstd=rstd)); print(f'ln fwd               : {us:7.1f} us  {M*d*6/us/1e6:6.2f} TB/s')
dx = torch.empty_like(x); dxb = torch.empty_like(o)
us = timeit(lambda: ops.layernorm_bwd(do, x, gam, mean, rstd, dx, dx_bf16=dxb, dres=x)); print(f'ln bwd               : {us:7.1f} us  {M*d*16/us/1e6:6.2f} TB/s')
# head-sized fp32 GEMMs (latency bound)
for (Ms, Ns, Ks, kw, name) in [(320, 1536, 512, dict(tb=True), 'in_proj  x@W.T'), (320, 512, 512, dict(tb=True), 'out_proj x@W.T'),
                               (320, 512, 1536, dict(), 'dx = dy@W'), (1536, 512, 320, dict(ta=True), 'dW = dy.T@x'),
                               (64, 400, 512, dict(tb=True), 'classifier'), (64, 2048, 512, dict(tb=True), 'mlp up')]:
    if kw.get('tb'):
        a_ = torch.randn(Ms, Ks, device='cuda'); b_ = torch.randn(Ns, Ks, device='cuda')
    elif kw.get('ta'):
        a_ = torch.randn(Ks, Ms, device='cuda'); b_ = torch.randn(Ks, Ns, device='cuda')
    else:
        a_ = torch.randn(Ms, Ks, device='cuda'); b_ = torch.randn(Ks, Ns, device='cuda')
    c_ = torch.empty(Ms, Ns, device='cuda')
    us = timeit(lambda: ops.sgemm(a_, b_, c_, **kw)); print(f'sgemm {name:16s} [{Ms}x{Ns}x{Ks}]: {us:7.1f} us')
