"""(r04: the same eight-columns-per-lane form for add_ln_fwd measured SLOWER -- 90.9 vs 83.4 us: its two fp32 streams then move 32-byte
strided pieces -- and was not kept; neither was a form that keeps the fp32 streams' mapping and regroups the two 16-bit streams through a wave-private LDS
slice into 16-byte pieces: 84.4 vs 82.4 us.)  LayerNorm backward at the vision tower's size (50432 x 768), 16-bit cotangent in, residual-stream gradient in half / fp32: us per launch, TB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prcv2025reid_amd import ops, _lib
rows, cols = 256 * 197, 768
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn(rows, cols, device='cuda', generator=g); gam = torch.ones(cols, device='cuda')
mean = x.mean(1); rstd = 1.0 / x.std(1)
dy = torch.randn(rows, cols, device='cuda', generator=g).to(_lib.t16())
dxb = torch.empty(rows, cols, device='cuda', dtype=_lib.t16())
for dt, nb in ((torch.float16, 12), (torch.float32, 16)):
    dres = torch.randn(rows, cols, device='cuda', generator=g).to(dt); dx = torch.empty(rows, cols, device='cuda', dtype=dt)
    for _ in range(5): ops.layernorm_bwd(dy, x, gam, mean, rstd, dx, dx_bf16=dxb, dres=dres)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.layernorm_bwd(dy, x, gam, mean, rstd, dx, dx_bf16=dxb, dres=dres)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f'ln_bwd dx {str(dt):14s}: {us:6.1f} us  {rows * cols * nb / us / 1e6:5.2f} TB/s')

# the fused residual add + LayerNorm forward at the same size (x fp32 in/out, branch output 16-bit in, h 16-bit out: 12 B per element)
y = torch.randn(rows, cols, device='cuda', generator=g).to(torch.float16)
xo = torch.empty_like(x); hb = torch.empty(rows, cols, device='cuda', dtype=_lib.t16()); beta = torch.zeros(cols, device='cuda')
m2 = torch.empty(rows, device='cuda'); r2 = torch.empty(rows, device='cuda')
def fwd(): ops.add_layernorm_fwd(x, y, xo, gam, beta, hb, m2, r2)
for _ in range(5): fwd()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): fwd()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
print(f'add_ln_fwd                : {us:6.1f} us  {rows * cols * 12 / us / 1e6:5.2f} TB/s')
