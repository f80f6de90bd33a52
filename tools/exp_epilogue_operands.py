"""Which epilogue operand costs the out-projection GEMM its time?  Same launch with the fp32 residual (a) streamed from HBM,
(b) periodic over 128 rows (always L2-resident), (c) absent; and with a 16-bit instead of an fp32 output."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prcv2025reid_amd import ops, _lib
T16 = _lib.t16()
M, d = 64 * 4 * 197, 768
g = torch.Generator(device='cuda').manual_seed(0)
rnd = lambda *s, sc=1.0, dt=None: (torch.randn(*s, device='cuda', generator=g) * sc).to(dt or T16)
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
h = rnd(M, d); W = rnd(d, d, sc=0.03); b = rnd(d, dt=torch.float32)
x = rnd(M, d, dt=torch.float32); xo = torch.empty(M, d, device='cuda'); xb = torch.empty(M, d, device='cuda', dtype=T16)
cases = [('f32 out + f32 R from HBM', lambda: ops.gemm(h, W, xo, bias=b, R=x)),
         ('f32 out + f32 R periodic 128 rows (L2)', lambda: ops.gemm(h, W, xo, bias=b, R=x, r_period=128)),
         ('f32 out, no R', lambda: ops.gemm(h, W, xo, bias=b)),
         ('16-bit out, no R', lambda: ops.gemm(h, W, xb, bias=b)),
         ('16-bit out + f32 R from HBM', lambda: ops.gemm(h, W, xb, bias=b, R=x))]
for rnd_ in range(2):
    for name, fn in cases:
        print(f'{name:45s} {timeit(fn):7.1f} us', flush=True)
_lib.check(_lib.lib().reid_set_knob(b'GEMM_DBG', 1))
print(f'{"K loop only":45s} {timeit(cases[0][1]):7.1f} us')
