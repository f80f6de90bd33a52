"""Is the GEMM epilogue bound per CU or by the chip?  Epilogue-only launches (REID_GEMM_DBG=4: no K loop) of the 256x256 ping-pong kernel
on fc1-shaped (GELU, two 16-bit outputs), dGELU-shaped and residual-shaped problems with FEWER tiles than CUs up to the training size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prcv2025reid_amd import ops, _lib
T16 = _lib.t16()
def knob(n, v): _lib.check(_lib.lib().reid_set_knob(n.encode(), v))
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = torch.Generator(device='cuda').manual_seed(0)
d, ff = 768, 3072
knob('GEMM_TILE', 12)
for tiles_m in (2, 5, 10, 21, 42, 84, 197):
    M = tiles_m * 256
    h = torch.randn(M, d, device='cuda', generator=g).to(T16); W1 = (torch.randn(ff, d, device='cuda', generator=g) * 0.03).to(T16)
    b1 = torch.randn(ff, device='cuda', generator=g)
    u = torch.empty(M, ff, device='cuda', dtype=T16); g2 = torch.empty(M, ff, device='cuda', dtype=T16)
    x = torch.randn(M, d, device='cuda', generator=g); xo = torch.empty(M, d, device='cuda')
    gact = torch.randn(M, ff, device='cuda', generator=g).to(T16); W2 = (torch.randn(d, ff, device='cuda', generator=g) * 0.03).to(T16)
    cases = [('fc1 gelu2 ', lambda: ops.gemm(h, W1, g2, bias=b1, act='gelu', C2=u), M * ff * 4, tiles_m * 12),
             ('fc2b dgelu', lambda: ops.gemm(h, W1, g2, act='dgelu', aux=u), M * ff * 4, tiles_m * 12),
             ('fc2 res32 ', lambda: ops.gemm(gact, W2, xo, R=x), M * d * 8, tiles_m * 3)]
    for name, fn, nbytes, tiles in cases:
        res = {}
        for dbg, lab in ((4, 'epilogue only'), (1, 'k loop only'), (-1, 'full')):
            knob('GEMM_DBG', dbg)
            res[lab] = min(timeit(fn) for _ in range(2))
        knob('GEMM_DBG', -1)
        rounds = -(-tiles // 256)
        print(f'{name} tiles {tiles:5d} ({tiles / 256:5.2f} rounds): epilogue only {res["epilogue only"]:7.1f} us = {res["epilogue only"] / rounds:6.1f} us/round, '
              f'{nbytes / res["epilogue only"] / 1e6:6.2f} TB/s | k loop {res["k loop only"]:7.1f} | full {res["full"]:7.1f} | full - k loop {res["full"] - res["k loop only"]:6.1f}', flush=True)
knob('GEMM_TILE', -1)
