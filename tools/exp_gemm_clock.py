"""Shader clock held inside the K loop of the persistent ping-pong GEMM (library built with -DREID_GEMM_TRACE -DREID_GEMM_TRACE_CLOCK):
delta s_memtime / delta s_memrealtime x 100 MHz over the K loop of every tile, after ~2 s of back-to-back launches on random data."""
import os, sys, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from prcv2025reid_amd import ops, _lib
T16 = _lib.t16()
M, d, ff = int(os.environ.get("M_ROWS", 64 * 4 * 197)), 768, 3072
g = torch.Generator(device='cuda').manual_seed(0)
h = torch.randn(M, d, device='cuda', generator=g).to(T16)
Wq = (torch.randn(3 * d, d, device='cuda', generator=g) * 0.03).to(T16); bq = torch.randn(3 * d, device='cuda', generator=g)
qkv = torch.empty(M, 3 * d, device='cuda', dtype=T16)
gact = torch.randn(M, ff, device='cuda', generator=g).to(T16); W2 = (torch.randn(d, ff, device='cuda', generator=g) * 0.03).to(T16)
dh = torch.empty(M, d, device='cuda', dtype=T16)
lib = _lib.lib()
cases = [('qkv plain16 (N=2304, K=768)', lambda: ops.gemm(h, Wq, qkv, bias=bq), ((M + 255) // 256) * 9, 10),
         ('fc1b plain16 (N=768, K=3072)', lambda: ops.gemm(gact, W2, dh), ((M + 223) // 224) * 3, 46)]
for name, fn, nt, nkt in cases:
    trace = torch.zeros(nt + 64, 8, dtype=torch.int64, device='cuda')
    t0 = time.time()
    while time.time() - t0 < 2.0:
        for _ in range(50): fn()
        torch.cuda.synchronize()
    lib.reid_debug_gemm_trace(ctypes.c_void_p(trace.data_ptr()))
    fn(); torch.cuda.synchronize()
    lib.reid_debug_gemm_trace(ctypes.c_void_p(0))
    t = trace.cpu().numpy().astype(np.int64)
    t = t[t[:, 0] > 0]
    dreal = (t[:, 2] - t[:, 1]) / 100.0            # us
    dclk = (t[:, 7] - t[:, 6]).astype(np.float64)  # shader cycles
    ghz = dclk / dreal / 1e3
    print(f'== {name}: {len(t)} tiles; K loop (first {nkt} K-tiles) {dreal.mean():.2f} us = {dclk.mean():.0f} cycles; clock median {np.median(ghz):.3f} GHz (p10 {np.percentile(ghz, 10):.3f}, p90 {np.percentile(ghz, 90):.3f}); '
          f'cycles per K-tile {dclk.mean() / nkt:.0f} (MFMA-bound: 2048)')
