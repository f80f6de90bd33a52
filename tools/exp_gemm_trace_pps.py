"""Per-tile timeline of the PERSISTENT ping-pong GEMM (library built with -DREID_GEMM_TRACE): s_memrealtime (100 MHz) at
tile start / accumulators initialised / K loop minus two K-tiles / offsets of the next tile set / K loop done / epilogue issued."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from prcv2025reid_amd import ops, _lib
T16 = _lib.t16()
M, d, ff = int(os.environ.get("M_ROWS", 64 * 4 * 197)), 768, 3072
g = torch.Generator(device='cuda').manual_seed(0)
h = torch.randn(M, d, device='cuda', generator=g).to(T16); W1 = (torch.randn(ff, d, device='cuda', generator=g) * 0.03).to(T16)
Wq = (torch.randn(3 * d, d, device='cuda', generator=g) * 0.03).to(T16); bq = torch.randn(3 * d, device='cuda', generator=g)
b1 = torch.randn(ff, device='cuda', generator=g)
u = torch.empty(M, ff, device='cuda', dtype=T16); g2 = torch.empty(M, ff, device='cuda', dtype=T16)
qkv = torch.empty(M, 3 * d, device='cuda', dtype=T16)
gact = torch.randn(M, ff, device='cuda', generator=g).to(T16); W2 = (torch.randn(d, ff, device='cuda', generator=g) * 0.03).to(T16)
dh = torch.empty(M, d, device='cuda', dtype=T16)
lib = _lib.lib()
cases = [('qkv plain16 (N=2304, K=768)', lambda: ops.gemm(h, Wq, qkv, bias=bq), ((M + 255) // 256) * 9),
         ('fc1 gelu_dsave (N=3072, K=768)', lambda: ops.gemm(h, W1, g2, bias=b1, act='gelu_dsave', C2=u), ((M + 255) // 256) * 12),
         ('fc1b plain16   (N=768, K=3072)', lambda: ops.gemm(gact, W2, dh), ((M + 223) // 224) * 3)]
names = ['acc init (bias)', 'K loop - 2', 'set next tile', 'last 2 K-tiles', 'epilogue (row 0)']
for name, fn, nt in cases:
    trace = torch.zeros(nt + 64, 8, dtype=torch.int64, device='cuda')
    for _ in range(3): fn()
    lib.reid_debug_gemm_trace(ctypes.c_void_p(trace.data_ptr()))
    fn(); torch.cuda.synchronize()
    lib.reid_debug_gemm_trace(ctypes.c_void_p(0))
    t = trace.cpu().numpy().astype(np.int64)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    ts = (t[:, :7] - t0) / 100.0
    dd = np.diff(ts[:, :6], axis=1)
    print(f'== {name}: span {ts[:, 5].max():.1f} us, {len(t)} tiles traced')
    for i, n in enumerate(names):
        print(f'   {n:20s}: mean {dd[:, i].mean():6.2f}  p10 {np.percentile(dd[:, i], 10):6.2f}  p50 {np.percentile(dd[:, i], 50):6.2f}  p90 {np.percentile(dd[:, i], 90):6.2f} us')
    print(f'   epilogue end row 1 - row 0: mean {(ts[:, 6] - ts[:, 5]).mean():6.2f} us;  tile period: {(ts[:, 5] - ts[:, 0]).mean():6.2f} us')
    hw = t[:, 7]; xcc = (hw >> 32) & 0xf; hwid = hw & 0xffffffff
    cu = ((xcc << 8) | ((hwid >> 8) & 0xff)).astype(np.int64)
    c0 = np.unique(cu)[3]
    print('   CU', hex(int(c0)), ':', '  '.join('[' + ' '.join('%.1f' % x for x in ts[i, :7]) + ']' for i in sorted(np.where(cu == c0)[0], key=lambda i: ts[i, 0])[:6]))
