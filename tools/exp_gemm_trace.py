"""Per-workgroup timeline of the 256 x 256 ping-pong GEMM (library built with -DREID_GEMM_TRACE): s_memrealtime (100 MHz) at workgroup
start / K loop end / epilogue issued / stores acknowledged + the CU, for the fc1 (GELU, two outputs) and fc2-backward shapes."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from prcv2025reid_amd import ops, _lib
T16 = _lib.t16()
M, d, ff = 64 * 4 * 197, 768, 3072
g = torch.Generator(device='cuda').manual_seed(0)
h = torch.randn(M, d, device='cuda', generator=g).to(T16); W1 = (torch.randn(ff, d, device='cuda', generator=g) * 0.03).to(T16)
b1 = torch.randn(ff, device='cuda', generator=g)
u = torch.empty(M, ff, device='cuda', dtype=T16); g2 = torch.empty(M, ff, device='cuda', dtype=T16)
gact = torch.randn(M, ff, device='cuda', generator=g).to(T16); W2 = (torch.randn(d, ff, device='cuda', generator=g) * 0.03).to(T16)
dh = torch.empty(M, d, device='cuda', dtype=T16)
lib = _lib.lib()
cases = [('fc1 gelu_dsave (N=3072, K=768)', lambda: ops.gemm(h, W1, g2, bias=b1, act='gelu_dsave', C2=u), 197 * 12),
         ('fc2b mul_aux   (N=3072, K=768)', lambda: ops.gemm(h, W1, g2, act='mul_aux', aux=u), 197 * 12),
         ('fc1b plain16   (N=768, K=3072)', lambda: ops.gemm(gact, W2, dh), 197 * 3)]
for name, fn, nwg in cases:
    trace = torch.zeros(nwg, 8, dtype=torch.int64, device='cuda')
    for _ in range(3): fn()
    lib.reid_debug_gemm_trace(ctypes.c_void_p(trace.data_ptr()))
    fn(); torch.cuda.synchronize()
    lib.reid_debug_gemm_trace(ctypes.c_void_p(0))
    t = trace.cpu().numpy().astype(np.int64)
    t0 = t[:, 0].min()
    ts = (t[:, :4] - t0) / 100.0
    hw = t[:, 6]; xcc = (hw >> 32) & 0xf; hwid = hw & 0xffffffff
    cu = ((xcc << 8) | ((hwid >> 8) & 0xff)).astype(np.int64)
    dd = np.diff(ts, axis=1)
    print(f'== {name}: span {ts[:, 3].max():.1f} us, {nwg} workgroups, {len(np.unique(cu))} CUs')
    for i, n in enumerate(('K loop (start -> last MFMA phase)', 'epilogue issue', 'store ack')):
        print(f'   {n:36s}: mean {dd[:, i].mean():6.2f}  p10 {np.percentile(dd[:, i], 10):6.2f}  p50 {np.percentile(dd[:, i], 50):6.2f}  p90 {np.percentile(dd[:, i], 90):6.2f} us')
    gaps = []
    for c in np.unique(cu):
        idx = sorted(np.where(cu == c)[0], key=lambda i: ts[i, 0])
        for a, b in zip(idx[:-1], idx[1:]):
            gaps.append(ts[b, 0] - ts[a, 3])
    gaps = np.array(gaps)
    print(f'   gap: end of a workgroup -> start of the next on the same CU: mean {gaps.mean():5.2f}  p10 {np.percentile(gaps, 10):5.2f}  p50 {np.percentile(gaps, 50):5.2f}  p90 {np.percentile(gaps, 90):5.2f} us')
    # how many workgroups are in their epilogue at the same time (sampled every 0.5 us)
    grid = np.arange(0, ts[:, 3].max(), 0.5)
    inepi = np.array([((ts[:, 1] <= x) & (x < ts[:, 3])).sum() for x in grid])
    inloop = np.array([((ts[:, 0] <= x) & (x < ts[:, 1])).sum() for x in grid])
    print(f'   workgroups in their epilogue at a time: mean {inepi.mean():.1f}  p90 {np.percentile(inepi, 90):.0f}  max {inepi.max()};  in the K loop: mean {inloop.mean():.1f}')
    c0 = np.unique(cu)[3]
    print('   CU', hex(int(c0)), ':', '  '.join('[%.1f %.1f %.1f %.1f]' % tuple(ts[i]) for i in sorted(np.where(cu == c0)[0], key=lambda i: ts[i, 0])))
