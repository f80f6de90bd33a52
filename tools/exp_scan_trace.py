"""Per-workgroup timeline of scan::scan_filter_kernel (library built with tools/build_variant.sh scantrace -DREID_SCAN_TRACE;
REID_LIB_BF16=prcv2025reid_amd/csrc/libreid_hip_scantrace.so): s_memrealtime (100 MHz) of thread 0 at the phase boundaries."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from prcv2025reid_amd import _lib
from prcv2025reid_amd.retrieval import GalleryIndex
dev = torch.device('cuda', 0)
Ng, D, k = 200000, 512, 10
g = torch.Generator(device=dev).manual_seed(2)
G = torch.nn.functional.normalize(torch.randn(Ng, D, device=dev, generator=g), dim=1)
index = GalleryIndex(G, normalized=True)
lib = _lib.lib()
for Nq in [int(a) for a in sys.argv[1:]] or [5, 128]:
    Q = torch.nn.functional.normalize(torch.randn(Nq, D, device=dev, generator=g), dim=1)
    for _ in range(3): index.topk(Q, k=k, normalized=True)
    torch.cuda.synchronize()
    trace = torch.zeros(256, 8, dtype=torch.int64, device=dev)
    lib.reid_debug_scan_trace(ctypes.c_void_p(trace.data_ptr()))
    index.topk(Q, k=k, normalized=True)
    torch.cuda.synchronize()
    lib.reid_debug_scan_trace(ctypes.c_void_p(0))
    t = trace.cpu().numpy().astype(np.int64); t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    names = ['start', 'prologue done (queries in registers, step 0 issued)', 'main loop done', 'bars complete (polls)', 'revisit done', 'flushed', None, 'step 0 landed + barrier']
    print(f'--- {Nq} queries: kernel span {(t[:, 5].max() - t0) / 100.0:.1f} us')
    for i in (0, 1, 7, 2, 3, 4, 5):
        v = (t[:, i] - t0) / 100.0
        print(f'  {names[i]:52s}: mean {v.mean():7.2f}  min {v.min():7.2f}  max {v.max():7.2f} us')
    ph = np.diff(np.stack([t[:, j] for j in (0, 1, 2, 3, 4, 5)], axis=1), axis=1) / 100.0
    for n, j in zip(('prologue', 'main loop', 'polls', 'revisit', 'flush'), range(5)):
        v = ph[:, j]
        print(f'  phase {n:10s}: p10 {np.percentile(v, 10):6.2f}  p50 {np.percentile(v, 50):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f} us  (workgroup {int(v.argmax())})')
    polls = t[:, 6] & 0xffff; nrev = (t[:, 6] >> 16) & 0xffff; fv = (t[:, 6] >> 32) & 0xffff
    print('  polls: mean %.1f max %d;  steps revisited: mean %.2f max %d;  first step with a bar (wave 0): mean %.2f' % (polls.mean(), polls.max(), nrev.mean(), nrev.max(), fv.mean()))
