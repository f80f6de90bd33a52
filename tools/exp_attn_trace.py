"""(Needs profiles/r02_attn_persistent_experiment.patch applied: the trace hooks are part of that experiment, not of the shipped kernels.)
Per-workgroup timeline of attn_fwd_kernel<7,true> (library built with -DREID_ATTN_TRACE): s_memrealtime (100 MHz) at the phase
boundaries of every workgroup + its CU, for 256 images x 12 heads.  Prints phase durations and how the workgroups of one CU overlap."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from prcv2025reid_amd import ops, _lib
n_img, S, d, heads = 256, 197, 768, 12
M = n_img * S
T16 = _lib.t16()
g = torch.Generator(device='cuda').manual_seed(0)
qkv = (torch.randn(M, 3 * d, device='cuda', generator=g)).to(T16)
o = torch.empty(M, d, device='cuda', dtype=T16); lse = torch.empty(n_img, heads, S, device='cuda')
nwg = n_img * heads
trace = torch.zeros(nwg, 8, dtype=torch.int64, device='cuda')
lib = _lib.lib()
for _ in range(3): ops.attn_fwd(qkv, o, lse, n_img, S, heads)
lib.reid_debug_attn_trace(ctypes.c_void_p(trace.data_ptr()))
ops.attn_fwd(qkv, o, lse, n_img, S, heads)
torch.cuda.synchronize()
lib.reid_debug_attn_trace(ctypes.c_void_p(0))
t = trace.cpu().numpy().astype(np.int64)
t0 = t[:, 0].min()
ts = (t[:, :6] - t0) / 100.0          # us
hw = t[:, 6]
xcc = (hw >> 32) & 0xf; hwid = hw & 0xffffffff
cu = ((xcc << 8) | ((hwid >> 8) & 0xff)).astype(np.int64)     # xcc, se/sh/cu bits
names = ['stage (start -> K,V,Q landed)', 'sweep 1', 'sweep 2', 'store issue', 'store ack']
d = np.diff(ts, axis=1)
print('kernel span %.1f us, %d workgroups, %d distinct CUs' % (ts[:, 5].max(), nwg, len(np.unique(cu))))
for i, n in enumerate(names):
    print(f'{n:32s}: mean {d[:, i].mean():6.2f}  p10 {np.percentile(d[:, i], 10):6.2f}  p50 {np.percentile(d[:, i], 50):6.2f}  p90 {np.percentile(d[:, i], 90):6.2f} us')
life = ts[:, 5] - ts[:, 0]
print('workgroup lifetime: mean %.2f  p10 %.2f  p90 %.2f us' % (life.mean(), np.percentile(life, 10), np.percentile(life, 90)))
# per CU: how many workgroups, busy fraction (union of lifetimes), gaps between end of one and start of next in the same slot
gaps = []; conc = []
for c in np.unique(cu):
    idx = np.where(cu == c)[0]
    iv = sorted((ts[i, 0], ts[i, 5]) for i in idx)
    # average number of resident workgroups over the kernel span
    conc.append(sum(b - a for a, b in iv) / ts[:, 5].max())
    ends = sorted(b for a, b in iv); starts = sorted(a for a, b in iv)
    for s_ in starts[2:]:
        prev = max(e for e in ends if e <= s_ + 1e-9) if any(e <= s_ + 1e-9 for e in ends) else None
        if prev is not None: gaps.append(s_ - prev)
print('resident workgroups per CU (time average): mean %.2f  min %.2f  max %.2f' % (np.mean(conc), np.min(conc), np.max(conc)))
print('gap from a workgroup end to the next start on the same CU: mean %.2f  p50 %.2f  p90 %.2f us' % (np.mean(gaps), np.percentile(gaps, 50), np.percentile(gaps, 90)))
c0 = np.unique(cu)[0]
print('timeline of CU', hex(int(c0)))
for i in sorted(np.where(cu == c0)[0], key=lambda i: ts[i, 0]):
    print('  wg %4d: ' % i + '  '.join('%6.2f' % v for v in ts[i]))
