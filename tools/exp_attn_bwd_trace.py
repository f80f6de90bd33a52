"""Per-workgroup timeline of attn_bwd_fused_kernel<7> (library built with -DREID_ATTN_TRACE: tools/build_variant.sh attntrace -DREID_ATTN_TRACE;
REID_LIB_BF16=prcv2025reid_amd/csrc/libreid_hip_attntrace.so): s_memrealtime (100 MHz) of wave 0 at the phase boundaries, 256 images x 12 heads."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from prcv2025reid_amd import ops, _lib
n_img, S, d, heads = 256, 197, 768, 12
M = n_img * S
T16 = _lib.t16()
g = torch.Generator(device='cuda').manual_seed(0)
qkv = torch.randn(M, 3 * d, device='cuda', generator=g).to(T16)
o = torch.empty(M, d, device='cuda', dtype=T16); lse = torch.empty(n_img, heads, S, device='cuda')
do = torch.randn(M, d, device='cuda', generator=g).to(T16); dqkv = torch.empty_like(qkv); delta = torch.empty_like(lse)
ops.attn_fwd(qkv, o, lse, n_img, S, heads)
lib = _lib.lib()
_lib.check(lib.reid_set_knob(b'ATTN_BWD', -1))
nwg = n_img * heads
trace = torch.zeros(nwg, 8, dtype=torch.int64, device='cuda')
for _ in range(3): ops.attn_bwd(qkv, o, do, lse, dqkv, delta, n_img, S, heads)
torch.cuda.synchronize()
lib.reid_debug_attn_bwd_trace(ctypes.c_void_p(trace.data_ptr()))
ops.attn_bwd(qkv, o, do, lse, dqkv, delta, n_img, S, heads)
torch.cuda.synchronize()
lib.reid_debug_attn_bwd_trace(ctypes.c_void_p(0))
t = trace.cpu().numpy().astype(np.int64)
ts = (t[:, :7] - t[:, 0].min()) / 100.0
names = ['stage (start -> 4 images landed)', 'delta + barrier', 'phase 1 (7 query tiles)', 'dK/dV stores issued', 'phase 2 (7 key tiles)', 'dQ stores issued']
dd = np.diff(ts, axis=1)
print('kernel span %.1f us, %d workgroups' % (ts[:, 6].max(), nwg))
for i, n in enumerate(names):
    print(f'{n:36s}: mean {dd[:, i].mean():6.2f}  p10 {np.percentile(dd[:, i], 10):6.2f}  p50 {np.percentile(dd[:, i], 50):6.2f}  p90 {np.percentile(dd[:, i], 90):6.2f} us')
life = ts[:, 6] - ts[:, 0]
print('workgroup lifetime (wave 0): mean %.2f  p10 %.2f  p90 %.2f us' % (life.mean(), np.percentile(life, 10), np.percentile(life, 90)))

# ---- the persistent form: stamps per ITEM (wave 0 of the workgroup that owns it)
_lib.check(lib.reid_set_knob(b'ATTN_BWD', 3))
trace.zero_()
for _ in range(3): ops.attn_bwd(qkv, o, do, lse, dqkv, delta, n_img, S, heads)
torch.cuda.synchronize()
lib.reid_debug_attn_bwd_trace(ctypes.c_void_p(trace.data_ptr()))
ops.attn_bwd(qkv, o, do, lse, dqkv, delta, n_img, S, heads)
torch.cuda.synchronize()
lib.reid_debug_attn_bwd_trace(ctypes.c_void_p(0))
t = trace.cpu().numpy().astype(np.int64)
ts = (t - t[:, 0].min()) / 100.0
names = ['top: wait (this item landed) + barrier', 'K/V staging issue + delta + barrier', 'phase 1', 'own frags, barrier, stage next Q/dO/O, wait K/V, barrier',
         'dK/dV stores issued', 'phase 2', 'dQ stores issued']
dd = np.diff(ts, axis=1)
print('persistent: kernel span %.1f us, %d items' % (ts[:, 7].max(), nwg))
for i, n in enumerate(names):
    print(f'{n:60s}: mean {dd[:, i].mean():6.2f}  p10 {np.percentile(dd[:, i], 10):6.2f}  p50 {np.percentile(dd[:, i], 50):6.2f}  p90 {np.percentile(dd[:, i], 90):6.2f} us')
life = ts[:, 7] - ts[:, 0]
print('item time (wave 0): mean %.2f  p10 %.2f  p90 %.2f us' % (life.mean(), np.percentile(life, 10), np.percentile(life, 90)))
