"""Where does attn_fwd_kernel<7,true> spend its time?  REID_ATTN_DBG early exits: 3 = after staging K,V + barrier,
2 = + first pass (row maxima), 1 = everything but the output stores, 0 = full kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prcv2025reid_amd import ops, _lib
n_img, S, d, heads = 256, 197, 768, 12
M = n_img * S
T16 = _lib.t16()
g = torch.Generator(device='cuda').manual_seed(0)
qkv = (torch.randn(M, 3 * d, device='cuda', generator=g)).to(T16)
o = torch.empty(M, d, device='cuda', dtype=T16); lse = torch.empty(n_img, heads, S, device='cuda')
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def knob(name, v):
    _lib.check(_lib.lib().reid_set_knob(name.encode(), v))
for dbg in (3, 2, 1, 0):
    knob('ATTN_DBG', dbg if dbg else -1)
    print('dbg', dbg, round(timeit(lambda: ops.attn_fwd(qkv, o, lse, n_img, S, heads)), 1), 'us', flush=True)
# (the persistent-workgroup / stagger / split variants this script also timed in r02 live in profiles/r02_attn_persistent_experiment.patch;
#  results: profiles/r02_attn_persistent.log, r02_attn_stagger_split.log)
knob('ATTN_DBG', -1)
