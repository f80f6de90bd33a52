"""Attention forward / backward at the benchmark size (256 sequences x 197 tokens x 12 heads), A/B over knob settings in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prcv2025reid_amd import ops, _lib
T16 = _lib.t16()
n_seq, S, heads = int(os.environ.get('N_SEQ', '256')), 197, 12
d = heads * 64
g = torch.Generator(device='cuda').manual_seed(0)
qkv = torch.randn(n_seq * S, 3 * d, device='cuda', generator=g).to(T16)
o = torch.empty(n_seq * S, d, device='cuda', dtype=T16); lse = torch.empty(n_seq, heads, S, device='cuda')
do = torch.randn(n_seq * S, d, device='cuda', generator=g).to(T16); dqkv = torch.empty_like(qkv); delta = torch.empty_like(lse)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
variants = [v.split(':') for v in os.environ.get('VARIANTS', 'base:GEMM_PERSIST=1').split(';')]
res = {}
for rnd in range(3):
    for vn, spec in variants:
        for kv in spec.split(','):
            k, v = kv.split('='); _lib.check(_lib.lib().reid_set_knob(k.encode(), int(v)))
        f = timeit(lambda: ops.attn_fwd(qkv, o, lse, n_seq, S, heads))
        b = timeit(lambda: ops.attn_bwd(qkv, o, do, lse, dqkv, delta, n_seq, S, heads))
        res.setdefault(vn, []).append((f, b))
for vn, r in res.items():
    f = min(x[0] for x in r); b = min(x[1] for x in r)
    items = n_seq * heads
    print(f'{vn}: fwd {f:7.1f} us ({items * 100.8e3 / f / 1e6:5.2f} TB/s of the 310 MB floor traffic, {4 * S * S * 64 * items / f / 1e6:6.1f} TF)   bwd (delta + dkv + dq) {b:7.1f} us')
