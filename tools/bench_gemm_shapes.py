"""The eight ViT-B/16 GEMMs of a training step in their merged-weight form (row groups, no K extension), A/B over knob settings
in ONE process, interleaved rounds.  VARIANTS="name:KNOB=v,KNOB=v;name2:..." (knobs of reid_set_knob)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prcv2025reid_amd import ops, _lib
T16 = _lib.t16()
S, d, ff = 197, 768, 3072
n_per = int(os.environ.get('IMGS_PER_MOD', '64'))
M = 4 * n_per * S
ends = [(i + 1) * n_per * S for i in range(4)]; rg = (ends, [0, 1, 2, 3])
g = torch.Generator(device='cuda').manual_seed(0)
def rnd(*shape, scale=1.0, dt=None):
    return (torch.randn(*shape, device='cuda', generator=g) * scale).to(dt or T16)
def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
h = rnd(M, d); gact = rnd(M, ff)
Wqkv = rnd(4, 3 * d, d, scale=0.03); bq = rnd(3 * d, dt=torch.float32)
Wo = rnd(4, d, d, scale=0.03); bo = rnd(d, dt=torch.float32)
W1 = rnd(4, ff, d, scale=0.03); b1 = rnd(ff, dt=torch.float32)
W2 = rnd(4, d, ff, scale=0.03)
WqT = rnd(4, d, 3 * d, scale=0.03)
x = rnd(M, d, dt=torch.float32); xo = torch.empty(M, d, device='cuda')
qkv = torch.empty(M, 3 * d, device='cuda', dtype=T16); u = torch.empty(M, ff, device='cuda', dtype=T16); g2 = torch.empty(M, ff, device='cuda', dtype=T16)
dh = torch.empty(M, d, device='cuda', dtype=T16)
cases = [
 ('qkv  16+bias          ', 2.0*M*3*d*d, lambda: ops.gemm(h, Wqkv, qkv, bias=bq, row_groups=rg)),
 ('out  f32+bias+R       ', 2.0*M*d*d, lambda: ops.gemm(h, Wo, xo, bias=bo, R=x, row_groups=rg)),
 ('fc1  16+gelu+dsave    ', 2.0*M*ff*d, lambda: ops.gemm(h, W1, g2, bias=b1, act='gelu_dsave', C2=u, row_groups=rg)),
 ('fc2  f32+bias+R K=3072', 2.0*M*d*ff, lambda: ops.gemm(gact, W2, xo, bias=bo, R=x, row_groups=rg)),
 ('fc2b 16*aux           ', 2.0*M*ff*d, lambda: ops.gemm(h, W1, g2, act='mul_aux', aux=u, row_groups=rg)),
 ('fc1b 16 K=3072        ', 2.0*M*d*ff, lambda: ops.gemm(gact, W2, dh, row_groups=rg)),
 ('outb 16               ', 2.0*M*d*d, lambda: ops.gemm(h, Wo, dh, row_groups=rg)),
 ('qkvb 16 K=2304        ', 2.0*M*d*3*d, lambda: ops.gemm(qkv, WqT, dh, row_groups=rg)),
]
def set_knobs(spec):
    for kv in spec.split(','):
        if kv:
            k, v = kv.split('=')
            _lib.check(_lib.lib().reid_set_knob(k.encode(), int(v)))
variants = [v.split(':') for v in os.environ.get('VARIANTS', 'pp:GEMM_PERSIST=0;pps:GEMM_PERSIST=1').split(';')]
knobs = sorted({kv.split('=')[0] for _, spec in variants for kv in spec.split(',') if kv})
tot = {n: 0.0 for n, _ in variants}; flt = 0.0
for name, fl, fn in cases:
    res = []
    for rnd_ in range(3):                      # interleaved rounds in ONE process (cdna guide rule 24)
        for vn, spec in variants:
            for k in knobs:
                _lib.check(_lib.lib().reid_set_knob(k.encode(), -1))
            set_knobs(spec)
            res.append((vn, timeit(fn, reps=5)))
    best = {vn: min(t for tt, t in res if tt == vn) for vn, _ in variants}
    for vn in best: tot[vn] += best[vn]
    flt += fl
    print(name + ': ' + '  '.join(f'{vn}: {best[vn]:7.1f} us {fl/best[vn]/1e6:6.1f} TF' for vn, _ in variants), flush=True)
print(f'sum over the eight shapes ({flt/1e9:.0f} GFLOP): ' + '  '.join(f'{vn}: {tot[vn]:7.1f} us = {flt/tot[vn]/1e6:6.1f} TF' for vn, _ in variants), flush=True)
