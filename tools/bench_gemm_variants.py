"""Model-shaped GEMM variants (epilogues as the engine uses them), per tile config (REID_GEMM_TILE)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prcv2025reid_amd import ops, _lib
T16 = _lib.t16()
M, d, ff, Rp = 64 * 4 * 197, 768, 3072, 32
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
h = rnd(M, d); gact = rnd(M, ff); T = rnd(M, 3 * Rp)
Wqkv = rnd(3 * d, d, scale=0.03); Bq = rnd(3 * d, Rp, scale=0.1); bq = rnd(3 * d, dt=torch.float32)
Wo = rnd(d, d, scale=0.03); Bo = rnd(d, Rp, scale=0.1); bo = rnd(d, dt=torch.float32)
W1 = rnd(ff, d, scale=0.03); B1 = rnd(ff, Rp, scale=0.1); b1 = rnd(ff, dt=torch.float32)
W2 = rnd(d, ff, scale=0.03)
x = rnd(M, d, dt=torch.float32); xo = torch.empty(M, d, device='cuda')
qkv = torch.empty(M, 3 * d, device='cuda', dtype=T16); u = torch.empty(M, ff, device='cuda', dtype=T16); g2 = torch.empty(M, ff, device='cuda', dtype=T16)
dh = torch.empty(M, d, device='cuda', dtype=T16)
cases = [
 ('qkv  bf16+bias+lora(g)', 2.0*M*3*d*(d+Rp), lambda: ops.gemm(h, Wqkv, qkv, A2=T, B2=Bq, K2=Rp, k2_group_n=d, bias=bq)),
 ('out  f32+bias+R(f32)  ', 2.0*M*d*(d+Rp), lambda: ops.gemm(h, Wo, xo, A2=T[:, :Rp], B2=Bo, K2=Rp, bias=bo, R=x)),
 ('fc1  bf16+gelu+C2     ', 2.0*M*ff*(d+Rp), lambda: ops.gemm(h, W1, g2, A2=T[:, :Rp], B2=B1, K2=Rp, bias=b1, act='gelu', C2=u)),
 ('fc2  f32+bias+R K=3072', 2.0*M*d*(ff+Rp), lambda: ops.gemm(gact, W2, xo, A2=T[:, :Rp], B2=Bo, K2=Rp, bias=bo, R=x)),
 ('fc2b bf16+dgelu(aux)  ', 2.0*M*ff*(d+Rp), lambda: ops.gemm(h, W1, g2, A2=T[:, :Rp], B2=B1, K2=Rp, act='dgelu', aux=u)),
 ('fc1d bf16+gelu+dsave  ', 2.0*M*ff*(d+Rp), lambda: ops.gemm(h, W1, g2, A2=T[:, :Rp], B2=B1, K2=Rp, bias=b1, act='gelu_dsave', C2=u)),
 ('fc2m bf16*aux         ', 2.0*M*ff*(d+Rp), lambda: ops.gemm(h, W1, g2, A2=T[:, :Rp], B2=B1, K2=Rp, act='mul_aux', aux=u)),
 ('fc1b bf16 K=3072      ', 2.0*M*d*(ff+Rp), lambda: ops.gemm(gact, W2, dh, A2=T[:, :Rp], B2=Bo, K2=Rp)),
 ('qkvb bf16 K=2304+96   ', 2.0*M*d*(3*d+3*Rp), lambda: ops.gemm(qkv, rnd(d, 3*d, scale=0.03), dh, A2=T, B2=rnd(d, 3*Rp, scale=0.1), K2=3*Rp)),
]
# variants = knob settings, e.g. VARIANTS="base:GEMM_PERSIST=0;pers:GEMM_PERSIST=1,GEMM_STAGGER=0;stag:GEMM_PERSIST=1,GEMM_STAGGER=100"
import ctypes
def set_knobs(spec):
    for kv in spec.split(','):
        if kv:
            k, v = kv.split('=')
            _lib.check(_lib.lib().reid_set_knob(k.encode(), int(v)))
variants = [v.split(':') for v in os.environ.get('VARIANTS', 'base:GEMM_PERSIST=0;stag:GEMM_PERSIST=1').split(';')]
knobs = sorted({kv.split('=')[0] for _, spec in variants for kv in spec.split(',') if kv})
tot = {n: 0.0 for n, _ in variants}
for name, fl, fn in cases:
    res = []
    for rnd_ in range(3):                      # interleaved rounds in ONE process (cdna guide rule 24)
        for vn, spec in variants:
            for k in knobs:
                _lib.check(_lib.lib().reid_set_knob(k.encode(), -1))
            set_knobs(spec)
            res.append((vn, timeit(fn, reps=5)))
    best = {vn: min(u for tt, u in res if tt == vn) for vn, _ in variants}
    for vn in best: tot[vn] += best[vn]
    print(name + ': ' + '  '.join(f'{vn}: {best[vn]:7.1f} us {fl/best[vn]/1e6:6.1f} TF' for vn, _ in variants), flush=True)
print('sum over the seven shapes: ' + '  '.join(f'{vn}: {tot[vn]:7.1f} us' for vn, _ in variants), flush=True)
