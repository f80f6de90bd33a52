"""The fused adapter-gradient kernel alone on the chip (step shape: 50 432 rows, N = 768, Rp = 32) against the two launches it replaces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prcv2025reid_amd import ops, _lib

M, N, Rp, r = 50432, 768, 32, 8
g = torch.Generator(device='cuda').manual_seed(0)
dY = torch.randn(M, N, device='cuda', generator=g).to(_lib.t16())
mods_row = (torch.arange(M, device='cuda') // 197 // 64).view(-1, 1)
T = (torch.randn(M, Rp, device='cuda', generator=g) * ((torch.arange(Rp, device='cuda').view(1, -1) // r) == mods_row)).to(_lib.t16())
BT = (torch.randn(Rp, N, device='cuda', generator=g) * 0.1).to(_lib.t16())
mods = torch.arange(256, device='cuda', dtype=torch.int32) // 64
U = torch.empty(M, Rp, device='cuda', dtype=_lib.t16()); dB = torch.zeros(N, Rp, device='cuda')


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


t = timeit(lambda: ops.lora_bwd_fused(dY, T, BT, U, dB, mods, 197, r, 2.0))
print(f'one image per workgroup (r04 default, 256 workgroups): {t:7.1f} us = {M * N * 2 / t / 1e6:7.2f} TB/s of dY = {M * N * 2 / t / 1e3 / 256:6.1f} GB/s per workgroup')
_lib.check(_lib.lib().reid_set_knob(b'LORA_IMPL', 1))
for knob in (-1, 192, 768, 1536):
    _lib.check(_lib.lib().reid_set_knob(b'TN_BLOCKS', knob))
    t = timeit(lambda: ops.lora_bwd_fused(dY, T, BT, U, dB, mods, 197, r, 2.0))
    slabs = 64 if knob < 0 else knob // 6
    print(f'fused, {slabs:4d} slabs: {t:7.1f} us = {M * N * 2 / t / 1e6:7.2f} TB/s of dY = {M * N * 2 / t / 1e3 / slabs:6.1f} GB/s per workgroup')
_lib.check(_lib.lib().reid_set_knob(b'TN_BLOCKS', -1))
_lib.check(_lib.lib().reid_set_knob(b'LORA_IMPL', -1))
t1 = timeit(lambda: ops.gemm(dY, BT, U, img_mod=mods, mask_r=r, mask_period=Rp, rows_per_img=197, alpha=2.0))
t2 = timeit(lambda: ops.gemm_tn(dY, T, dB, beta=1.0))
print(f'two launches: U {t1:.1f} us + dB {t2:.1f} us = {t1 + t2:.1f} us')

# ---- dA = U^T X: one image per workgroup (r04) against reid_gemm_tn
for K, G in ((768, 1), (3072, 1), (768, 3)):
    X = torch.randn(M, K, device='cuda', generator=g).to(_lib.t16())
    keepG = ((torch.arange(G * Rp, device='cuda').view(1, -1) % Rp) // r) == mods_row
    UU = (torch.randn(M, G * Rp, device='cuda', generator=g) * keepG).to(_lib.t16())
    dA = torch.zeros(G * Rp, K, device='cuda')
    t1 = timeit(lambda: ops.lora_da_fused(X, UU, dA, mods, 197, r, n_groups=G))
    t2 = timeit(lambda: ops.gemm_tn(UU, X, dA, beta=1.0))
    print(f'dA K={K} groups={G}: image kernel {t1:6.1f} us = {M * K * 2 / t1 / 1e6:5.2f} TB/s of X | gemm_tn {t2:6.1f} us = {M * K * 2 / t2 / 1e6:5.2f} TB/s')
