"""Where the time of lora_bwd_image_kernel goes: the kernel with parts left out (wrong results; library built with -DREID_EXPERIMENTS:
tools/build_variant.sh exp -DREID_EXPERIMENTS; REID_LIB_BF16=prcv2025reid_amd/csrc/libreid_hip_exp.so).  Bits: 1 no U arithmetic, 2 no U output,
4 no dB arithmetic, 8 no dB flush, 16 no T staging, 32 no B^T fragment prologue."""
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
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for bits in (0, 1, 2, 3, 4, 8, 12, 16, 32, 15, 63):
    _lib.check(_lib.lib().reid_set_knob(b'LORA_IMPL', 16 + bits))
    t = timeit(lambda: ops.lora_bwd_fused(dY, T, BT, U, dB, mods, 197, r, 2.0))
    print(f'left out {bits:2d} ({bits:06b}): {t:6.1f} us')
_lib.check(_lib.lib().reid_set_knob(b'LORA_IMPL', -1))
