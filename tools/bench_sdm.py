"""Fused SDM loss (csrc/sdm.hip) at the size that exposes the hardware (N = M = 8192, D = 512) and at
training sizes.  Prints whole-call times (HIP events) and the fraction of the op's COMPULSORY bytes / 8 TB/s they correspond to
(section 8d: 5 B D 4 + 8 B bytes forward, the same again for the gradients).  Run under `rocprofv3 --kernel-trace --stats` for
per-kernel durations."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prcv2025reid_amd import ops


def run(P, N, Mg, D=512, reps=5):
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.randn(P * N, D, device=dev, generator=g); gal = torch.randn(Mg, D, device=dev, generator=g)
    ql = torch.randint(0, 1000, (N,), device=dev, generator=g); gl = torch.randint(0, 1000, (Mg,), device=dev, generator=g)
    qv = (torch.rand(P * N, device=dev, generator=g) > 0.1).to(torch.uint8); gv = (torch.rand(Mg, device=dev, generator=g) > 0.1).to(torch.uint8)
    ws = torch.empty(ops.sdm_ws_floats(P, N, Mg, D), device=dev); res = torch.zeros(2 * P, device=dev)
    dq = torch.zeros(P * N, D, device=dev); dg = torch.zeros(Mg, D, device=dev)
    gs = torch.ones(P, device=dev)
    fwd = lambda: ops.sdm_fwd(q, gal, ql, gl, qv, gv, 0.2, ws, res, P=P)
    bwd = lambda: ops.sdm_bwd(q, gal, ql, gl, qv, gv, 0.2, ws, gs, dq, dg, P=P)
    out = dict(P=P, N=N, Mg=Mg, D=D, ws_MB=ws.numel() * 4 / 1e6)
    for name, fn in (('fwd', fwd), ('bwd', bwd)):
        fn(); fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        out[name + '_us'] = e0.elapsed_time(e1) / reps * 1e3
    flops = 2.0 * P * N * Mg * D
    comp = ((P * N + Mg) * D * 4 + (P * N + Mg) * 8)            # read the features + labels once
    out['loss0'] = float(res[0])
    out['fwd_tflops_fp32'] = flops / (out['fwd_us'] * 1e-6) / 1e12
    out['bwd_tflops_fp32'] = 3 * flops / (out['bwd_us'] * 1e-6) / 1e12
    out['compulsory_MB_fwd'] = comp / 1e6
    out['hbm_frac_whole_op_fwd'] = comp / (out['fwd_us'] * 1e-6) / 8e12
    out['hbm_frac_whole_op_fwd_bwd'] = 2 * comp / ((out['fwd_us'] + out['bwd_us']) * 1e-6) / 8e12
    return out


if __name__ == '__main__':
    big = int(os.environ.get('SDM_N', '8192'))
    for P, N, Mg in ((1, big, big), (4, 1024, 1024), (4, 128, 128), (4, 64, 64)):
        print(json.dumps(run(P, N, Mg)), flush=True)
