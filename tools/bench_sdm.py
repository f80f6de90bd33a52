"""SDM loss kernels at a size that exposes bandwidth (N = M = 8192, D = 512).

At the training sizes (N, M <= 1024) the SDM loss is launch-latency bound; here the masked-softmax statistics
(`sdm_side_kernel`, 4 bytes per entry) and the in-place gradient (`sdm_ds_kernel`, 8 bytes per entry) are HBM-bound, the
fp32 similarity GEMMs are VALU-bound.  Run under `rocprofv3 --kernel-trace --stats` for per-kernel durations; the script
itself prints whole-call times.
"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prcv2025reid_amd import ops


def main():
    N = Mg = int(os.environ.get('SDM_N', '8192'))
    D = 512
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.randn(N, D, device=dev, generator=g); gal = torch.randn(Mg, D, device=dev, generator=g)
    ql = torch.randint(0, 1000, (N,), device=dev, generator=g); gl = torch.randint(0, 1000, (Mg,), device=dev, generator=g)
    qv = (torch.rand(N, device=dev, generator=g) > 0.1).to(torch.uint8); gv = (torch.rand(Mg, device=dev, generator=g) > 0.1).to(torch.uint8)
    ws = torch.empty(ops.sdm_ws_floats(N, Mg), device=dev); res = torch.zeros(2, device=dev)
    dq = torch.zeros(N, D, device=dev); dg = torch.zeros(Mg, D, device=dev)
    gs = torch.tensor([1.0], device=dev)
    out = {}
    for name, fn in (('fwd', lambda: ops.sdm_fwd(q, gal, ql, gl, qv, gv, 0.2, ws, res)),):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        out[name + '_ms'] = (time.perf_counter() - t0) / 3 * 1e3
    loss = float(res[0])
    # backward consumes S in place: one forward per backward
    tb = 0.0
    for _ in range(3):
        ops.sdm_fwd(q, gal, ql, gl, qv, gv, 0.2, ws, res); torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.sdm_bwd(q, gal, ql, gl, qv, gv, 0.2, ws, gs, dq, dg)
        torch.cuda.synchronize()
        tb += time.perf_counter() - t0
    out['bwd_ms'] = tb / 3 * 1e3
    out.update(N=N, Mg=Mg, D=D, loss=loss, side_bytes=N * Mg * 4, ds_bytes=N * Mg * 8)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
