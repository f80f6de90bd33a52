"""Retrieval for 8..512 queries against the 200k gallery: wall time per call (host + device, synchronised loop) per query count.
With a query count as argument: that case only, 50 calls (for rocprofv3 --kernel-trace --stats: device time per kernel)."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prcv2025reid_amd.retrieval import GalleryIndex


def main():
    dev = torch.device('cuda', 0)
    Ng, D, k = 200000, 512, 10
    g = torch.Generator(device=dev).manual_seed(2)
    G = torch.nn.functional.normalize(torch.randn(Ng, D, device=dev, generator=g), dim=1)
    index = GalleryIndex(G, normalized=True)
    only = int(sys.argv[1]) if len(sys.argv) > 1 else None
    out = {}
    for Nq in ([only] if only else [5, 8, 16, 32, 64, 128, 256, 512]):
        Q = torch.nn.functional.normalize(torch.randn(Nq, D, device=dev, generator=g), dim=1)
        for _ in range(3):
            idx, sc = index.topk(Q, k=k, normalized=True)
        torch.cuda.synchronize()
        reps = 50
        t0 = time.perf_counter()
        for _ in range(reps):
            idx, sc = index.topk(Q, k=k, normalized=True)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / reps
        sim = Q.double() @ G.double().t()
        ref = torch.argsort(sim.float(), dim=1, descending=True, stable=True)[:, :k]
        out[Nq] = {'us_per_call': round(t * 1e6, 1), 'gallery16_GBps': round(Ng * D * 2 / t / 1e9, 1), 'identical_to_f64_order': bool((ref == idx.long()).all())}
        print(Nq, out[Nq], flush=True)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
