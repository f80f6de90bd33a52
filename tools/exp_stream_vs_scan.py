"""1-6 queries x 200k x 512: the one-pass fp32 form (stream=True, <= 4 queries) against the batched entry point (stream=False: the
query-resident 16-bit scan + exact re-score), wall time per GalleryIndex.topk call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prcv2025reid_amd.retrieval import GalleryIndex
dev = torch.device('cuda', 0)
Ng, D, k = 200000, 512, 10
g = torch.Generator(device=dev).manual_seed(2)
G = torch.nn.functional.normalize(torch.randn(Ng, D, device=dev, generator=g), dim=1)
index = GalleryIndex(G, normalized=True)
def timeit(fn, reps=200):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
for Nq in (1, 2, 3, 4, 5, 6):
    Q = torch.nn.functional.normalize(torch.randn(Nq, D, device=dev, generator=g), dim=1)
    ts = timeit(lambda: index.topk(Q, k=k, normalized=True, stream=True)) if Nq <= 4 else float('nan')
    tb = timeit(lambda: index.topk(Q, k=k, normalized=True, stream=False))
    same = torch.equal(index.topk(Q, k=k, normalized=True, stream=False)[0], index.topk(Q, k=k, normalized=True, stream=True)[0]) if Nq <= 4 else None
    print(f'{Nq} queries: one-pass fp32 {ts:6.1f} us | batched entry (scan) {tb:6.1f} us | same lists {same}', flush=True)
