"""reid_merge_lora_table at the model's size (48 linears of ViT-B/16, r = 8, four modalities): us per launch, TB/s on its 1.7 GB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prcv2025reid_amd import ops
d, ff, r, nmod, L = 768, 3072, 8, 4, 12
Rp = 32
shapes = [(3 * d, d, 3), (d, d, 1), (ff, d, 1), (d, ff, 1)]
g = torch.Generator(device='cuda').manual_seed(0)
Ws, rows, off, woff = [], [], 0, 0
arena_n = sum(G * Rp * K + N * Rp for N, K, G in shapes) * L
arena = torch.randn(arena_n, device='cuda', generator=g) * 0.1
for l in range(L):
    for N, K, G in shapes:
        W = torch.randn(N, K, device='cuda', generator=g) * 0.03; Ws.append(W)
        a = off; off += G * Rp * K; b = off; off += N * Rp
        rows.append([W.data_ptr(), a, b, woff, woff + nmod * N * K, N, K, G]); woff += 2 * nmod * N * K
table = torch.tensor(rows, dtype=torch.int64, device='cuda')
weff = torch.empty(woff, device='cuda', dtype=torch.bfloat16 if os.environ.get('REID_FLAVOR', 'bf16') == 'bf16' else torch.float16)
tiles = max((N // 64) * (K // 64) for N, K, G in shapes)
byts = sum(N * K * 4 + 2 * nmod * N * K * 2 for N, K, G in shapes) * L
def run(): ops.merge_lora_table(table, len(rows), tiles, arena, weff, Rp, r, nmod, 2.0)
for _ in range(3): run()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f'merge_lora_table: {us:7.1f} us  {byts / us / 1e6:5.2f} TB/s on {byts / 1e9:.2f} GB')
