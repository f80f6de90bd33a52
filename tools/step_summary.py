"""Per-step kernel time by family from a rocprofv3 --stats kernel_stats.csv: python tools/step_summary.py <csv> <steps>"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
fam = [('gemm_pp', r'mer_gemm_pps?_kernel'), ('gemm128', r'mer_gemm_kernel<128, 128'), ('gemm_skinny', r'mer_gemm_kernel<(64|256|128), (32|64)'),
       ('gemm_tn', r'gemm_tn_kernel'), ('lora image (r04)', r'lora_bwd_image|lora_da_image|u_finish'), ('lora_fused(64 CUs)', r'lora_bwd_fused'), ('attn', r'attn_'), ('ln', r'(add_)?ln_(fwd|bwd)'), ('merge/pack', r'merge_lora|pack_table'), ('sdm', r'sdm_'),
       ('opt', r'opt_'), ('head', r'sgemm|small_attn|bnneck|ce_|masked_mean|eltwise|l2norm'), ('aten/other', r'.')]
tot = {f: [0.0, 0] for f, _ in fam}
for r in rows:
    for f, pat in fam:
        if re.search(pat, r['Name']):
            tot[f][0] += float(r['TotalDurationNs']); tot[f][1] += int(r['Calls']); break
allt = sum(v[0] for v in tot.values())
for f, _ in fam:
    print(f'{f:18s} {tot[f][0] / steps / 1e6:8.3f} ms/step  {tot[f][1] / steps:8.1f} launches/step')
print(f'{"sum":18s} {allt / steps / 1e6:8.3f} ms/step')
print('top kernels:')
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:22]:
    print(f"  {r['Name'][:96]:96s} {int(r['Calls']) / steps:7.1f}/step  avg {float(r['AverageNs']) / 1e3:8.1f} us  {float(r['TotalDurationNs']) / steps / 1e6:7.3f} ms/step")
