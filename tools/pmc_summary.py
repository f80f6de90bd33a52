"""rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE output (two separate passes, --output-format csv) -> profiles/<tag>_pmc_traffic.{md,json}.

usage: python tools/pmc_summary.py <fetch_dir> <write_dir> <tag> "<command that was profiled>"
traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes (gfx950: FETCH_SIZE counts wide coalesced reads at half;
MI355X_MICROARCH.md, section HBM).
"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def load(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') != counter:
                continue
            name = re.sub(r'^void ', '', r['Kernel_Name'])
            name = re.sub(r'\(anonymous namespace\)::', '', name)
            name = re.sub(r'\(.*$', '', name)
            a = acc[name]; a[0] += 1; a[1] += float(r['Counter_Value'])
    return acc


def main():
    fd, wd, tag, cmd = sys.argv[1:5]
    F, W = load(fd, 'FETCH_SIZE'), load(wd, 'WRITE_SIZE')
    out = {}
    for k in sorted(set(F) | set(W), key=lambda k: -(2 * F.get(k, [0, 0])[1] + W.get(k, [0, 0])[1])):
        n = max(F.get(k, [0, 0])[0], W.get(k, [0, 0])[0])
        if n == 0:
            continue
        f = F.get(k, [0, 0.0])[1] / max(1, F.get(k, [0, 0])[0]); w = W.get(k, [0, 0.0])[1] / max(1, W.get(k, [0, 0])[0])
        out[k] = dict(launches=n, fetch_kb=f, write_kb=w, traffic_bytes_per_launch=(2 * f + w) * 1024)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    json.dump(out, open(os.path.join(root, 'profiles', f'{tag}_pmc_traffic.json'), 'w'), indent=1)
    with open(os.path.join(root, 'profiles', f'{tag}_pmc_traffic.md'), 'w') as fh:
        fh.write(f'# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `{cmd}`, 1x MI355X\n\n')
        fh.write('traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes  (gfx950: FETCH_SIZE counts wide coalesced reads at half, '
                 'MI355X_MICROARCH.md "HBM")\n\n| kernel | launches | FETCH_SIZE avg (KB) | WRITE_SIZE avg (KB) | traffic / launch (MB) |\n|---|---|---|---|---|\n')
        for k, v in out.items():
            fh.write(f"| `{k[:120]}` | {v['launches']} | {v['fetch_kb']:.0f} | {v['write_kb']:.0f} | {v['traffic_bytes_per_launch'] / 1e6:.1f} |\n")
    print(f'{len(out)} kernels')


if __name__ == '__main__':
    main()
