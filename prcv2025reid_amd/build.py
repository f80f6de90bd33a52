"""Build libreid_hip.so for gfx950 with hipcc (in-tree, no torch linkage).

    python -m prcv2025reid_amd.build [--force]

The library is plain HIP behind a C ABI (include/reid_hip.h); the Python side
loads it with ctypes (prcv2025reid_amd/_lib.py).  hipcc cross-compiles without a
GPU, so this also runs in the CPU-only build container.
"""
import glob
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB = os.path.join(CSRC, 'libreid_hip.so')            # bf16 operands
LIB_F16 = os.path.join(CSRC, 'libreid_hip_f16.so')    # IEEE f16 operands (same sources, -DREID_FLAVOR_F16)
FLAVORS = (('bf16', LIB, []), ('f16', LIB_F16, ['-DREID_FLAVOR_F16']))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wno-unused-value'] + os.environ.get('REID_EXTRA_HIPCC_FLAGS', '').split()


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def needs_build() -> bool:
    if not (os.path.exists(LIB) and os.path.exists(LIB_F16)):
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(LIB_F16))
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + \
        [os.path.join(os.path.dirname(CSRC), '..', 'include', 'reid_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    procs = []
    for flavor, lib, defs in FLAVORS:
        for src in sources():
            obj = src[:-4] + f'.{flavor}.o'
            cmd = [HIPCC] + FLAGS + defs + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, obj, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{out}')
        if verbose and out.strip():
            print(out)
    for flavor, lib, defs in FLAVORS:
        objs = [src[:-4] + f'.{flavor}.o' for src in sources()]
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
