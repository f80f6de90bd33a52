"""CPU: the C-ABI libraries build, load and export every symbol include/reid_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, 'include', 'reid_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    names = re.findall(r'^\s*(?:const\s+char\s*\*|int64_t|int32_t|int)\s+(reid_\w+)\s*\(', src, flags=re.M)
    return sorted(set(names))


@pytest.fixture(scope='module')
def libs():
    from prcv2025reid_amd import build, _lib
    build.build(verbose=False)
    return {f: ctypes.CDLL(p) for f, p in _lib.LIB_PATHS.items()}


def test_header_declares_what_python_binds():
    from prcv2025reid_amd import _lib
    assert set(_lib.EXPORTS) == set(header_functions())


@pytest.mark.parametrize('flavor', ['bf16', 'f16'])
def test_library_exports_every_declared_symbol(libs, flavor):
    h = libs[flavor]
    missing = [n for n in header_functions() if not hasattr(h, n)]
    assert not missing, missing
    assert h.reid_version() == 200              # round 2 ABI (reid_set_knob, fused SDM)
    assert h.reid_flavor() == (1 if flavor == 'f16' else 0)


def test_argument_validation_needs_no_gpu(libs):
    # bad arguments are rejected on the host before any launch
    from prcv2025reid_amd._lib import GemmArgs
    h = libs['bf16']
    h.reid_last_error.restype = ctypes.c_char_p
    a = GemmArgs()
    assert h.reid_mer_gemm(ctypes.byref(a), None) == -1
    assert b'non-null' in h.reid_last_error()
    assert h.reid_attn_fwd(None, 0, None, None, 0, None, 1, 500, 12, 0, None) == -1
    assert h.reid_gemm_tn(None, None, None, 1, 8, 8, 8, 8, 8, ctypes.c_float(1), ctypes.c_float(0), None) == -1


def test_model_refuses_cpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from prcv2025reid_amd import _lib
    from prcv2025reid_amd.config import TrainingConfig
    from prcv2025reid_amd.model import CLIPBasedMultiModalReIDModel
    with pytest.raises(_lib.ReidHipError):
        CLIPBasedMultiModalReIDModel(TrainingConfig(device='cpu'))
