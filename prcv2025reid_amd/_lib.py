"""ctypes binding of libreid_hip.so (include/reid_hip.h).  No CPU fallback exists:
if the library is missing or a call fails this module raises."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATHS = {'bf16': os.path.join(_HERE, 'csrc', 'libreid_hip.so'), 'f16': os.path.join(_HERE, 'csrc', 'libreid_hip_f16.so')}
if os.environ.get('REID_LIB_BF16'):                 # experiment builds (tools/): another bf16-flavor library file
    LIB_PATHS['bf16'] = os.environ['REID_LIB_BF16']
LIB_PATH = LIB_PATHS['bf16']
T16_DTYPES = {'bf16': torch.bfloat16, 'f16': torch.float16}
_flavor = os.environ.get('REID_T16', 'bf16')


def set_flavor(name: str):
    """Select the 16-bit MFMA operand format for everything launched afterwards ('bf16' or 'f16')."""
    global _flavor
    if name not in LIB_PATHS:
        raise ValueError(f'unknown 16-bit flavor {name!r}')
    _flavor = name


def flavor() -> str:
    return _flavor


def t16() -> torch.dtype:
    return T16_DTYPES[_flavor]


BF16, F32, F16 = 0, 1, 2          # reid_dtype: BF16 = the flavor's 16-bit format, F16 = IEEE half whatever the flavor
ACT_NONE, ACT_GELU, ACT_QUICK_GELU, ACT_RELU, ACT_DGELU, ACT_DQUICK_GELU, ACT_DRELU, ACT_MUL_AUX, ACT_GELU_DSAVE = range(9)

EXPORTS = [
    'reid_last_error', 'reid_version', 'reid_flavor', 'reid_check_device', 'reid_set_knob', 'reid_mer_gemm', 'reid_gemm_tn',
    'reid_layernorm_fwd', 'reid_layernorm_bwd', 'reid_patch_im2col', 'reid_cls_rows',
    'reid_attn_fwd', 'reid_attn_bwd', 'reid_cast_f32_bf16', 'reid_cast_bf16_f32', 'reid_gather_rows_f32',
    'reid_bnneck_stats', 'reid_bnneck_fwd', 'reid_bnneck_bwd_p1', 'reid_bnneck_bwd_p2',
    'reid_ce_ls_fwd', 'reid_ce_ls_bwd', 'reid_sdm_fwd', 'reid_sdm_bwd', 'reid_sdm_ws_floats',
    'reid_topk_ws_bytes', 'reid_cosine_topk', 'reid_cosine_topk_exact', 'reid_cosine_topk_exact_slots', 'reid_l2norm_rows', 'reid_sgemm', 'reid_pack_bf16_table',
    'reid_eltwise_f32', 'reid_small_attn_fwd', 'reid_small_attn_bwd', 'reid_masked_mean',
    'reid_opt_entry_bytes', 'reid_opt_ws_floats', 'reid_opt_state_floats', 'reid_opt_sumsq', 'reid_opt_clip', 'reid_opt_adamw',
    'reid_rank_metrics', 'reid_scatter_add_rows_f32', 'reid_embed_tokens',
    'reid_topk_stream_ok', 'reid_topk_scan_ok', 'reid_topk_stream_ws_bytes', 'reid_cosine_topk_stream', 'reid_merge_lora_table', 'reid_add_layernorm_fwd', 'reid_lora_bwd_fused', 'reid_lora_da_fused',
]


class GemmArgs(C.Structure):
    _fields_ = [('A', C.c_void_p), ('B', C.c_void_p), ('A2', C.c_void_p), ('B2', C.c_void_p),
                ('bias', C.c_void_p), ('R', C.c_void_p), ('aux', C.c_void_p), ('C', C.c_void_p), ('C2', C.c_void_p),
                ('img_mod', C.c_void_p),
                ('M', C.c_int32), ('N', C.c_int32), ('K', C.c_int32), ('K2', C.c_int32),
                ('lda', C.c_int32), ('ldb', C.c_int32), ('lda2', C.c_int32), ('ldb2', C.c_int32),
                ('ldr', C.c_int32), ('ldaux', C.c_int32), ('ldc', C.c_int32), ('ldc2', C.c_int32),
                ('k2_group_n', C.c_int32),
                ('act', C.c_int32), ('c_dtype', C.c_int32), ('c2_dtype', C.c_int32), ('r_dtype', C.c_int32),
                ('r_period', C.c_int32),
                ('mask_r', C.c_int32), ('mask_period', C.c_int32), ('rows_per_img', C.c_int32),
                ('c_group', C.c_int32), ('c_group_stride', C.c_int32), ('c_row_off', C.c_int32),
                ('alpha', C.c_float), ('row_scale', C.c_void_p),
                ('n_row_groups', C.c_int32), ('row_group_end', C.c_int32 * 8), ('row_group_b', C.c_int32 * 8),
                ('b_group_stride', C.c_int64)]


_libs = {}


class ReidHipError(RuntimeError):
    pass


def lib():
    """The loaded library of the current flavor; raises (never falls back) when it is absent."""
    h = _libs.get(_flavor)
    if h is None:
        path = LIB_PATHS[_flavor]
        if not os.path.exists(path):
            raise ReidHipError(f'{path} not found: build it with `python -m prcv2025reid_amd.build` '
                               '(there is no CPU or PyTorch fallback for the hot path)')
        h = C.CDLL(path)
        h.reid_last_error.restype = C.c_char_p
        missing = [n for n in EXPORTS if not hasattr(h, n)]
        if missing:
            raise ReidHipError(f'{path} lacks symbols {missing}: stale build, run `python -m prcv2025reid_amd.build --force`')
        h.reid_sdm_ws_floats.restype = C.c_int64
        h.reid_topk_ws_bytes.restype = C.c_int64
        h.reid_topk_stream_ws_bytes.restype = C.c_int64
        if h.reid_flavor() != (1 if _flavor == 'f16' else 0):
            raise ReidHipError(f'{path} was built for the other 16-bit flavor')
        _libs[_flavor] = h
    return h


def check(rc: int):
    if rc != 0:
        raise ReidHipError(f'libreid_hip: rc={rc}: {lib().reid_last_error().decode()}')


def stream_ptr() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def dt(t, allow_half: bool = False) -> int:
    if t.dtype == torch.bfloat16 or t.dtype == torch.float16:
        if t.dtype != t16():
            if allow_half and t.dtype == torch.float16:          # an IEEE-half tensor in the bf16 flavor (REID_F16: not an MFMA operand)
                return F16
            raise TypeError(f'{t.dtype} tensor passed to the {_flavor} flavor of libreid_hip')
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise TypeError(f'unsupported dtype {t.dtype}')


def _req(t, dtype=None, name='tensor'):
    if not t.is_cuda:
        raise ReidHipError(f'{name} must be a CUDA(HIP) tensor')
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f'{name}: expected {dtype}, got {t.dtype}')
    if t.dim() >= 1 and t.stride(-1) != 1:
        raise ValueError(f'{name}: last dim must be contiguous')


def ld(t) -> int:
    return t.stride(0) if t.dim() == 2 else t.shape[-1]
