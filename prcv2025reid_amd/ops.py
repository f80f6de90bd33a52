"""Thin, allocation-explicit Python wrappers over the C ABI (one function per entry point).

Every wrapper takes torch CUDA tensors only as device memory + stream handles; all
arithmetic happens inside libreid_hip.so.  Nothing here has a CPU path.
"""
import ctypes as C
from typing import Optional

import torch

from . import _lib as L
from ._lib import BF16, F32, GemmArgs, check, lib, ptr, stream_ptr

ACT = {'none': L.ACT_NONE, 'gelu': L.ACT_GELU, 'quick_gelu': L.ACT_QUICK_GELU, 'relu': L.ACT_RELU,
       'dgelu': L.ACT_DGELU, 'dquick_gelu': L.ACT_DQUICK_GELU, 'drelu': L.ACT_DRELU,
       'mul_aux': L.ACT_MUL_AUX, 'gelu_dsave': L.ACT_GELU_DSAVE}


# optional per-launch timing of the dominant kernel (bench.py roofline): list of (flops, bytes, start_event, end_event)
_gemm_profile = None
def gemm_profile_begin():
    global _gemm_profile
    _gemm_profile = []


def gemm_profile_end():
    global _gemm_profile
    p, _gemm_profile = _gemm_profile, None
    return p


def gemm(A, B, C_out, *, A2=None, B2=None, K2=0, k2_group_n=0, bias=None, R=None, r_period=0, aux=None,
         C2=None, act='none', img_mod=None, mask_r=0, mask_period=0, rows_per_img=0,
         c_group=0, c_group_stride=0, c_row_off=0, alpha=1.0, M=None, row_scale=None, row_groups=None):
    """C_out = epilogue(A @ B.T + A2 @ B2.T + bias); see reid_mer_gemm in include/reid_hip.h.
    ``row_groups`` = (row_ends, weight_indices): B is then a stack [n_mats, N, K]; rows [row_ends[g-1], row_ends[g]) use B[weight_indices[g]]."""
    a = GemmArgs()
    a.A, a.B, a.C = ptr(A), ptr(B), ptr(C_out)
    a.M = A.shape[0] if M is None else M
    if row_groups is not None:
        ends, widx = row_groups
        a.N, a.K = B.shape[1], B.shape[2]
        a.ldb = B.stride(1)
        a.n_row_groups = len(ends)
        for i, (e_, w_) in enumerate(zip(ends, widx)):
            a.row_group_end[i] = e_; a.row_group_b[i] = w_
        a.b_group_stride = B.stride(0)
    else:
        a.N, a.K = B.shape[0], B.shape[1]
        a.ldb = B.stride(0)
    a.lda, a.ldc = A.stride(0), C_out.stride(0)
    a.c_dtype = L.dt(C_out, allow_half=True)
    if A2 is not None:
        a.A2, a.B2 = ptr(A2), ptr(B2)
        a.K2 = K2 or B2.shape[1]
        a.lda2, a.ldb2 = A2.stride(0), B2.stride(0)
        a.k2_group_n = k2_group_n
    if bias is not None:
        a.bias = ptr(bias)
    if R is not None:
        a.R = ptr(R); a.ldr = R.stride(0); a.r_dtype = L.dt(R); a.r_period = r_period
    if aux is not None:
        a.aux = ptr(aux); a.ldaux = aux.stride(0)
    if C2 is not None:
        a.C2 = ptr(C2); a.ldc2 = C2.stride(0); a.c2_dtype = L.dt(C2)
    a.act = ACT[act]
    if mask_r:
        a.img_mod = ptr(img_mod); a.mask_r = mask_r; a.mask_period = mask_period; a.rows_per_img = rows_per_img
    a.c_group, a.c_group_stride, a.c_row_off = c_group, c_group_stride, c_row_off
    a.alpha = alpha
    if row_scale is not None:
        a.row_scale = ptr(row_scale); a.rows_per_img = rows_per_img
    if _gemm_profile is not None and a.N > 96:   # mer_gemm_kernel<128,128,2,2> (dominant); skinny LoRA projections use other tiles
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib().reid_mer_gemm(C.byref(a), stream_ptr()))
        e1.record()
        kk = a.K + a.K2
        flops = 2.0 * a.M * a.N * kk
        # compulsory bytes of the launch: both operands once, C once, plus every epilogue operand it must read or write
        # (residual R, saved pre-activation aux, second output C2) -- r01 left the last three out and overstated traffic / algorithmic
        nbytes = 2.0 * (a.M * kk + a.N * kk) + a.M * a.N * (4 if a.c_dtype == F32 else 2)
        if R is not None:
            nbytes += (r_period if r_period > 0 else a.M) * a.N * (2 if a.r_dtype == BF16 else 4)
        if aux is not None:
            nbytes += a.M * a.N * 2
        if C2 is not None:
            nbytes += a.M * a.N * (2 if a.c2_dtype == BF16 else 4)
        _gemm_profile.append((flops, nbytes, e0, e1))
        return C_out
    check(lib().reid_mer_gemm(C.byref(a), stream_ptr()))
    return C_out


def gemm_tn(X, Y, C_out, alpha=1.0, beta=0.0, M=None):
    """C_out[P,Q] = beta*C_out + alpha * X[:M].T @ Y[:M]  (fp32 out, bf16 in)."""
    M = X.shape[0] if M is None else M
    check(lib().reid_gemm_tn(ptr(X), ptr(Y), ptr(C_out), M, X.shape[1], Y.shape[1], X.stride(0), Y.stride(0),
                             C_out.stride(0), C.c_float(alpha), C.c_float(beta), stream_ptr()))
    return C_out


def layernorm_fwd(x, gamma, beta, y_bf16=None, y_f32=None, mean=None, rstd=None, row_index=None, rows=None,
                  eps=1e-5):
    rows = (row_index.shape[0] if row_index is not None else x.shape[0]) if rows is None else rows
    y = y_bf16 if y_bf16 is not None else y_f32
    check(lib().reid_layernorm_fwd(ptr(x), x.stride(0), ptr(row_index), ptr(gamma), ptr(beta), ptr(y_bf16), ptr(y_f32),
                                   y.stride(0), ptr(mean), ptr(rstd), rows, x.shape[1], C.c_float(eps), stream_ptr()))


def add_layernorm_fwd(x, y, x_out, gamma, beta, h, mean=None, rstd=None, row_scale=None, rows_per_img=0, eps=1e-5):
    """x_out = x + row_scale[row // rows_per_img] * y;  h = LayerNorm(x_out) (16-bit);  see reid_add_layernorm_fwd."""
    check(lib().reid_add_layernorm_fwd(ptr(x), x.stride(0), ptr(y), L.dt(y, allow_half=True), y.stride(0), ptr(row_scale), rows_per_img, ptr(x_out), x_out.stride(0),
                                       ptr(gamma), ptr(beta), ptr(h), h.stride(0), ptr(mean), ptr(rstd), x.shape[0], x.shape[1],
                                       C.c_float(eps), stream_ptr()))


_ln_profile = None


def ln_profile_begin():
    global _ln_profile
    _ln_profile = []


def ln_profile_end():
    global _ln_profile
    p, _ln_profile = _ln_profile, None
    return p


def layernorm_bwd(dy, x, gamma, mean, rstd, dx, dx_bf16=None, dres=None, row_index=None, dgamma=None, dbeta=None,
                  rows=None, bf16_row_scale=None, rows_per_img=0):
    rows = (row_index.shape[0] if row_index is not None else x.shape[0]) if rows is None else rows
    if dres is not None and dres.dtype != dx.dtype:
        raise ValueError('layernorm_bwd: dres and dx must have one dtype (float32 or float16)')
    if _ln_profile is not None and rows >= 4096 and row_index is None:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        cols = x.shape[1]
        # algorithmic bytes per row: dy (2 or 4) + x (4) + dres (4 or 2) + dx (4 or 2) + 16-bit copy (2)
        nbytes = rows * cols * ((2 if dy.dtype != torch.float32 else 4) + 4 + (dx.element_size() if dres is not None else 0) +
                                dx.element_size() + (2 if dx_bf16 is not None else 0))
        e0.record()
        check(lib().reid_layernorm_bwd(ptr(dy), L.dt(dy), dy.stride(0), ptr(x), x.stride(0), ptr(row_index), ptr(gamma),
                                       ptr(mean), ptr(rstd), ptr(dres), ptr(dx), L.F16 if dx.dtype == torch.float16 else L.F32, ptr(dx_bf16), dx.stride(0), ptr(dgamma),
                                       ptr(dbeta), rows, x.shape[1], ptr(bf16_row_scale), rows_per_img, stream_ptr()))
        e1.record()
        _ln_profile.append((nbytes, e0, e1))
        return
    check(lib().reid_layernorm_bwd(ptr(dy), L.dt(dy), dy.stride(0), ptr(x), x.stride(0), ptr(row_index), ptr(gamma),
                                   ptr(mean), ptr(rstd), ptr(dres), ptr(dx), L.F16 if dx.dtype == torch.float16 else L.F32, ptr(dx_bf16), dx.stride(0), ptr(dgamma),
                                   ptr(dbeta), rows, x.shape[1], ptr(bf16_row_scale), rows_per_img, stream_ptr()))


def patch_im2col(images, patches, patch, cin):
    n, _, H, W = images.shape
    check(lib().reid_patch_im2col(ptr(images), ptr(patches), n, H, W, patch, cin, stream_ptr()))


def cls_rows(cls, pos0, x, n_img, tokens):
    check(lib().reid_cls_rows(ptr(cls), ptr(pos0), ptr(x), x.stride(0), n_img, tokens, x.shape[1], stream_ptr()))


def attn_fwd(qkv, out, lse, n_seq, S, heads, causal=False, key_mask=None, q_tiles=0):
    check(lib().reid_attn_fwd(ptr(qkv), qkv.stride(0), ptr(key_mask), ptr(out), out.stride(0), ptr(lse), n_seq, S, heads,
                              int(causal), int(q_tiles), stream_ptr()))


def attn_bwd(qkv, out, dout, lse, dqkv, delta_ws, n_seq, S, heads, causal=False, key_mask=None, q_tiles=0):
    check(lib().reid_attn_bwd(ptr(qkv), qkv.stride(0), ptr(key_mask), ptr(out), ptr(dout), out.stride(0), ptr(lse),
                              ptr(dqkv), dqkv.stride(0), ptr(delta_ws), n_seq, S, heads, int(causal), int(q_tiles), stream_ptr()))


def cast_f32_bf16(src, dst):
    check(lib().reid_cast_f32_bf16(ptr(src), ptr(dst), C.c_int64(src.numel()), stream_ptr()))
    return dst


def cast_bf16_f32(src, dst):
    check(lib().reid_cast_bf16_f32(ptr(src), ptr(dst), C.c_int64(src.numel()), stream_ptr()))
    return dst


def lora_bwd_fused(dY, T, BT, U, dB, img_mod, rows_per_img, mask_r, scale, u_partial=None):
    """U = mask(dY . B) * scale and dB += dY^T . T from one pass over dY (Rp = 32; N = 768, or a multiple of 768 as column blocks with
    the fp32 scratch ``u_partial`` [M, 32]); see reid_lora_bwd_fused."""
    N = dY.shape[1]
    nb = N // 768
    if nb > 1 and u_partial is None:
        raise ValueError('lora_bwd_fused: a cotangent wider than 768 columns needs u_partial')
    for q in range(nb):
        mode = 0 if nb == 1 else ((1 if q > 0 else 0) | (2 if q + 1 < nb else 0))
        dYq, BTq, dBq = dY[:, q * 768:(q + 1) * 768], BT[:, q * 768:(q + 1) * 768], dB[q * 768:(q + 1) * 768]
        check(lib().reid_lora_bwd_fused(ptr(dYq), dYq.stride(0), ptr(T), T.stride(0), ptr(BTq), BTq.stride(0), ptr(U), U.stride(0),
                                        ptr(dBq), dBq.stride(0), ptr(img_mod), rows_per_img, mask_r, dY.shape[0], 768, T.shape[1],
                                        C.c_float(scale), ptr(u_partial), mode, stream_ptr()))


def lora_da_fused(X, U, dA, img_mod, rows_per_img, mask_r, n_groups=1):
    """dA += U^T . X for the adapter groups of one MERLinear, one pass over X (one image per workgroup); see reid_lora_da_fused."""
    check(lib().reid_lora_da_fused(ptr(X), X.stride(0), ptr(U), U.stride(0), ptr(dA), dA.stride(0), ptr(img_mod), rows_per_img, mask_r,
                                   X.shape[0], X.shape[1], U.shape[1] // n_groups, n_groups, stream_ptr()))


def lora_da_fused_ok(K, Rp, rows_per_img, mask_r, n_groups):
    return K % 768 == 0 and Rp == 32 and rows_per_img >= 32 and mask_r <= 16 and 16 % mask_r == 0 and n_groups in (1, 3)


def lora_bwd_fused_ok(N, Rp):
    return N % 768 == 0 and N // 768 in (1, 2, 3, 4) and Rp == 32


def merge_lora_table(table, n_entries, max_tiles, arena, weff, Rp, r, nmod, scaling):
    check(lib().reid_merge_lora_table(ptr(table), n_entries, max_tiles, ptr(arena), ptr(weff), Rp, r, nmod, C.c_float(scaling), stream_ptr()))


def to_bf16(src: torch.Tensor) -> torch.Tensor:
    """New bf16 tensor with the values of fp32 ``src`` (round-to-nearest-even, HIP kernel)."""
    src = src.contiguous()
    dst = torch.empty(src.shape, dtype=L.t16(), device=src.device)
    if src.numel():
        cast_f32_bf16(src, dst)
    return dst


def pack_bf16_table(src_arena, dst_arena, table, n_entries):
    check(lib().reid_pack_bf16_table(ptr(src_arena), ptr(dst_arena), ptr(table), n_entries, stream_ptr()))


to_t16 = to_bf16


def gather_rows(src, index, dst):
    check(lib().reid_gather_rows_f32(ptr(src), src.stride(0), ptr(index), ptr(dst), dst.stride(0), index.shape[0],
                                     src.shape[1], stream_ptr()))
    return dst


def l2norm_rows(x, y=None, y_bf16=None, eps=1e-12, scale=1.0):
    o = y if y is not None else y_bf16
    check(lib().reid_l2norm_rows(ptr(x), x.stride(0), ptr(y), ptr(y_bf16), o.stride(0), x.shape[0], x.shape[1],
                                 C.c_float(eps), C.c_float(scale), stream_ptr()))


def sgemm(A, B, C_out, *, ta=False, tb=False, alpha=1.0, beta=0.0, bias=None, act='none'):
    """fp32 C = act(alpha * op(A) @ op(B) + bias) + beta*C on the vector ALU (small head GEMMs, exact fp32)."""
    M = A.shape[1] if ta else A.shape[0]
    K = A.shape[0] if ta else A.shape[1]
    N = B.shape[0] if tb else B.shape[1]
    sam, sak = (1, A.stride(0)) if ta else (A.stride(0), 1)
    sbk, sbn = (1, B.stride(0)) if tb else (B.stride(0), 1)
    check(lib().reid_sgemm(ptr(A), ptr(B), ptr(C_out), M, N, K, C.c_int64(sam), C.c_int64(sak), C.c_int64(sbk),
                           C.c_int64(sbn), C_out.stride(0),
                           C.c_float(alpha), C.c_float(beta), ptr(bias), ACT[act], stream_ptr()))
    return C_out


# ----------------------------------------------------------------------------------------- head
def bnneck_stats(x, sum_, sqsum):
    check(lib().reid_bnneck_stats(ptr(x), x.stride(0), x.shape[0], x.shape[1], ptr(sum_), ptr(sqsum), stream_ptr()))


def bnneck_fwd(x, gamma, beta, running_mean, running_var, sum_, sqsum, count, training, y, y_bf16, mean, invstd, rnorm,
               eps=1e-5, momentum=0.1, scale=8.0):
    check(lib().reid_bnneck_fwd(ptr(x), x.stride(0), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), ptr(sum_),
                                ptr(sqsum), C.c_float(count), int(training), ptr(y), ptr(y_bf16), y.stride(0), ptr(mean),
                                ptr(invstd), ptr(rnorm), x.shape[0], x.shape[1], C.c_float(eps), C.c_float(momentum),
                                C.c_float(scale), stream_ptr()))


def bnneck_bwd_p1(dy, x, gamma, beta, mean, invstd, rnorm, dz, sum_dz, sum_dz_xhat, scale=8.0):
    check(lib().reid_bnneck_bwd_p1(ptr(dy), dy.stride(0), ptr(x), x.stride(0), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd),
                                   ptr(rnorm), ptr(dz), ptr(sum_dz), ptr(sum_dz_xhat), x.shape[0], x.shape[1],
                                   C.c_float(scale), stream_ptr()))


def bnneck_bwd_p2(dz, x, gamma, mean, invstd, sum_dz, sum_dz_xhat, count, training, dx):
    check(lib().reid_bnneck_bwd_p2(ptr(dz), ptr(x), x.stride(0), ptr(gamma), ptr(mean), ptr(invstd), ptr(sum_dz),
                                   ptr(sum_dz_xhat), C.c_float(count), int(training), ptr(dx), dx.stride(0), x.shape[0],
                                   x.shape[1], stream_ptr()))


def ce_ls_fwd(logits, labels, valid, row_loss, loss_sum, smoothing=0.1):
    check(lib().reid_ce_ls_fwd(ptr(logits), logits.stride(0), ptr(labels), ptr(valid), logits.shape[0], logits.shape[1],
                               C.c_float(smoothing), ptr(row_loss), ptr(loss_sum), stream_ptr()))


def ce_ls_bwd(logits, labels, valid, grad_scale, dlogits, smoothing=0.1):
    check(lib().reid_ce_ls_bwd(ptr(logits), logits.stride(0), ptr(labels), ptr(valid), logits.shape[0], logits.shape[1],
                               C.c_float(smoothing), ptr(grad_scale), ptr(dlogits), dlogits.stride(0), stream_ptr()))


def sdm_ws_floats(P, N, Mg, D):
    return int(lib().reid_sdm_ws_floats(P, N, Mg, D))


def sdm_fwd(q, g, q_label, g_label, q_valid, g_valid, tau, ws, result, P=1):
    """q [P*N, D] (P query sides stacked), g [Mg, D]; result f32 [2*P] = (loss, contributes) per pair."""
    N = q.shape[0] // P
    check(lib().reid_sdm_fwd(ptr(q), q.stride(0), ptr(g), g.stride(0), ptr(q_label), ptr(g_label), ptr(q_valid), ptr(g_valid),
                             P, N, g.shape[0], q.shape[1], C.c_float(tau), ptr(ws), ptr(result), stream_ptr()))


def sdm_bwd(q, g, q_label, g_label, q_valid, g_valid, tau, ws, gscale, dq, dg, P=1):
    N = q.shape[0] // P
    check(lib().reid_sdm_bwd(ptr(q), q.stride(0), ptr(g), g.stride(0), ptr(q_label), ptr(g_label), ptr(q_valid), ptr(g_valid),
                             P, N, g.shape[0], q.shape[1], C.c_float(tau), ptr(ws), ptr(gscale), ptr(dq), dq.stride(0),
                             ptr(dg), dg.stride(0), stream_ptr()))


# ----------------------------------------------------------------------------------------- retrieval
def topk_ws_bytes(Nq, Ng, k):
    return int(lib().reid_topk_ws_bytes(Nq, Ng, k))


def cosine_topk(Qb, Gb, Qf, Gf, k, ws, out_idx, out_score, exclude_q=None, exclude_g=None):
    check(lib().reid_cosine_topk(ptr(Qb), ptr(Gb), ptr(Qf), ptr(Gf), Qf.shape[0], Gf.shape[0], Qf.shape[1], k,
                                 ptr(exclude_q), ptr(exclude_g), ptr(ws), ptr(out_idx), ptr(out_score), stream_ptr()))


def topk_stream_ok(Nq, Ng, D, k) -> bool:
    return bool(lib().reid_topk_stream_ok(Nq, Ng, D, k))


def topk_scan_ok(Nq, Ng, D, k):
    return bool(lib().reid_topk_scan_ok(Nq, Ng, D, k))


def topk_stream_ws_bytes(k):
    return int(lib().reid_topk_stream_ws_bytes(k))


def cosine_topk_stream(Qf, Gf, k, ws, out_idx, out_score, exclude_q=None, exclude_g=None):
    check(lib().reid_cosine_topk_stream(ptr(Qf), ptr(Gf), Qf.shape[0], Gf.shape[0], Qf.shape[1], k, ptr(exclude_q), ptr(exclude_g),
                                        ptr(ws), ptr(out_idx), ptr(out_score), stream_ptr()))


def cosine_topk_exact(Qf, Gf, k, scratch, out_idx, out_score, exclude_q=None, exclude_g=None):
    check(lib().reid_cosine_topk_exact(ptr(Qf), ptr(Gf), Qf.shape[0], Gf.shape[0], Qf.shape[1], k, ptr(exclude_q),
                                       ptr(exclude_g), ptr(scratch), ptr(out_idx), ptr(out_score), stream_ptr()))


def cosine_topk_exact_slots(Qf, Gf, k, n_slots, slots, scratch, out_idx, out_score, exclude_q=None, exclude_g=None):
    check(lib().reid_cosine_topk_exact_slots(ptr(Qf), ptr(Gf), Qf.shape[0], Gf.shape[0], Qf.shape[1], k, ptr(exclude_q), ptr(exclude_g),
                                             n_slots, ptr(slots), ptr(scratch), ptr(out_idx), ptr(out_score), stream_ptr()))


# ----------------------------------------------------------------------------------------- small fp32 head pieces
ELT = {'add': 0, 'relu': 1, 'relu_bwd': 2, 'gelu': 3, 'gelu_bwd': 4, 'mul': 5, 'nan_to_num': 6, 'keep_mask': 7}


def eltwise(op, x, y=None, out=None, alpha=1.0):
    out = torch.empty_like(x) if out is None else out
    check(lib().reid_eltwise_f32(ELT[op], ptr(x), ptr(y), ptr(out), C.c_int64(x.numel()), C.c_float(alpha), stream_ptr()))
    return out


def small_attn_fwd(qkv, key_mask, out, probs, n_seq, S, heads, drop=None):
    check(lib().reid_small_attn_fwd(ptr(qkv), qkv.stride(0), ptr(key_mask), ptr(drop), ptr(out), out.stride(0), ptr(probs), n_seq, S,
                                    heads, stream_ptr()))


def small_attn_bwd(qkv, probs, dout, dqkv, n_seq, S, heads, drop=None):
    check(lib().reid_small_attn_bwd(ptr(qkv), qkv.stride(0), ptr(probs), ptr(drop), ptr(dout), dout.stride(0), ptr(dqkv),
                                    dqkv.stride(0), n_seq, S, heads, stream_ptr()))


def masked_mean(x, mask, out, B, M, D, backward=False):
    check(lib().reid_masked_mean(ptr(x), ptr(mask), ptr(out), B, M, D, int(backward), stream_ptr()))
    return out


def rank_metrics(scores, g_pid, g_img, q_pid, q_slot, q_excl, csr_off, csr_idx, Ng, max_pos, ap, rank1, npos):
    """Per-query AP / first-positive rank / #positives from fp32 score rows (reid_rank_metrics, include/reid_hip.h)."""
    check(lib().reid_rank_metrics(ptr(scores), C.c_int64(scores.stride(0)), ptr(g_pid), ptr(g_img), ptr(q_pid), ptr(q_slot),
                                  ptr(q_excl), ptr(csr_off), ptr(csr_idx), scores.shape[0], Ng, max_pos, ptr(ap), ptr(rank1), ptr(npos),
                                  stream_ptr()))


def scatter_add_rows(src, index, out):
    """out[index[r]] += src[r] (f32 rows; int32 index)."""
    check(lib().reid_scatter_add_rows_f32(ptr(src), src.stride(0), ptr(index), ptr(out), out.stride(0), src.shape[0], src.shape[1],
                                          out.shape[0], stream_ptr()))


def embed_tokens(tok, pos, ids, out):
    """out[b*T + t] = tok[ids[b, t]] + pos[t]  (text-tower input, f32)."""
    B, T = ids.shape
    check(lib().reid_embed_tokens(ptr(tok), ptr(pos), ptr(ids), ptr(out), B, T, tok.shape[1], tok.shape[0], stream_ptr()))
    return out
