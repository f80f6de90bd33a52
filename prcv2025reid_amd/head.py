"""Autograd boundaries of the head kernels: BN-neck, classifier, label-smoothed CE, SDM loss.

Reference: BNNeck.forward models/model.py:208-224; compute_loss :512-659; sdm_loss_stable
models/sdm_loss.py:13-149.  All arithmetic is in libreid_hip.so (fp32); torch only carries the tensors.
"""
import torch

from . import ops


class BNNeckFn(torch.autograd.Function):
    """y = 8 * normalize(BatchNorm1d(x)); batch statistics in training (running stats updated in place)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training: bool, momentum: float, eps: float):
        x = x.contiguous().float()
        B, D = x.shape
        dev = x.device
        s1 = torch.empty(D, device=dev); s2 = torch.empty(D, device=dev)
        y = torch.empty(B, D, device=dev)
        mean = torch.empty(D, device=dev); invstd = torch.empty(D, device=dev); rn = torch.empty(B, device=dev)
        if training:
            ops.bnneck_stats(x, s1, s2)
        ops.bnneck_fwd(x, gamma.detach(), beta.detach(), running_mean, running_var, s1, s2, float(B), training, y, None, mean,
                       invstd, rn, eps=eps, momentum=momentum, scale=8.0)
        ctx.save_for_backward(x, gamma.detach(), beta.detach(), mean, invstd, rn)
        ctx.training = training
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, invstd, rn = ctx.saved_tensors
        B, D = x.shape
        dev = x.device
        dy = dy.contiguous().float()
        dz = torch.empty(B, D, device=dev); a = torch.empty(D, device=dev); b = torch.empty(D, device=dev)
        dx = torch.empty(B, D, device=dev)
        ops.bnneck_bwd_p1(dy, x, gamma, beta, mean, invstd, rn, dz, a, b, scale=8.0)
        ops.bnneck_bwd_p2(dz, x, gamma, mean, invstd, a, b, float(B), ctx.training, dx)
        return dx, b, a, None, None, None, None, None


class LinearF32Fn(torch.autograd.Function):
    """y = x @ W.T (+ bias) in exact fp32 (vector-ALU GEMM); used for the identity classifier."""

    @staticmethod
    def forward(ctx, x, W, bias):
        x = x.contiguous().float()
        y = torch.empty(x.shape[0], W.shape[0], device=x.device)
        ops.sgemm(x, W.detach(), y, tb=True, bias=None if bias is None else bias.detach())
        ctx.save_for_backward(x, W.detach())
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = dy.contiguous().float()
        dx = dW = db = None                                   # only what autograd asks for (frozen weights cost nothing)
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            ops.sgemm(dy, W, dx)                              # [B,C] @ [C,D]
        if ctx.needs_input_grad[1]:
            dW = torch.empty_like(W)
            ops.sgemm(dy, x, dW, ta=True)                     # [C,B] @ [B,D]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(0)
        return dx, dW, db


class CrossEntropyLSFn(torch.autograd.Function):
    """Mean over valid rows of label-smoothed CE; returns (loss, count) with count a float tensor."""

    @staticmethod
    def forward(ctx, logits, labels, valid, smoothing: float):
        logits = logits.contiguous().float()
        acc = torch.zeros(2, device=logits.device)
        ops.ce_ls_fwd(logits, labels, valid, None, acc, smoothing)
        cnt = acc[1]
        loss = acc[0] / cnt.clamp_min(1.0)
        ctx.save_for_backward(logits, labels, valid, cnt)
        ctx.smoothing = smoothing
        ctx.mark_non_differentiable(cnt)
        return loss, cnt

    @staticmethod
    def backward(ctx, dloss, _dcnt):
        logits, labels, valid, cnt = ctx.saved_tensors
        gs = (dloss / cnt.clamp_min(1.0)).reshape(1).float().contiguous()
        dl = torch.empty_like(logits)
        ops.ce_ls_bwd(logits, labels, valid, gs, dl, ctx.smoothing)
        return dl, None, None, None


class SDMFn(torch.autograd.Function):
    """(losses [P], contributes [P]) of sdm_loss_stable between P stacked modality feature sets q [P, N, D] and the vis features
    g [Mg, D] -- the fused kernel of csrc/sdm.hip: all pairs in one launch, no [N, Mg] matrix in memory."""

    @staticmethod
    def forward(ctx, q, g, q_label, g_label, q_valid, g_valid, tau: float):
        P, N, D = q.shape
        q2 = q.reshape(P * N, D).contiguous().float(); g = g.contiguous().float()
        ws = torch.empty(ops.sdm_ws_floats(P, N, g.shape[0], D), device=q.device)
        res = torch.zeros(2 * P, device=q.device)
        qv = None if q_valid is None else q_valid.reshape(P * N).contiguous()
        ops.sdm_fwd(q2, g, q_label, g_label, qv, g_valid, tau, ws, res, P=P)
        ctx.save_for_backward(q2, g, q_label, g_label, qv, g_valid, ws)
        ctx.tau, ctx.P = tau, P
        res = res.view(P, 2)
        loss, flag = res[:, 0].contiguous(), res[:, 1].contiguous()
        ctx.mark_non_differentiable(flag)
        return loss, flag

    @staticmethod
    def backward(ctx, dloss, _dflag):
        q2, g, ql, gl, qv, gv, ws = ctx.saved_tensors
        P = ctx.P
        dq = torch.zeros_like(q2); dg = torch.zeros_like(g)
        ops.sdm_bwd(q2, g, ql, gl, qv, gv, ctx.tau, ws, dloss.reshape(P).float().contiguous(), dq, dg, P=P)
        return dq.view(P, q2.shape[0] // P, q2.shape[1]), dg, None, None, None, None, None


# ----------------------------------------------------------------------------------------------------------------
# Small fp32 pieces of SemanticDisentanglementModule (models/model.py:57-77) and FeatureFusion (:113-183)
class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.eltwise('add', a.contiguous().float(), b.contiguous().float())

    @staticmethod
    def backward(ctx, g):
        return g, g


class MulFn(torch.autograd.Function):
    """y = x * m with a constant multiplier tensor m (dropout masks 0 or 1/keep): nn.Dropout / DropPath-style scaling."""

    @staticmethod
    def forward(ctx, x, m):
        m = m.contiguous().float()
        ctx.save_for_backward(m)
        return ops.eltwise('mul', x.contiguous().float(), m)

    @staticmethod
    def backward(ctx, g):
        (m,) = ctx.saved_tensors
        return ops.eltwise('mul', g.contiguous().float(), m), None


class ActFn(torch.autograd.Function):
    """relu / gelu (exact erf form) with the matching derivative kernels."""

    @staticmethod
    def forward(ctx, x, kind: str):
        x = x.contiguous().float()
        ctx.save_for_backward(x)
        ctx.kind = kind
        return ops.eltwise(kind, x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.eltwise(ctx.kind + '_bwd', x, g.contiguous().float()), None


class NanToNumFn(torch.autograd.Function):
    """torch.nan_to_num(x, nan=0, posinf=1e4, neginf=-1e4) (model.py:165); the gradient passes where x is finite."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous().float()
        ctx.save_for_backward(x)
        return ops.eltwise('nan_to_num', x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.isfinite(x)


class LayerNormF32Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps: float):
        shp = x.shape
        x2 = x.contiguous().float().view(-1, shp[-1])
        y = torch.empty_like(x2)
        mean = torch.empty(x2.shape[0], device=x2.device); rstd = torch.empty_like(mean)
        ops.layernorm_fwd(x2, w.detach(), b.detach(), y_f32=y, mean=mean, rstd=rstd, eps=eps)
        ctx.save_for_backward(x2, w.detach(), mean, rstd)
        ctx.shp = shp
        return y.view(shp)

    @staticmethod
    def backward(ctx, g):
        x2, w, mean, rstd = ctx.saved_tensors
        g2 = g.contiguous().float().view(x2.shape)
        dx = torch.empty_like(x2)
        want = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dw = torch.zeros_like(w) if want else None; db = torch.zeros_like(w) if want else None
        ops.layernorm_bwd(g2, x2, w, mean, rstd, dx, dgamma=dw, dbeta=db)
        return dx.view(ctx.shp), dw, db, None


class LinearNdF32Fn(torch.autograd.Function):
    """y = x @ W.T + b for x of any leading shape, exact fp32 (vector-ALU GEMM)."""

    @staticmethod
    def forward(ctx, x, W, b):
        shp = x.shape
        x2 = x.contiguous().float().view(-1, shp[-1])
        W = W.detach().contiguous()
        y = torch.empty(x2.shape[0], W.shape[0], device=x2.device)
        ops.sgemm(x2, W, y, tb=True, bias=None if b is None else b.detach().contiguous())
        ctx.save_for_backward(x2, W)
        ctx.shp = shp; ctx.has_bias = b is not None
        return y.view(shp[:-1] + (W.shape[0],))

    @staticmethod
    def backward(ctx, g):
        x2, W = ctx.saved_tensors
        g2 = g.contiguous().float().view(x2.shape[0], W.shape[0])
        dx = dW = None                                        # only what autograd asks for: the SDM module and the fusion block
        if ctx.needs_input_grad[0]:                           # are frozen under the reference's default rule (train.py:1418-1425)
            dx = torch.empty_like(x2)
            ops.sgemm(g2, W, dx)
        if ctx.needs_input_grad[1]:
            dW = torch.empty_like(W)
            ops.sgemm(g2, x2, dW, ta=True)
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            ones = torch.ones(1, g2.shape[0], device=g2.device)
            db = torch.empty(1, W.shape[0], device=g2.device)
            ops.sgemm(ones, g2, db)
            db = db.view(-1)
        return (None if dx is None else dx.view(ctx.shp)), dW, db


class SmallAttnFn(torch.autograd.Function):
    """softmax(q k^T / 8 + key padding) v over S <= 8 tokens (nn.MultiheadAttention core, head_dim 64), fp32."""

    @staticmethod
    def forward(ctx, qkv, key_mask, n_seq: int, S: int, heads: int, drop=None):
        """``drop``: optional f32 [n_seq, heads, 8, 8] attention-dropout multipliers (0 or 1/keep)."""
        qkv = qkv.contiguous().float()
        d = heads * 64
        out = torch.empty(n_seq * S, d, device=qkv.device)
        probs = torch.zeros(n_seq * heads * 64, device=qkv.device)
        drop = None if drop is None else drop.contiguous().float()
        ops.small_attn_fwd(qkv, key_mask, out, probs, n_seq, S, heads, drop=drop)
        ctx.save_for_backward(qkv, probs, drop)
        ctx.dims = (n_seq, S, heads)
        return out

    @staticmethod
    def backward(ctx, g):
        qkv, probs, drop = ctx.saved_tensors
        n_seq, S, heads = ctx.dims
        dqkv = torch.empty_like(qkv)
        ops.small_attn_bwd(qkv, probs, g.contiguous().float(), dqkv, n_seq, S, heads, drop=drop)
        return dqkv, None, None, None, None, None


class MaskedMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask):
        B, M, D = x.shape
        x = x.contiguous().float(); mask = mask.contiguous().float()
        out = torch.empty(B, D, device=x.device)
        ops.masked_mean(x, mask, out, B, M, D)
        ctx.save_for_backward(mask)
        ctx.dims = (B, M, D)
        return out

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        B, M, D = ctx.dims
        dx = torch.empty(B, M, D, device=g.device)
        ops.masked_mean(g.contiguous().float(), mask, dx, B, M, D, backward=True)
        return dx, None
