"""GPU: the stochastic regularisers of the training forward -- kernels with explicit multipliers against torch, DropPath through the
vision executor against autograd through the oracle with the SAME per-sample factors, and the model-level behaviour
(eval unaffected, train deterministic per seed, masks with the right statistics)."""
import numpy as np
import pytest
import torch

from helpers import load_case, case_inputs
from oracle import reid_oracle as O

pytestmark = pytest.mark.gpu


def test_gemm_row_scale_and_ln_bwd_scale():
    from prcv2025reid_amd import ops, _lib
    _lib.set_flavor('bf16')
    g = torch.Generator(device='cuda').manual_seed(0)
    S, n_img, d = 7, 9, 128
    M = S * n_img
    A = torch.randn(M, 64, device='cuda', generator=g).bfloat16()
    W = (torch.randn(d, 64, device='cuda', generator=g) * 0.1).bfloat16()
    bias = torch.randn(d, device='cuda', generator=g)
    R = torch.randn(M, d, device='cuda', generator=g)
    sc = torch.tensor([0.0, 1.25, 1.25, 0.0, 1.25, 1.25, 1.25, 0.0, 1.25], device='cuda')
    out = torch.empty(M, d, device='cuda')
    ops.gemm(A, W, out, bias=bias, R=R, row_scale=sc, rows_per_img=S)
    want = R + sc.repeat_interleave(S).view(-1, 1) * (A.float() @ W.float().t() + bias)
    assert float((out - want).abs().max()) < 2e-3
    # layer-norm backward: the bf16 copy carries the next branch's factor, the fp32 gradient does not
    x = torch.randn(M, d, device='cuda', generator=g); dy = torch.randn(M, d, device='cuda', generator=g)
    gam = torch.rand(d, device='cuda', generator=g) + 0.5
    mean = x.mean(1); rstd = (x.var(1, unbiased=False) + 1e-5).rsqrt()
    dx = torch.empty(M, d, device='cuda'); dxb = torch.empty(M, d, device='cuda', dtype=torch.bfloat16)
    ops.layernorm_bwd(dy, x, gam, mean, rstd, dx, dx_bf16=dxb, bf16_row_scale=sc, rows_per_img=S)
    xr = x.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (d,), gam, None, 1e-5).backward(dy)
    assert float((dx - xr.grad).abs().max()) < 1e-4
    want_b = (xr.grad * sc.repeat_interleave(S).view(-1, 1))
    assert float((dxb.float() - want_b).abs().max()) < 2e-2 and float(dxb[:S].abs().max()) == 0.0


def test_small_attn_with_dropout_multipliers():
    from prcv2025reid_amd.head import SmallAttnFn
    g = torch.Generator(device='cuda').manual_seed(1)
    B, Mtok, heads, D = 6, 5, 8, 512
    qkv = torch.randn(B * Mtok, 3 * D, device='cuda', generator=g, requires_grad=True)
    km = (torch.rand(B, Mtok, device='cuda', generator=g) > 0.3); km[:, 0] = True
    drop = (torch.rand(B, heads, 8, 8, device='cuda', generator=g) > 0.3).float() / 0.7
    out = SmallAttnFn.apply(qkv, km.to(torch.uint8).contiguous(), B, Mtok, heads, drop)
    go = torch.randn_like(out)
    out.backward(go)
    q2 = qkv.detach().clone().requires_grad_(True)
    q, k, v = [t.view(B, Mtok, heads, 64).transpose(1, 2) for t in q2.view(B, Mtok, 3 * D).split(D, dim=2)]
    sc = (q @ k.transpose(-1, -2)) / 8.0
    sc = sc.masked_fill(~km.view(B, 1, 1, Mtok), float('-inf'))
    p = torch.softmax(sc, -1) * drop[:, :, :Mtok, :Mtok]
    ref = (p @ v).transpose(1, 2).reshape(B * Mtok, D)
    ref.backward(go)
    assert float((out - ref).abs().max()) < 1e-5
    assert float((qkv.grad - q2.grad).abs().max()) < 1e-4


@pytest.mark.parametrize('flavor,tol', [('bf16', 3e-2), ('f16', 5e-3)])
def test_drop_path_forward_backward_vs_oracle(flavor, tol):
    """Vision executor with explicit DropPath factors == autograd through the oracle with the same factors."""
    from prcv2025reid_amd import _lib
    from test_model_gpu import build_model
    z, meta = load_case('tiny_train_frozen')
    cfg, arch, state, batch, tokens = case_inputs(meta)
    model = build_model(meta, state, True, flavor)
    imgs = batch['images']['nir'][:5]
    g = torch.Generator().manual_seed(3)
    L = arch['vision_layers']
    scales = []
    for l in range(L):
        sa = (torch.rand(5, generator=g) > 0.3).float() / 0.7
        sm = (torch.rand(5, generator=g) > 0.3).float() / 0.7
        scales.append((sa, sm))
    scales[0] = (None, scales[0][1])                                     # a branch without DropPath
    cot = torch.randn(5, arch['fusion_dim'], generator=g)
    # oracle
    st = {k: (v.clone().requires_grad_(True) if '.loras.' in k else v) for k, v in state.items()}
    fo = O.encode_vision(imgs, 'nir', st, arch, drop_scales=scales)
    fo.backward(cot)
    # HIP
    from prcv2025reid_amd.engine import VisionEncodeFn
    _lib.set_flavor(flavor)
    model.engine.refresh()
    model.engine.pending_drop_scales = [(None if a is None else a.cuda(), None if b is None else b.cuda()) for a, b in scales]
    fh = VisionEncodeFn.apply(model.engine, (model.vision_modalities.index('nir'),), model.lora_arena, 1, imgs.cuda())
    fh.backward(cot.cuda())
    assert float((fh.detach().cpu() - fo.detach()).norm() / fo.detach().norm()) < tol
    num = den = 0.0
    for k, v in st.items():
        if '.loras.nir.' in k and v.grad is not None:
            gh = model.lora_grad_view(k).detach().cpu()
            num += float((gh - v.grad).pow(2).sum()); den += float(v.grad.pow(2).sum())
    assert den > 0 and (num / den) ** 0.5 < tol


def test_model_train_regularisers():
    from prcv2025reid_amd.model import CLIPBasedMultiModalReIDModel, apply_reference_freeze
    from helpers import case_config
    z, meta = load_case('tiny_train_frozen')
    cfg, arch, state, batch, tokens = case_inputs(meta)
    images = {m: t.cuda() for m, t in batch['images'].items()}
    masks = {m: torch.ones_like(t) for m, t in batch['modality_mask'].items()}

    def make(train):
        c = case_config(meta, device='cuda')
        c.drop_path, c.dropout_rate, c.fusion_dropout, c.sdm_dropout, c.modality_dropout = 0.3, 0.5, 0.1, 0.1, 0.5
        c.modality_dropout_warmup_epochs = 0
        m = CLIPBasedMultiModalReIDModel(c); m.set_num_classes(int(meta['num_classes'])); m.load_state_dict(state, strict=True)
        apply_reference_freeze(m); m.set_epoch(2); m.train(train)
        return m

    ev = make(False)
    with torch.no_grad():
        e1 = ev(images=images, texts=batch['texts'], modality_masks=masks)['bn_features']
        e2 = ev(images=images, texts=batch['texts'], modality_masks=masks)['bn_features']
    assert torch.equal(e1, e2)                                                    # eval: nothing random
    a, b = make(True), make(True)
    outs = []
    for m in (a, b):
        o = m(images=images, texts=batch['texts'], modality_masks=masks)
        L = m.compute_loss(o, batch['person_id'].cuda())
        L['total_loss'].backward()
        outs.append((o['logits'].detach().clone(), m.lora_arena.grad.detach().clone(), float(L['total_loss'].detach())))
    assert torch.equal(outs[0][0], outs[1][0])                                    # same seed -> same masks -> same logits
    assert abs(outs[0][2] - outs[1][2]) <= 1e-6 * abs(outs[1][2])                 # (the loss sums use fp32 atomics: last-bit freedom)
    assert np.isfinite(outs[0][2]) and float(outs[0][1].abs().sum()) > 0
    o2 = a(images=images, texts=batch['texts'], modality_masks=masks)             # next call: new masks
    assert not torch.equal(o2['logits'].detach(), outs[0][0])
    k = a._keep_mask((200000,), 0.5)
    assert abs(float(k.mean()) - 1.0) < 0.02 and set(torch.unique(k).tolist()) == {0.0, 2.0}
    sc = a.engine.drop_path_scales(4096, 0.3)
    assert sc[0] == (None, None) and abs(float(sc[-1][0].mean()) - 1.0) < 0.05
