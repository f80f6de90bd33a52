"""GPU: every HIP kernel against a plain PyTorch fp32 reference of the same op (through the C ABI)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', params=['bf16', 'f16'])
def ops(request):
    """Every kernel test runs once per build flavor (libreid_hip.so = bf16 operands, libreid_hip_f16.so = f16)."""
    from prcv2025reid_amd import ops as o, _lib
    _lib.set_flavor(request.param)
    _lib.check(_lib.lib().reid_check_device(0))
    yield o
    _lib.set_flavor('bf16')


def T16():
    from prcv2025reid_amd import _lib
    return _lib.t16()


def bf(x):
    return x.to(T16())


def rel_err(a, b):
    a = a.double(); b = b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (300, 768, 768), (1000, 2304, 768), (197 * 3, 3072, 768),
                                   (257, 768, 3072), (64, 512, 768), (70, 400, 512), (513, 32, 768), (513, 96, 768),
                                   (130, 64, 3072)])
def test_gemm_plain(ops, M, N, K):
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    A = bf(torch.randn(M, K, device='cuda', generator=g)); B = bf(torch.randn(N, K, device='cuda', generator=g) * 0.05)
    bias = torch.randn(N, device='cuda', generator=g)
    Cb = torch.empty(M, N, device='cuda', dtype=T16())
    Cf = torch.empty(M, N, device='cuda', dtype=torch.float32)
    ops.gemm(A, B, Cf, bias=bias)
    ops.gemm(A, B, Cb, bias=bias)
    ref = A.float() @ B.float().t() + bias
    assert rel_err(Cf, ref) < 2e-5
    assert rel_err(Cb.float(), ref) < 1e-2


def test_gemm_a_equals_identity_asymmetric_b(ops):
    # layout check of cdna_hip_programming.md section 3: A = I with an asymmetric B catches a transposed C write
    K = 128
    A = bf(torch.eye(K, device='cuda'))
    B = bf((torch.arange(256 * K, device='cuda').reshape(256, K) % 251).float())
    Cf = torch.empty(K, 256, device='cuda')
    ops.gemm(A, B, Cf)
    assert torch.equal(Cf, B.float().t())


@pytest.mark.parametrize('r,G', [(8, 1), (4, 1), (16, 1), (8, 3)])
def test_gemm_lora_extension(ops, r, G):
    g = torch.Generator(device='cuda').manual_seed(r * 10 + G)
    n_img, S, K, nmod = 9, 197, 768, 4
    M = n_img * S
    N = 768 * G
    Rp = ((nmod * r + 31) // 32) * 32
    img_mod = torch.randint(0, nmod, (n_img,), device='cuda', generator=g, dtype=torch.int32)
    x = bf(torch.randn(M, K, device='cuda', generator=g))
    W = bf(torch.randn(N, K, device='cuda', generator=g) * 0.03)
    bias = torch.randn(N, device='cuda', generator=g) * 0.1
    lora_A = torch.randn(G, nmod, r, K, device='cuda', generator=g) * 0.05      # per projection, per modality
    lora_B = torch.randn(G, nmod, 768, r, device='cuda', generator=g) * 0.1
    scaling = 1.0 / r
    # packed operands
    Acat = torch.zeros(G * Rp, K, device='cuda')
    B2 = torch.zeros(N, Rp, device='cuda')
    for gi in range(G):
        for m in range(nmod):
            Acat[gi * Rp + m * r: gi * Rp + (m + 1) * r] = lora_A[gi, m]
            B2[gi * 768:(gi + 1) * 768, m * r:(m + 1) * r] = lora_B[gi, m] * scaling
    Acat_b, B2_b = bf(Acat), bf(B2)
    T = torch.empty(M, G * Rp, device='cuda', dtype=T16())
    ops.gemm(x, Acat_b, T, img_mod=img_mod, mask_r=r, mask_period=Rp, rows_per_img=S)
    out = torch.empty(M, N, device='cuda', dtype=torch.float32)
    ops.gemm(x, W, out, A2=T, B2=B2_b, K2=Rp, k2_group_n=768 if G > 1 else 0, bias=bias)
    # reference: per-row modality routing (mer_lora.py:96) in fp32 on the bf16-rounded operands
    row_mod = img_mod.long().repeat_interleave(S)
    xf = x.float()
    ref = xf @ W.float().t() + bias
    for gi in range(G):
        for m in range(nmod):
            rows = (row_mod == m).nonzero().flatten()
            t = bf(xf[rows] @ bf(lora_A[gi, m]).float().t()).float()
            ref[rows, gi * 768:(gi + 1) * 768] += t @ bf(lora_B[gi, m] * scaling).float().t()
    # T is rounded to bf16: a rounding flip of one T element (different fp32 summation order) moves an output by ~1e-4
    assert rel_err(out, ref) < 5e-4
    # T itself: masked columns are exactly zero
    Tm = T.float().view(n_img, S, G, Rp)
    for i in range(n_img):
        m = int(img_mod[i])
        keep = torch.zeros(Rp, dtype=torch.bool, device='cuda'); keep[m * r:(m + 1) * r] = True
        assert float(Tm[i][:, :, ~keep].abs().max()) == 0.0


def test_gemm_epilogues(ops):
    g = torch.Generator(device='cuda').manual_seed(5)
    M, N, K = 393, 768, 256
    A = bf(torch.randn(M, K, device='cuda', generator=g)); B = bf(torch.randn(N, K, device='cuda', generator=g) * 0.1)
    bias = torch.randn(N, device='cuda', generator=g)
    R = torch.randn(M, N, device='cuda', generator=g)
    base = A.float() @ B.float().t() + bias
    out = torch.empty(M, N, device='cuda'); pre = torch.empty(M, N, device='cuda', dtype=T16())
    ops.gemm(A, B, out, bias=bias, R=R)
    assert rel_err(out, base + R) < 2e-5
    ops.gemm(A, B, out, bias=bias, act='gelu', C2=pre)
    assert rel_err(out, torch.nn.functional.gelu(base)) < 2e-5
    assert rel_err(pre.float(), base) < 1e-2
    # value + saved derivative (the training path's fc1): C2 = gelu'(acc); its backward partner multiplies by the saved factor
    dsv = torch.empty(M, N, device='cuda', dtype=T16())
    ops.gemm(A, B, out, bias=bias, act='gelu_dsave', C2=dsv)
    bb = base.clone().requires_grad_(True)
    torch.nn.functional.gelu(bb).sum().backward()
    assert rel_err(out, torch.nn.functional.gelu(base)) < 2e-5
    assert rel_err(dsv.float(), bb.grad) < 1e-2
    ops.gemm(A, B, out, act='mul_aux', aux=dsv)
    assert rel_err(out, (A.float() @ B.float().t()) * dsv.float()) < 2e-5
    ops.gemm(A, B, out, bias=bias, act='quick_gelu')
    assert rel_err(out, base * torch.sigmoid(1.702 * base)) < 2e-5
    ops.gemm(A, B, out, bias=bias, act='relu')
    assert rel_err(out, torch.relu(base)) < 2e-5
    # derivative forms: out = acc * act'(aux)
    u = bf(torch.randn(M, N, device='cuda', generator=g))
    uf = u.float().requires_grad_(True)
    torch.nn.functional.gelu(uf).sum().backward()
    ops.gemm(A, B, out, act='dgelu', aux=u)
    assert rel_err(out, (A.float() @ B.float().t()) * uf.grad) < 2e-5
    uf.grad = None
    (uf * torch.sigmoid(1.702 * uf)).sum().backward()
    ops.gemm(A, B, out, act='dquick_gelu', aux=u)
    assert rel_err(out, (A.float() @ B.float().t()) * uf.grad) < 2e-5
    # periodic residual + grouped output rows (patch rows -> token rows behind the CLS slot)
    n_img, S = 3, 131
    Mp = n_img * (S - 1)
    A = bf(torch.randn(Mp, K, device='cuda', generator=g))
    pos = torch.randn(S - 1, N, device='cuda', generator=g)
    x = torch.zeros(n_img * S, N, device='cuda')
    ops.gemm(A, B, x, bias=bias, R=pos, r_period=S - 1, c_group=S - 1, c_group_stride=S, c_row_off=1)
    ref = (A.float() @ B.float().t() + bias).view(n_img, S - 1, N) + pos
    xr = x.view(n_img, S, N)
    assert rel_err(xr[:, 1:], ref) < 2e-5 and float(xr[:, 0].abs().max()) == 0.0


@pytest.mark.parametrize('tile', [0, 12, 14, 3])
def test_gemm_large_tile_paths(ops, tile):
    """The 256 x 256 and 224 x 256 ping-pong tiles (12, 14), the 128 x 128 tile (3) and the
    default dispatch (0) on the training-size shapes they serve, with every lean epilogue and the LoRA K extension; ragged M."""
    from prcv2025reid_amd import _lib
    g = torch.Generator(device='cuda').manual_seed(77 + tile)
    M, K, r, nmod, S = 64 * 197 - 8, 768, 8, 4, 197                     # 12600 rows: the last tile row is ragged
    Rp = 32
    x = bf(torch.randn(M, K, device='cuda', generator=g))
    xf = x.float()
    _lib.check(_lib.lib().reid_set_knob(b'GEMM_TILE', tile if tile else -1))
    try:
        # qkv-like: 16-bit output, LoRA extension with three projection groups
        N = 2304
        W = bf(torch.randn(N, K, device='cuda', generator=g) * 0.03); bias = torch.randn(N, device='cuda', generator=g) * 0.1
        T = bf(torch.randn(M, 3 * Rp, device='cuda', generator=g) * 0.2); B2 = bf(torch.randn(N, Rp, device='cuda', generator=g) * 0.1)
        out16 = torch.empty(M, N, device='cuda', dtype=T16())
        ops.gemm(x, W, out16, A2=T, B2=B2, K2=Rp, k2_group_n=768, bias=bias)
        ref = xf @ W.float().t() + bias
        for gi in range(3):
            ref[:, gi * 768:(gi + 1) * 768] += T.float()[:, gi * Rp:(gi + 1) * Rp] @ B2.float()[gi * 768:(gi + 1) * 768].t()
        assert rel_err(out16.float(), ref) < 1e-2
        out32 = torch.empty(M, N, device='cuda')
        ops.gemm(x, W, out32, A2=T, B2=B2, K2=Rp, k2_group_n=768, bias=bias)
        assert rel_err(out32, ref) < 2e-5
        # out-projection-like: fp32 residual stream, per-row scale (DropPath), LoRA extension
        N = 768
        W = bf(torch.randn(N, K, device='cuda', generator=g) * 0.03); bias = torch.randn(N, device='cuda', generator=g) * 0.1
        B2 = bf(torch.randn(N, Rp, device='cuda', generator=g) * 0.1)
        R = torch.randn(M, N, device='cuda', generator=g)
        rs = (torch.rand(64, device='cuda', generator=g) > 0.2).float() / 0.8          # per sample (DropPath)
        out = torch.empty(M, N, device='cuda')
        ops.gemm(x, W, out, A2=T[:, :Rp], B2=B2, K2=Rp, bias=bias, R=R, row_scale=rs, rows_per_img=S)
        row_rs = rs[torch.arange(M, device='cuda') // S]
        ref = (xf @ W.float().t() + bias + T.float()[:, :Rp] @ B2.float().t()) * row_rs[:, None] + R
        assert rel_err(out, ref) < 2e-5
        # fc1-like: GELU with the pre-activation kept; then its backward form
        N = 3072
        W = bf(torch.randn(N, K, device='cuda', generator=g) * 0.03); bias = torch.randn(N, device='cuda', generator=g) * 0.1
        h = torch.empty(M, N, device='cuda', dtype=T16()); u = torch.empty(M, N, device='cuda', dtype=T16())
        ops.gemm(x, W, h, bias=bias, act='gelu', C2=u)
        base = xf @ W.float().t() + bias
        assert rel_err(u.float(), base) < 1e-2
        assert rel_err(h.float(), torch.nn.functional.gelu(base)) < 1e-2
        uf = u.float().requires_grad_(True)
        torch.nn.functional.gelu(uf).sum().backward()
        y = bf(torch.randn(M, K, device='cuda', generator=g))
        W2 = bf(torch.randn(N, K, device='cuda', generator=g) * 0.03)   # dU = (dY W2^T) * gelu'(u): [M, 768] x [3072, 768]^T
        du = torch.empty(M, N, device='cuda', dtype=T16())
        ops.gemm(y, W2, du, act='dgelu', aux=u)
        assert rel_err(du.float(), (y.float() @ W2.float().t()) * uf.grad) < 1e-2
        # the pair the training path uses: derivative saved by the forward epilogue, one multiply in the backward one
        dv = torch.empty(M, N, device='cuda', dtype=T16())
        ops.gemm(x, W, h, bias=bias, act='gelu_dsave', C2=dv)
        bb = base.clone().requires_grad_(True)
        torch.nn.functional.gelu(bb).sum().backward()
        assert rel_err(h.float(), torch.nn.functional.gelu(base)) < 1e-2
        assert rel_err(dv.float(), bb.grad) < 1e-2
        ops.gemm(y, W2, du, act='mul_aux', aux=dv)
        assert rel_err(du.float(), (y.float() @ W2.float().t()) * dv.float()) < 1e-2
        # fc2-like: K = 3072 into the residual stream
        A = h
        W = bf(torch.randn(768, 3072, device='cuda', generator=g) * 0.02)
        ops.gemm(A, W, out, R=R)
        assert rel_err(out, A.float() @ W.float().t() + R) < 2e-5
    finally:
        _lib.check(_lib.lib().reid_set_knob(b'GEMM_TILE', -1))


@pytest.mark.parametrize('M,P,Q', [(1000, 768, 32), (4099, 32, 768), (777, 3072, 64), (5000, 96, 768), (2048, 256, 384),
                                   (63, 768, 32)])
def test_gemm_tn(ops, M, P, Q):
    g = torch.Generator(device='cuda').manual_seed(M + P + Q)
    X = bf(torch.randn(M, P, device='cuda', generator=g)); Y = bf(torch.randn(M, Q, device='cuda', generator=g))
    Cm = torch.full((P, Q), 7.0, device='cuda')
    ops.gemm_tn(X, Y, Cm, alpha=0.5, beta=0.0)
    ref = 0.5 * X.float().t() @ Y.float()
    assert rel_err(Cm, ref) < 2e-5
    ops.gemm_tn(X, Y, Cm, alpha=0.5, beta=1.0)
    assert rel_err(Cm, 2 * ref) < 2e-5


@pytest.mark.parametrize('rows,cols', [(1000, 768), (77 * 3, 512), (5, 128)])
def test_layernorm(ops, rows, cols):
    g = torch.Generator(device='cuda').manual_seed(rows)
    x = torch.randn(rows, cols, device='cuda', generator=g) * 2 + 0.5
    gamma = 1 + 0.1 * torch.randn(cols, device='cuda', generator=g); beta = 0.1 * torch.randn(cols, device='cuda', generator=g)
    yb = torch.empty(rows, cols, device='cuda', dtype=T16()); yf = torch.empty(rows, cols, device='cuda')
    mean = torch.empty(rows, device='cuda'); rstd = torch.empty(rows, device='cuda')
    ops.layernorm_fwd(x, gamma, beta, y_bf16=yb, y_f32=yf, mean=mean, rstd=rstd)
    xr = x.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (cols,), gr, br, 1e-5)
    assert rel_err(yf, ref) < 1e-5 and rel_err(yb.float(), ref) < 1e-2
    dy = torch.randn(rows, cols, device='cuda', generator=g); dres = torch.randn(rows, cols, device='cuda', generator=g)
    ref.backward(dy)
    dx = torch.empty_like(x); dxb = torch.empty(rows, cols, device='cuda', dtype=T16())
    dgam = torch.zeros(cols, device='cuda'); dbet = torch.zeros(cols, device='cuda')
    ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx, dx_bf16=dxb, dres=dres, dgamma=dgam, dbeta=dbet)
    assert rel_err(dx, xr.grad + dres) < 2e-5
    assert rel_err(dgam, gr.grad) < 1e-4 and rel_err(dbet, br.grad) < 1e-4
    assert rel_err(dxb.float(), xr.grad + dres) < 1e-2
    # the residual-stream gradient in IEEE half (dres read and dx written): one rounding of the fp32 result, saturating
    dres_h = dres.half(); dx_h = torch.empty(rows, cols, device='cuda', dtype=torch.float16)
    ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx_h, dx_bf16=dxb, dres=dres_h)
    want = xr.grad + dres_h.float()
    assert rel_err(dx_h.float(), want) < 5e-4 and (dx_h.float() - want).abs().max() <= want.abs().max() * 2.0 ** -11 * 1.01
    ops.layernorm_bwd(dy * 1e5, x, gamma, mean, rstd, dx_h)
    assert torch.isfinite(dx_h.float()).all() and dx_h.float().abs().max() == 65504.0
    with pytest.raises(ValueError):
        ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx_h, dres=dres)
    dyb = bf(dy)
    ops.layernorm_bwd(dyb, x, gamma, mean, rstd, dx)
    xr.grad = None
    torch.nn.functional.layer_norm(xr, (cols,), gamma, beta, 1e-5).backward(dyb.float())
    assert rel_err(dx, xr.grad) < 2e-5
    # gathered rows (CLS rows)
    idx = torch.arange(0, rows, 7, device='cuda', dtype=torch.int32)
    y2 = torch.empty(idx.shape[0], cols, device='cuda'); m2 = torch.empty(idx.shape[0], device='cuda'); r2 = torch.empty_like(m2)
    ops.layernorm_fwd(x, gamma, beta, y_f32=y2, mean=m2, rstd=r2, row_index=idx)
    assert rel_err(y2, ref.detach()[idx.long()]) < 1e-5
    dxs = torch.zeros_like(x)
    ops.layernorm_bwd(dy[:idx.shape[0]].contiguous(), x, gamma, m2, r2, dxs, row_index=idx)
    xr.grad = None
    torch.nn.functional.layer_norm(xr[idx.long()], (cols,), gamma, beta, 1e-5).backward(dy[:idx.shape[0]])
    assert rel_err(dxs, xr.grad) < 2e-5


@pytest.mark.parametrize('rows,rpi,cols,scaled', [(197 * 3, 197, 768, True), (64, 1, 768, False), (50 * 5 + 3, 50, 512, True), (8, 4, 64, True)])
def test_add_layernorm(ops, rows, rpi, cols, scaled):
    """x_out = x + scale[image] * y (fp32 add of the 16-bit branch output), h = LN(x_out): clip_backbone.py:76/:83 fused into the
    LayerNorm that follows.  The sum is bit-exact (one fma per element); LN as test_layernorm."""
    g = torch.Generator(device='cuda').manual_seed(rows + cols)
    x = torch.randn(rows, cols, device='cuda', generator=g) * 2 + 0.5
    y = bf(torch.randn(rows, cols, device='cuda', generator=g))
    n_img = (rows + rpi - 1) // rpi
    sc = (torch.rand(n_img, device='cuda', generator=g) > 0.3).float() / 0.7 if scaled else None
    gamma = 1 + 0.1 * torch.randn(cols, device='cuda', generator=g); beta = 0.1 * torch.randn(cols, device='cuda', generator=g)
    xo = torch.empty_like(x); h = torch.empty(rows, cols, device='cuda', dtype=T16())
    mean = torch.empty(rows, device='cuda'); rstd = torch.empty(rows, device='cuda')
    ops.add_layernorm_fwd(x, y, xo, gamma, beta, h, mean, rstd, row_scale=sc, rows_per_img=rpi if scaled else 0)
    if scaled:
        srow = sc.repeat_interleave(rpi)[:rows, None]
        want = torch.addcmul(x.double(), srow.double(), y.double()).float()
    else:
        want = x + y.float()
    assert float((xo - want).abs().max()) <= 1e-6 * float(want.abs().max())
    ref = torch.nn.functional.layer_norm(xo, (cols,), gamma, beta, 1e-5)
    assert rel_err(h.float(), ref) < 1e-2
    assert rel_err(mean, xo.mean(1)) < 1e-5 and rel_err(rstd, (xo.var(1, unbiased=False) + 1e-5).rsqrt()) < 1e-5
    # in place: x_out aliasing x
    x2 = x.clone()
    ops.add_layernorm_fwd(x2, y, x2, gamma, beta, h, row_scale=sc, rows_per_img=rpi if scaled else 0)
    assert torch.equal(x2, xo)


@pytest.mark.parametrize('M,N,K,groups', [(197 * 8, 768, 768, True), (1000, 768, 3072, False), (64, 256, 768, False), (50432, 768, 768, True)])
def test_branch_output_in_ieee_half_both_flavors(ops, M, N, K, groups):
    """r04: the out-projection / fc2 GEMM stores its branch output as IEEE half (REID_F16) in BOTH flavors and the add + LayerNorm kernel
    reads it as such: C = half(A W^T + b) against fp64 within half's rounding (2^-11 relative, where bf16 storage would give 2^-8),
    finite overflow saturates to +-65504, and the fused add sees exactly those half values."""
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    A = bf(torch.randn(M, K, device='cuda', generator=g))
    nW = 4 if groups else 1
    W = bf(torch.randn(nW, N, K, device='cuda', generator=g) / math.sqrt(K))
    bias = torch.randn(N, device='cuda', generator=g)
    C = torch.empty(M, N, device='cuda', dtype=torch.float16)
    if groups:
        q = (M // 197) // 4 * 197 if M % 197 == 0 else M // 4
        ends = [q, 2 * q, 3 * q, M]
        ops.gemm(A, W, C, bias=bias, row_groups=(ends, [2, 0, 3, 1]))
        want = torch.empty(M, N, device='cuda', dtype=torch.float64)
        lo = 0
        for e_, w_ in zip(ends, [2, 0, 3, 1]):
            want[lo:e_] = A[lo:e_].double() @ W[w_].double().t() + bias.double(); lo = e_
    else:
        ops.gemm(A, W[0], C, bias=bias)
        want = A.double() @ W[0].double().t() + bias.double()
    err = (C.double() - want).abs()
    tol = 2.0 ** -11 * want.abs() + 2.0 ** -24 + 1e-5 * want.abs().max()      # half rounding + fp32 accumulation noise
    assert bool((err <= tol).all()), float((err / tol).max())
    # saturation: a bias far outside half's range comes out as +-65504, not inf
    big = bias.clone(); big[0] = 1e6; big[1] = -1e6
    ops.gemm(A[:64], W[0], C[:64], bias=big)
    assert float(C[:64, 0].float().min()) == 65504.0 and float(C[:64, 1].float().max()) == -65504.0 and bool(torch.isfinite(C[:64].float()).all())
    # consumed by the add + LayerNorm as half: x_out = x + C exactly
    ops.gemm(A, W[0], C, bias=bias)
    x = torch.randn(M, N, device='cuda', generator=g)
    xo = torch.empty_like(x); h = torch.empty(M, N, device='cuda', dtype=T16())
    gamma = torch.ones(N, device='cuda'); beta = torch.zeros(N, device='cuda')
    ops.add_layernorm_fwd(x, C, xo, gamma, beta, h)
    assert torch.equal(xo, x + C.float())
    # an activation / second output with a half C is refused (only the plain store exists in this format)
    from prcv2025reid_amd import _lib
    if _lib.flavor() == 'bf16':
        with pytest.raises(_lib.ReidHipError):
            ops.gemm(A, W[0], C, bias=bias, act='gelu')


def _attn_ref(qkv, n_seq, S, heads, causal, key_mask):
    d = heads * 64
    q, k, v = [t.view(n_seq, S, heads, 64).transpose(1, 2) for t in qkv.split(d, dim=1)]
    s = (q @ k.transpose(-1, -2)) * 0.125
    if key_mask is not None:
        s = s.masked_fill(~key_mask.bool().view(n_seq, 1, 1, S), float('-inf'))
    if causal:
        s = s.masked_fill(torch.ones(S, S, device=qkv.device).triu(1).bool(), float('-inf'))
    p = torch.softmax(s, dim=-1)
    return (p @ v).transpose(1, 2).reshape(n_seq * S, d), torch.logsumexp(s, dim=-1)


@pytest.mark.parametrize('n_seq,S,heads,causal,masked', [(3, 197, 12, False, False), (4, 77, 8, True, True),
                                                         (6, 5, 8, False, True), (2, 33, 2, False, False),
                                                         (2, 224, 2, False, False), (5, 16, 8, True, False)])
def test_attention(ops, n_seq, S, heads, causal, masked):
    g = torch.Generator(device='cuda').manual_seed(S + heads)
    d = heads * 64
    qkv = bf(torch.randn(n_seq * S, 3 * d, device='cuda', generator=g))
    km = None
    if masked:
        km = (torch.rand(n_seq, S, device='cuda', generator=g) > 0.3).to(torch.uint8)
        km[:, 0] = 1
    out = torch.empty(n_seq * S, d, device='cuda', dtype=T16())
    lse = torch.empty(n_seq, heads, S, device='cuda')
    ops.attn_fwd(qkv, out, lse, n_seq, S, heads, causal=causal, key_mask=km)
    qf = qkv.float().requires_grad_(True)
    ref, ref_lse = _attn_ref(qf, n_seq, S, heads, causal, km)
    assert rel_err(out.float(), ref) < 2e-2
    assert float((lse - ref_lse).abs().max()) < 1e-3
    dout = bf(torch.randn(n_seq * S, d, device='cuda', generator=g))
    ref.backward(dout.float())
    dqkv = torch.zeros(n_seq * S, 3 * d, device='cuda', dtype=T16())
    delta = torch.empty(n_seq, heads, S, device='cuda')
    ops.attn_bwd(qkv, out, dout, lse, dqkv, delta, n_seq, S, heads, causal=causal, key_mask=km)
    for i, nm in enumerate('qkv'):
        a = dqkv[:, i * d:(i + 1) * d].float(); b = qf.grad[:, i * d:(i + 1) * d]
        assert rel_err(a, b) < 3e-2, nm


@pytest.mark.parametrize('n_seq,S,heads,q_tiles', [(48, 197, 12, 0), (3, 197, 12, 0), (70, 50, 8, 0), (9, 64, 4, 0), (40, 197, 12, 1), (300, 5, 2, 0)])
def test_attention_backward_forms(ops, n_seq, S, heads, q_tiles):
    """The three forms of the attention backward (REID_ATTN_BWD 1 = two kernels, 2 = one pass with one item per workgroup, 3 = one pass,
    persistent workgroups with the next item's images staged under the current item's arithmetic) against each other and against fp32
    autograd, on item counts below and above two per CU, tile counts 1..7, and with the query-tile limit of the pruned last block."""
    from prcv2025reid_amd import _lib
    g = torch.Generator(device='cuda').manual_seed(n_seq + S)
    d = heads * 64
    qkv = bf(torch.randn(n_seq * S, 3 * d, device='cuda', generator=g))
    out = torch.empty(n_seq * S, d, device='cuda', dtype=T16())
    lse = torch.empty(n_seq, heads, S, device='cuda')
    ops.attn_fwd(qkv, out, lse, n_seq, S, heads, q_tiles=q_tiles)
    dout = bf(torch.randn(n_seq * S, d, device='cuda', generator=g))
    qrows = min(S, 32 * q_tiles) if q_tiles else S
    if q_tiles:                                           # rows beyond the limit take no part: their cotangent and output are unused
        keep = torch.zeros(n_seq, S, 1, device='cuda'); keep[:, :qrows] = 1
        dout = bf(dout.float().view(n_seq, S, d) * keep).view(n_seq * S, d)
    res = {}
    for impl in (1, 2, 3):
        _lib.check(_lib.lib().reid_set_knob(b'ATTN_BWD', impl))
        try:
            dqkv = torch.full((n_seq * S, 3 * d), float('nan'), device='cuda', dtype=T16())
            delta = torch.empty(n_seq, heads, S, device='cuda')
            ops.attn_bwd(qkv, out, dout, lse, dqkv, delta, n_seq, S, heads, q_tiles=q_tiles)
            torch.cuda.synchronize()
            res[impl] = dqkv.float()
        finally:
            _lib.check(_lib.lib().reid_set_knob(b'ATTN_BWD', -1))
    assert bool(torch.isfinite(res[2]).all()) and bool(torch.isfinite(res[3]).all())
    # the one-pass forms run the same arithmetic in the same order per element as the two-kernel form (delta from the same products)
    for impl in (2, 3):
        assert rel_err(res[impl], res[1]) < 2e-3, impl
    assert torch.equal(res[2], res[3]) or rel_err(res[3], res[2]) < 1e-6
    qf = qkv.float().requires_grad_(True)
    ref, _ = _attn_ref(qf, n_seq, S, heads, False, None)
    ref.backward(dout.float())
    want = qf.grad
    if q_tiles:
        want = want.clone().view(n_seq, S, 3 * d)
        want[:, qrows:, :d] = 0                           # dQ of the rows left out is exactly zero (their own softmax rows are not formed)
        want = want.view(n_seq * S, 3 * d)
    for i, nm in enumerate('qkv'):
        assert rel_err(res[3][:, i * d:(i + 1) * d], want[:, i * d:(i + 1) * d]) < 3e-2, nm


def test_patch_and_cls(ops):
    g = torch.Generator(device='cuda').manual_seed(3)
    img = torch.randn(5, 3, 224, 224, device='cuda', generator=g)
    for cin in (3, 1):
        P = torch.empty(5 * 196, cin * 256, device='cuda', dtype=T16())
        ops.patch_im2col(img, P, 16, cin)
        x = img if cin == 3 else img.mean(1, keepdim=True)
        ref = x.view(5, cin, 14, 16, 14, 16).permute(0, 2, 4, 1, 3, 5).reshape(5 * 196, cin * 256)
        assert rel_err(P.float(), ref) < 5e-3
    cls = torch.randn(768, device='cuda', generator=g); pos = torch.randn(197, 768, device='cuda', generator=g)
    x = torch.zeros(5 * 197, 768, device='cuda')
    ops.cls_rows(cls, pos, x, 5, 197)
    assert torch.allclose(x.view(5, 197, 768)[:, 0], (cls + pos[0]).expand(5, -1))
    assert float(x.view(5, 197, 768)[:, 1:].abs().max()) == 0


def test_cast_and_l2norm(ops):
    g = torch.Generator(device='cuda').manual_seed(4)
    x = torch.randn(1000003, device='cuda', generator=g)
    assert torch.equal(ops.to_bf16(x), x.to(T16()))
    f = torch.randn(300, 512, device='cuda', generator=g)
    y = torch.empty_like(f); yb = torch.empty(300, 512, device='cuda', dtype=T16())
    ops.l2norm_rows(f, y=y, y_bf16=yb)
    assert rel_err(y, torch.nn.functional.normalize(f, dim=1)) < 1e-6


@pytest.mark.parametrize('ta,tb', [(False, False), (False, True), (True, False)])
def test_sgemm(ops, ta, tb):
    g = torch.Generator(device='cuda').manual_seed(9)
    M, N, K = 70, 130, 100
    A = torch.randn((K, M) if ta else (M, K), device='cuda', generator=g)
    B = torch.randn((N, K) if tb else (K, N), device='cuda', generator=g)
    bias = torch.randn(N, device='cuda', generator=g)
    Cm = torch.randn(M, N, device='cuda', generator=g); C0 = Cm.clone()
    ops.sgemm(A, B, Cm, ta=ta, tb=tb, alpha=0.5, beta=2.0, bias=bias, act='relu')
    ref = torch.relu(0.5 * (A.t() if ta else A) @ (B.t() if tb else B) + bias) + 2.0 * C0
    assert rel_err(Cm, ref) < 1e-5


@pytest.mark.parametrize('rows,training', [(64, True), (37, True), (64, False), (1024, True)])
def test_bnneck(ops, rows, training):
    g = torch.Generator(device='cuda').manual_seed(rows)
    D = 512
    x = torch.randn(rows, D, device='cuda', generator=g) * 1.5 + 0.3
    gamma = 1 + 0.1 * torch.randn(D, device='cuda', generator=g); beta = 0.1 * torch.randn(D, device='cuda', generator=g)
    rm = 0.1 * torch.randn(D, device='cuda', generator=g); rv = 1 + 0.1 * torch.rand(D, device='cuda', generator=g)
    bn = torch.nn.BatchNorm1d(D).cuda()
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
    bn.train(training)
    xr = x.clone().requires_grad_(True)
    ref = torch.nn.functional.normalize(bn(xr), dim=1) * 8.0
    s1 = torch.empty(D, device='cuda'); s2 = torch.empty(D, device='cuda')
    y = torch.empty(rows, D, device='cuda'); yb = torch.empty(rows, D, device='cuda', dtype=T16())
    mean = torch.empty(D, device='cuda'); invstd = torch.empty(D, device='cuda'); rn = torch.empty(rows, device='cuda')
    rm2, rv2 = rm.clone(), rv.clone()
    if training:
        ops.bnneck_stats(x, s1, s2)
    ops.bnneck_fwd(x, gamma, beta, rm2, rv2, s1, s2, float(rows), training, y, yb, mean, invstd, rn)
    assert rel_err(y, ref) < 2e-5
    assert rel_err(rm2, bn.running_mean) < 1e-5 and rel_err(rv2, bn.running_var) < 1e-5
    dy = torch.randn(rows, D, device='cuda', generator=g)
    ref.backward(dy)
    dz = torch.empty(rows, D, device='cuda'); a = torch.empty(D, device='cuda'); b = torch.empty(D, device='cuda')
    dx = torch.empty(rows, D, device='cuda')
    ops.bnneck_bwd_p1(dy, x, gamma, beta, mean, invstd, rn, dz, a, b)
    ops.bnneck_bwd_p2(dz, x, gamma, mean, invstd, a, b, float(rows), training, dx)
    assert rel_err(dx, xr.grad) < 1e-4
    assert rel_err(b, bn.weight.grad) < 1e-4 and rel_err(a, bn.bias.grad) < 1e-4


def test_ce_label_smoothing(ops):
    g = torch.Generator(device='cuda').manual_seed(2)
    rows, C = 67, 400
    z = torch.randn(rows, C, device='cuda', generator=g) * 3
    lab = torch.randint(0, C, (rows,), device='cuda', generator=g)
    lab[3] = -1; lab[5] = C + 2
    valid = torch.ones(rows, device='cuda', dtype=torch.uint8); valid[7] = 0
    ok = (valid.bool()) & (lab >= 0) & (lab < C)
    zr = z.clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(zr[ok], lab[ok], label_smoothing=0.1)
    rl = torch.empty(rows, device='cuda'); acc = torch.zeros(2, device='cuda')
    ops.ce_ls_fwd(z, lab, valid, rl, acc)
    assert abs(float(acc[0] / acc[1]) - float(ref)) < 1e-5 and int(acc[1]) == int(ok.sum())
    (ref * 0.7).backward()
    gs = torch.tensor([0.7 / float(acc[1])], device='cuda')
    dl = torch.empty(rows, C, device='cuda')
    ops.ce_ls_bwd(z, lab, valid, gs, dl)
    assert rel_err(dl, zr.grad) < 1e-5


@pytest.mark.parametrize('P,N,Mg,D', [(1, 16, 48, 512), (1, 64, 64, 512), (4, 64, 64, 512), (3, 130, 70, 512), (2, 516, 1024, 512),
                                      (4, 200, 200, 256), (1, 1030, 650, 512)])
def test_sdm(ops, P, N, Mg, D):
    """Fused SDM (csrc/sdm.hip): P stacked query sides against one gallery side, vs the oracle's sdm_loss per pair
    (1e-5 on the loss, 1e-4 relative on both gradients); rows / columns masked out; tiles of 64 and of 128."""
    from oracle import reid_oracle as O
    g = torch.Generator(device='cuda').manual_seed(N + 7 * P)
    q = torch.randn(P * N, D, device='cuda', generator=g); gal = torch.randn(Mg, D, device='cuda', generator=g)
    ql = torch.randint(0, 10, (N,), device='cuda', generator=g); gl = torch.randint(0, 10, (Mg,), device='cuda', generator=g)
    qv = (torch.rand(P * N, device='cuda', generator=g) > 0.2).to(torch.uint8); gv = (torch.rand(Mg, device='cuda', generator=g) > 0.2).to(torch.uint8)
    if P > 1:
        qv[N:2 * N] = 0; qv[N] = 1; ql = ql.clone(); ql[0] = 77          # pair 1: one valid row whose label has no partner -> no positive
        gl = gl.clone(); gl[gl == 77] = 0
    ws = torch.empty(ops.sdm_ws_floats(P, N, Mg, D), device='cuda'); res = torch.zeros(2 * P, device='cuda')
    ops.sdm_fwd(q, gal, ql, gl, qv, gv, 0.2, ws, res, P=P)
    qc = q.cpu().requires_grad_(True); gc = gal.cpu().requires_grad_(True)
    gi = gv.cpu().bool()
    gs = torch.linspace(0.7, 1.3, P)
    tot = 0.0
    for p_ in range(P):
        qi = qv.cpu()[p_ * N:(p_ + 1) * N].bool()
        y = (ql.cpu()[qi].view(-1, 1) == gl.cpu()[gi].view(1, -1)).float()
        ref = O.sdm_loss(qc[p_ * N:(p_ + 1) * N][qi], gc[gi], y, tau=0.2)
        contributes = float(y.sum()) > 0
        assert abs(float(res[2 * p_]) - float(ref)) < 1e-5, (p_, float(res[2 * p_]), float(ref))
        assert float(res[2 * p_ + 1]) == (1.0 if contributes else 0.0)
        tot = tot + ref * gs[p_]
    tot.backward()
    dq = torch.zeros(P * N, D, device='cuda'); dg = torch.zeros(Mg, D, device='cuda')
    ops.sdm_bwd(q, gal, ql, gl, qv, gv, 0.2, ws, gs.cuda(), dq, dg, P=P)
    assert rel_err(dq.cpu(), qc.grad) < 1e-4 and rel_err(dg.cpu(), gc.grad) < 1e-4
    assert float(dq[(qv == 0)].abs().max()) == 0.0                        # masked rows get no gradient
    # no positives at all -> 0 and "does not contribute"
    ops.sdm_fwd(q, gal, ql, gl + 100, qv, gv, 0.2, ws, res, P=P)
    assert float(res.abs().max()) == 0.0
    assert ops.sdm_ws_floats(1, 8192, 8192, 512) < 3 * (8192 + 8192) * 512          # O((N + M) D): no N x M term


@pytest.mark.parametrize('Nq,Ng,k', [(64, 4096, 10), (200, 20000, 10), (33, 777, 100), (130, 5000, 1)])
def test_cosine_topk_matches_fp32_stable_argsort(ops, Nq, Ng, k):
    g = torch.Generator(device='cuda').manual_seed(Nq + Ng)
    D = 512
    Q = torch.nn.functional.normalize(torch.randn(Nq, D, device='cuda', generator=g), dim=1)
    G = torch.nn.functional.normalize(torch.randn(Ng, D, device='cuda', generator=g), dim=1)
    G[5] = G[3]; G[100] = G[3]                      # exact ties: index order must decide
    Q[0] = G[3]
    exq = torch.full((Nq,), -1, device='cuda', dtype=torch.int32); exg = torch.full((Ng,), -1, device='cuda', dtype=torch.int32)
    exq[1] = 7; exg[torch.randint(0, Ng, (50,), device='cuda', generator=g)] = 7
    ws = torch.empty(ops.topk_ws_bytes(Nq, Ng, k), device='cuda', dtype=torch.uint8)
    idx = torch.empty(Nq, k, device='cuda', dtype=torch.int32); sc = torch.empty(Nq, k, device='cuda')
    ops.cosine_topk(ops.to_bf16(Q), ops.to_bf16(G), Q, G, k, ws, idx, sc, exclude_q=exq, exclude_g=exg)
    assert int((idx[:, 0] == -2).sum()) == 0
    sim = (Q.double() @ G.double().t())
    sim = sim.masked_fill((exq.view(-1, 1) >= 0) & (exq.view(-1, 1) == exg.view(1, -1)), -1e9)
    # fp32 oracle order == order of the exact scores unless two exact scores are closer than fp32 rounding; compare
    # against the ranking by (fp32 score of the kernel's own arithmetic) is circular, so check with tolerance-aware rule
    ref = torch.argsort(sim.float(), dim=1, descending=True, stable=True)[:, :k]
    same = (ref == idx.long())
    if not bool(same.all()):
        bad = (~same).nonzero()
        for qi, r in bad.tolist():
            a, b = int(ref[qi, r]), int(idx[qi, r])
            assert abs(float(sim[qi, a] - sim[qi, b])) < 2e-7, (qi, r, a, b)
    if k >= 3:
        assert idx[0, :3].tolist() == [3, 5, 100]


@pytest.mark.parametrize('n_img,rpi,r,lddy_extra', [(24, 197, 8, 0), (7, 197, 4, 1536), (256, 1, 8, 0), (3, 50, 8, 768), (100, 197, 8, 0), (5, 32, 2, 0), (4, 33, 8, 0)])
def test_lora_bwd_fused_equals_two_launch_path(ops, n_img, rpi, r, lddy_extra):
    """reid_lora_bwd_fused (U = mask(dY.B) * s and dB += dY^T.T from one pass over dY) against the two launches it replaces
    (reid_mer_gemm with the modality mask + reid_gemm_tn) and against fp64: ragged last row step, dY as a column block of a wider
    matrix (the q|k|v cotangent), class-row form (one row per image), accumulation into a non-zero dB.  Both kernels behind the entry
    point: one image per workgroup (r04: the default where an image spans a 32-row step; T modality-masked, as the forward produces it)
    and the row-slab kernel (REID_LORA_IMPL=1; any T)."""
    from prcv2025reid_amd import _lib
    M, N, Rp = n_img * rpi, 768, 32
    g = torch.Generator(device='cuda').manual_seed(M + r)
    wide = torch.randn(M, N + lddy_extra, device='cuda', generator=g)
    dYw = bf(wide); dY = dYw[:, lddy_extra // 2: lddy_extra // 2 + N] if lddy_extra else dYw
    mods = torch.randint(0, 4, (n_img,), device='cuda', generator=g).to(torch.int32)
    row_mod = mods.long().repeat_interleave(rpi)
    keep = (torch.arange(Rp, device='cuda').view(1, -1) // r) == row_mod.view(-1, 1)
    T_any = bf(torch.randn(M, Rp, device='cuda', generator=g) * 0.3)
    T_masked = bf(T_any.float() * keep)
    B = torch.randn(N, Rp, device='cuda', generator=g) * 0.1
    BT = bf(B.t().contiguous())
    scale = 32.0 / r
    dB0 = torch.randn(N, Rp, device='cuda', generator=g)
    for impl, Tm in ((-1, T_masked), (1, T_any), (1, T_masked)):
        # the two-launch path
        U_ref = torch.empty(M, Rp, device='cuda', dtype=T16()); dB_ref = dB0.clone()
        ops.gemm(dY, BT, U_ref, img_mod=mods, mask_r=r, mask_period=Rp, rows_per_img=rpi, alpha=scale)
        ops.gemm_tn(dY, Tm, dB_ref, beta=1.0)
        U = torch.full((M, Rp), 7.0, device='cuda').to(T16()); dB = dB0.clone()
        _lib.check(_lib.lib().reid_set_knob(b'LORA_IMPL', impl))
        try:
            ops.lora_bwd_fused(dY, Tm, BT, U, dB, mods, rpi, r, scale)
        finally:
            _lib.check(_lib.lib().reid_set_knob(b'LORA_IMPL', -1))
        U64 = (dY.double() @ BT.double().t()) * scale * keep
        dB64 = dB0.double() + dY.double().t() @ Tm.double()
        assert rel_err(U.float(), U64.float()) < 6e-3 and rel_err(dB, dB64.float()) < 1e-5, impl
        assert rel_err(U.float(), U_ref.float()) < 3e-3 and rel_err(dB, dB_ref) < 1e-5, impl
        assert float(U.float()[~keep].abs().max()) == 0.0                      # other modalities' columns are exactly zero


@pytest.mark.parametrize('n_img,rpi,r,K,G', [(24, 197, 8, 768, 1), (9, 197, 8, 3072, 1), (11, 197, 8, 768, 3), (3, 50, 4, 768, 3), (100, 197, 8, 768, 1), (5, 33, 2, 1536, 1)])
def test_lora_da_fused_equals_gemm_tn(ops, n_img, rpi, r, K, G):
    """reid_lora_da_fused (dA += U^T . X, one image per workgroup, the U windows of the image's modality only) against reid_gemm_tn(U, X)
    and fp64: one and three adapter groups (q|k|v), column blocks of a wide input (fc2's 3072), ragged last step, accumulation into a
    non-zero dA; rows of other modalities' adapters stay untouched."""
    M, Rp = n_img * rpi, 32
    g = torch.Generator(device='cuda').manual_seed(M + K + G)
    X = bf(torch.randn(M, K, device='cuda', generator=g))
    mods = torch.randint(0, 4, (n_img,), device='cuda', generator=g).to(torch.int32)
    row_mod = mods.long().repeat_interleave(rpi)
    keep = ((torch.arange(G * Rp, device='cuda').view(1, -1) % Rp) // r) == row_mod.view(-1, 1)
    U = bf(torch.randn(M, G * Rp, device='cuda', generator=g) * 0.2 * keep)
    dA0 = torch.randn(G * Rp, K, device='cuda', generator=g)
    ref = dA0.clone()
    ops.gemm_tn(U, X, ref, beta=1.0)
    dA = dA0.clone()
    assert ops.lora_da_fused_ok(K, Rp, rpi, r, G)
    ops.lora_da_fused(X, U, dA, mods, rpi, r, n_groups=G)
    want = dA0.double() + U.double().t() @ X.double()
    assert rel_err(dA, want.float()) < 1e-5 and rel_err(dA, ref) < 1e-5


def test_lora_bwd_fused_wide_cotangent_as_column_blocks(ops):
    """fc1's cotangent has 3072 columns: four launches over 768-column blocks, U summed through the fp32 scratch, equal the two-launch
    path on the whole matrix."""
    n_img, rpi, r, N, Rp = 5, 197, 8, 3072, 32
    M = n_img * rpi
    g = torch.Generator(device='cuda').manual_seed(11)
    dY = bf(torch.randn(M, N, device='cuda', generator=g))
    mods = torch.randint(0, 4, (n_img,), device='cuda', generator=g).to(torch.int32)
    keep = (torch.arange(Rp, device='cuda').view(1, -1) // r) == mods.long().repeat_interleave(rpi).view(-1, 1)
    Tm = bf(torch.randn(M, Rp, device='cuda', generator=g) * 0.3 * keep)            # modality-masked, as the forward produces it
    BT = bf((torch.randn(N, Rp, device='cuda', generator=g) * 0.05).t().contiguous())
    scale = 4.0
    U_ref = torch.empty(M, Rp, device='cuda', dtype=T16()); dB_ref = torch.zeros(N, Rp, device='cuda')
    ops.gemm(dY, BT, U_ref, img_mod=mods, mask_r=r, mask_period=Rp, rows_per_img=rpi, alpha=scale)
    ops.gemm_tn(dY, Tm, dB_ref, beta=1.0)
    from prcv2025reid_amd import _lib
    for impl in (-1, 1):
        U = torch.full((M, Rp), 7.0, device='cuda').to(T16()); dB = torch.zeros(N, Rp, device='cuda')
        scratch = torch.full((M, Rp), float('nan'), device='cuda')            # never read before it is written
        _lib.check(_lib.lib().reid_set_knob(b'LORA_IMPL', impl))
        try:
            ops.lora_bwd_fused(dY, Tm, BT, U, dB, mods, rpi, r, scale, u_partial=scratch)
        finally:
            _lib.check(_lib.lib().reid_set_knob(b'LORA_IMPL', -1))
        assert rel_err(U.float(), U_ref.float()) < 3e-3 and rel_err(dB, dB_ref) < 1e-5, impl
        assert bool(torch.isfinite(U.float()).all())


@pytest.mark.parametrize('Nq,Ng,D,k', [(128, 200000, 512, 10), (37, 50000, 256, 32), (5, 1000, 512, 7), (3, 40, 64, 10)])
def test_cosine_topk_fast_select_equals_first_form(ops, Nq, Ng, D, k):
    """Phase C of the batched retrieval in its parallel form (select_fast_kernel: k <= 32) returns the same indices and the same
    fp32 score bits as the first form (select_kernel, forced with knob TOPK_TILE=9), with planted exact ties, a same-image
    exclusion and a query that has fewer than k usable rows."""
    from prcv2025reid_amd import _lib
    g = torch.Generator(device='cuda').manual_seed(Nq * 7 + k)
    Q = torch.nn.functional.normalize(torch.randn(Nq, D, device='cuda', generator=g), dim=1)
    G = torch.nn.functional.normalize(torch.randn(Ng, D, device='cuda', generator=g), dim=1)
    G[5] = G[3]; G[17] = G[3]; Q[0] = G[3]
    exq = torch.full((Nq,), -1, device='cuda', dtype=torch.int32); exg = torch.full((Ng,), -1, device='cuda', dtype=torch.int32)
    exq[1] = 7; exg[torch.randint(0, Ng, (min(50, Ng // 2),), device='cuda', generator=g)] = 7
    Qb, Gb = ops.to_bf16(Q), ops.to_bf16(G)
    ws = torch.empty(ops.topk_ws_bytes(Nq, Ng, k), device='cuda', dtype=torch.uint8)
    res = []
    try:
        for knob in (-1, 9):
            _lib.check(_lib.lib().reid_set_knob(b'TOPK_TILE', knob))
            idx = torch.full((Nq, k), -7, device='cuda', dtype=torch.int32); sc = torch.zeros(Nq, k, device='cuda')
            ops.cosine_topk(Qb, Gb, Q, G, k, ws, idx, sc, exclude_q=exq, exclude_g=exg)
            res.append((idx, sc))
    finally:
        _lib.check(_lib.lib().reid_set_knob(b'TOPK_TILE', -1))
    assert int((res[0][0][:, 0] == -2).sum()) == 0 and int((res[1][0][:, 0] == -2).sum()) == 0
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1].view(torch.int32), res[1][1].view(torch.int32))
    if k >= 3:
        assert res[0][0][0, :3].tolist() == [3, 5, 17]


@pytest.mark.parametrize('Nq,Ng,D,k,dups', [(5, 70037, 512, 10, 0), (33, 65536 + 64, 512, 10, 3), (128, 200000, 512, 10, 0), (97, 131072 + 1, 256, 16, 3),
                                             (128, 81920, 512, 1, 0), (64, 70000, 512, 10, 4000)])
def test_cosine_topk_query_resident_scan_equals_tiled_filter(ops, Nq, Ng, D, k, dups):
    """<= 128 queries take the query-resident scan (scan::scan_filter_kernel: the 16-bit gallery streamed once through LDS, the queries as
    MFMA operands in registers, the bar of each query from the running group maxima of the scan itself) in place of the sample +
    threshold + tiled filter launches (knob TOPK_SCAN=0).  Same candidate-list contract, so the final lists and fp32 score bits must be
    identical: ragged last workgroup / last tile, planted exact ties, same-image exclusions that remove the best rows, near-duplicate
    clusters (dups: copies of the best row + 1e-4 noise; 4000 of them overflow the candidate list -> exact fallback through GalleryIndex),
    a query with an id nobody shares, k = 1 and k = 16 (= the kernel's group limit)."""
    from prcv2025reid_amd import _lib
    from prcv2025reid_amd.retrieval import GalleryIndex
    g = torch.Generator(device='cuda').manual_seed(Nq * 11 + k + dups)
    Q = torch.nn.functional.normalize(torch.randn(Nq, D, device='cuda', generator=g), dim=1)
    G = torch.nn.functional.normalize(torch.randn(Ng, D, device='cuda', generator=g), dim=1)
    G[5] = G[3]; G[Ng - 1] = G[3]; Q[0] = G[3]                                   # exact ties, one of them in the ragged tail
    if dups:
        where = torch.randint(0, Ng, (dups,), device='cuda', generator=g)
        G[where] = torch.nn.functional.normalize(Q[2].view(1, -1) + 1e-4 * torch.randn(dups, D, device='cuda', generator=g), dim=1)
    gid = torch.full((Ng,), -1, device='cuda', dtype=torch.int32); qid = torch.full((Nq,), -1, device='cuda', dtype=torch.int32)
    sim0 = Q[1].double() @ G.double().t()
    gid[torch.topk(sim0, 3).indices] = 7; qid[1] = 7                             # the three best rows of query 1 are its own image
    gid[torch.randint(0, Ng, (200,), device='cuda', generator=g)] = 9; qid[Nq - 1] = 11
    index = GalleryIndex(G, normalized=True, img_ids=gid)
    res = []
    try:
        for knob in (-1, 0):
            _lib.check(_lib.lib().reid_set_knob(b'TOPK_SCAN', knob))
            res.append(index.topk(Q, k=k, normalized=True, query_img_ids=qid, stream=False))
            res.append(index.topk(Q, k=k, normalized=True, stream=False))                     # (the kernel without exclusions)
    finally:
        _lib.check(_lib.lib().reid_set_knob(b'TOPK_SCAN', -1))
    for a, b in ((0, 2), (1, 3)):
        assert int((res[a][0] < -1).sum()) == 0
        assert torch.equal(res[a][0], res[b][0])
        assert torch.equal(res[a][1].view(torch.int32), res[b][1].view(torch.int32))
    if k >= 3 and not dups:
        assert res[1][0][0, :3].tolist() == [3, 5, Ng - 1]
    sim = Q.double() @ G.double().t()
    for r_, masked in ((res[0], True), (res[1], False)):
        sm = sim.masked_fill((qid.view(-1, 1) >= 0) & (qid.view(-1, 1) == gid.view(1, -1)), -1e9) if masked else sim
        ref = torch.argsort(sm.float(), dim=1, descending=True, stable=True)[:, :k]
        for qi, r in (ref != r_[0].long()).nonzero().tolist():                               # only fp32-rounding near-ties may differ
            assert abs(float(sm[qi, int(ref[qi, r])] - sm[qi, int(r_[0][qi, r])])) < 2e-7, (qi, r)


def test_cosine_topk_query_resident_scan_on_a_shared_chip(ops):
    """The scan's workgroups exchange their running maxima through memory and never wait for one another: when another stream holds
    most of the chip (here: large GEMMs launched back to back on a second stream), some workgroups run long before the others, find no
    complete bar, poll a bounded number of times and mark their queries for the exact pass.  Whatever the interleaving, the lists and
    score bits must equal the tiled path's (and the call must return)."""
    from prcv2025reid_amd import _lib
    from prcv2025reid_amd.retrieval import GalleryIndex
    Nq, Ng, D, k = 96, 150000, 512, 10
    g = torch.Generator(device='cuda').manual_seed(5)
    Q = torch.nn.functional.normalize(torch.randn(Nq, D, device='cuda', generator=g), dim=1)
    G = torch.nn.functional.normalize(torch.randn(Ng, D, device='cuda', generator=g), dim=1)
    index = GalleryIndex(G, normalized=True)
    try:
        _lib.check(_lib.lib().reid_set_knob(b'TOPK_SCAN', 0))
        want = index.topk(Q, k=k, normalized=True, stream=False)
    finally:
        _lib.check(_lib.lib().reid_set_knob(b'TOPK_SCAN', -1))
    torch.cuda.synchronize()
    A = torch.randn(16384, 4096, device='cuda', generator=g).to(T16()); B = torch.randn(4096, 4096, device='cuda', generator=g).to(T16())
    C = torch.empty(16384, 4096, device='cuda', dtype=T16())
    side = torch.cuda.Stream()
    got = []
    for rep in range(6):
        with torch.cuda.stream(side):
            for _ in range(4 + rep): ops.gemm(A, B, C)           # ~0.3 ms each: the chip is busy while the scan is dispatched
        got.append(index.topk(Q, k=k, normalized=True, stream=False))
    torch.cuda.synchronize()
    for i_, s_ in got:
        assert int((i_ < -1).sum()) == 0
        assert torch.equal(i_, want[0]) and torch.equal(s_.view(torch.int32), want[1].view(torch.int32))


def test_cosine_topk_query_resident_scan_without_a_bar(ops):
    """A query whose same-image id matches EVERY gallery row never gets a bar (none of its rows may count), and its whole 32-query wave
    then has none: the scan marks those queries for the exact pass instead of comparing against nothing.  Result = the tiled path's."""
    from prcv2025reid_amd import _lib
    from prcv2025reid_amd.retrieval import GalleryIndex
    Nq, Ng, D, k = 40, 66000, 512, 10
    g = torch.Generator(device='cuda').manual_seed(77)
    Q = torch.nn.functional.normalize(torch.randn(Nq, D, device='cuda', generator=g), dim=1)
    G = torch.nn.functional.normalize(torch.randn(Ng, D, device='cuda', generator=g), dim=1)
    gid = torch.full((Ng,), 4, device='cuda', dtype=torch.int32); qid = torch.full((Nq,), -1, device='cuda', dtype=torch.int32)
    qid[3] = 4                                              # every row of the gallery is "the same image" as query 3
    index = GalleryIndex(G, normalized=True, img_ids=gid)
    res = []
    try:
        for knob in (-1, 0):
            _lib.check(_lib.lib().reid_set_knob(b'TOPK_SCAN', knob))
            res.append(index.topk(Q, k=k, normalized=True, query_img_ids=qid, stream=False))
    finally:
        _lib.check(_lib.lib().reid_set_knob(b'TOPK_SCAN', -1))
    assert int((res[0][0] < -1).sum()) == 0
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1].view(torch.int32), res[1][1].view(torch.int32))
    sim = Q.double() @ G.double().t()
    ref = torch.argsort(sim.float(), dim=1, descending=True, stable=True)[:, :k]
    ok = torch.ones(Nq, dtype=torch.bool, device='cuda'); ok[3] = False
    assert int((ref[ok] != res[0][0][ok].long()).sum()) <= 2          # (fp32 near-ties only)


@pytest.mark.parametrize('Nq,Ng,D,k', [(1, 5000, 512, 10), (3, 20001, 512, 10), (4, 777, 256, 32), (3, 13, 512, 10), (2, 4096, 1024, 1),
                                        (4, 100000, 512, 10)])
def test_cosine_topk_stream_equals_batched_path(ops, Nq, Ng, D, k):
    """The one-pass form for a few queries (reference: one query at a time, eval_mm_protocol.py:401-455) returns the very
    lists and fp32 scores of the batched pipeline, ties and same-image exclusion included."""
    from prcv2025reid_amd.retrieval import GalleryIndex
    assert ops.topk_stream_ok(Nq, Ng, D, k) and not ops.topk_stream_ok(5, Ng, D, k) and not ops.topk_stream_ok(1, Ng, 320, 10)
    g = torch.Generator(device='cuda').manual_seed(Nq * 7 + Ng)
    Q = torch.nn.functional.normalize(torch.randn(Nq, D, device='cuda', generator=g), dim=1)
    G = torch.nn.functional.normalize(torch.randn(Ng, D, device='cuda', generator=g), dim=1)
    G[5] = G[3]; G[11] = G[3]                       # exact ties: index order must decide
    Q[0] = G[3]
    gid = torch.full((Ng,), -1, device='cuda', dtype=torch.int32); qid = torch.full((Nq,), -1, device='cuda', dtype=torch.int32)
    gid[torch.randint(0, Ng, (max(2, Ng // 50),), device='cuda', generator=g)] = 7; gid[3] = -1; gid[5] = -1; gid[11] = -1
    qid[Nq - 1] = 7
    index = GalleryIndex(G, normalized=True, img_ids=gid)
    kk = min(k, Ng)
    i_s, s_s = index.topk(Q, k=kk, normalized=True, query_img_ids=qid, stream=True)
    i_b, s_b = index.topk(Q, k=kk, normalized=True, query_img_ids=qid, stream=False)
    assert torch.equal(i_s, i_b)
    assert torch.equal(s_s, s_b)                    # same arithmetic for the fp32 score, bit for bit
    if kk >= 3:
        assert i_s[0, :3].tolist() == [3, 5, 11]
    sim = Q.double() @ G.double().t()
    sim = sim.masked_fill((qid.view(-1, 1) >= 0) & (qid.view(-1, 1) == gid.view(1, -1)), -1e9)
    ref = torch.argsort(sim.float(), dim=1, descending=True, stable=True)[:, :kk]
    for qi, r in (ref != i_s.long()).nonzero().tolist():       # only fp32-rounding near-ties may differ from the f64 order
        assert abs(float(sim[qi, int(ref[qi, r])] - sim[qi, int(i_s[qi, r])])) < 2e-7
    # default dispatch takes the streaming form for these shapes
    i_d, _ = index.topk(Q, k=kk, normalized=True, query_img_ids=qid)
    assert torch.equal(i_d, i_s)


def test_cosine_topk_one_query_fused_merge_repeated_calls(ops):
    """One query at a time at the full gallery size: the last-arriving workgroup merges the partial lists inside the scan launch (r04).
    Forty back-to-back calls on one index (the arrival counter must come back to zero every time), k changing in between, results equal
    to the batched pipeline's bit for bit."""
    from prcv2025reid_amd.retrieval import GalleryIndex
    g = torch.Generator(device='cuda').manual_seed(99)
    Ng, D = 200000, 512
    G = torch.nn.functional.normalize(torch.randn(Ng, D, device='cuda', generator=g), dim=1)
    Q = torch.nn.functional.normalize(torch.randn(40, D, device='cuda', generator=g), dim=1)
    G[77] = G[12345]; Q[3] = G[12345]
    from prcv2025reid_amd import _lib
    index = GalleryIndex(G, normalized=True)
    want_i, want_s = index.topk(Q, k=10, normalized=True, stream=False)
    _lib.check(_lib.lib().reid_set_knob(b'STREAM_FUSE', 1))                         # (off by default: no faster than the merge launch)
    try:
        _fused_merge_checks(index, Q, want_i, want_s)
    finally:
        _lib.check(_lib.lib().reid_set_knob(b'STREAM_FUSE', -1))


def _fused_merge_checks(index, Q, want_i, want_s):
    got = [index.topk(Q[i:i + 1], k=10, normalized=True) for i in range(40)]          # no synchronisation in between
    torch.cuda.synchronize()
    for i, (gi, gs) in enumerate(got):
        assert torch.equal(gi[0], want_i[i]) and torch.equal(gs[0], want_s[i]), i
    assert got[3][0][0, :2].tolist() == [77, 12345]
    w5_i, w5_s = index.topk(Q[:8], k=5, normalized=True, stream=False)
    for i in range(8):
        gi, gs = index.topk(Q[i:i + 1], k=5, normalized=True)
        assert torch.equal(gi[0], w5_i[i]) and torch.equal(gs[0], w5_s[i])
    gi, gs = index.topk(Q[:1], k=10, normalized=True)
    assert torch.equal(gi[0], want_i[0])
    assert int(index._ws_stream.view(torch.int32)[-64:].abs().sum()) == 0             # counter left at zero


def test_small_head_kernels(ops):
    from oracle import reid_oracle as O
    from prcv2025reid_amd import head as H
    g = torch.Generator(device='cuda').manual_seed(11)
    B, M, D, heads = 9, 5, 512, 8
    # small attention vs the oracle's attention_core with a key-padding mask
    qkv = torch.randn(B * M, 3 * D, device='cuda', generator=g)
    km = (torch.rand(B, M, device='cuda', generator=g) > 0.4).to(torch.uint8); km[:, 0] = 1
    qc = qkv.cpu().requires_grad_(True)
    q, k, v = [t.view(B, M, D) for t in qc.split(D, dim=1)]
    add = torch.zeros(B, 1, 1, M).masked_fill(~km.cpu().bool().view(B, 1, 1, M), float('-inf'))
    ref = O.attention_core(q, k, v, heads, add).reshape(B * M, D)
    qg = qkv.clone().requires_grad_(True)
    out = H.SmallAttnFn.apply(qg, km, B, M, heads)
    assert rel_err(out.detach().cpu(), ref.detach()) < 1e-5
    w = torch.randn(B * M, D, generator=torch.Generator().manual_seed(1))
    (ref * w).sum().backward(); (out * w.cuda()).sum().backward()
    assert rel_err(qg.grad.cpu(), qc.grad) < 1e-4
    # masked mean, activations, add, layernorm, linear
    x = torch.randn(B, M, D, device='cuda', generator=g).requires_grad_(True)
    mk = km.float()
    mm = H.MaskedMeanFn.apply(x, mk)
    refmm = (x.detach() * mk.unsqueeze(-1)).sum(1) / mk.sum(1, keepdim=True).clamp(min=1)
    assert rel_err(mm.detach(), refmm) < 1e-6
    mm.sum().backward()
    assert rel_err(x.grad, (mk / mk.sum(1, keepdim=True).clamp(min=1)).unsqueeze(-1).expand(B, M, D)) < 1e-6
    # the long form: one "sample" of 323 rows (global mean of the valid rows, model.py:141-149), D not a multiple of 64
    xl = torch.randn(1, 323, 200, device='cuda', generator=g).requires_grad_(True)
    ml = (torch.rand(1, 323, device='cuda', generator=g) > 0.5).float()
    mml = H.MaskedMeanFn.apply(xl, ml)
    assert rel_err(mml.detach(), (xl.detach() * ml.unsqueeze(-1)).sum(1) / ml.sum(1, keepdim=True)) < 1e-6
    (mml * 2.0).sum().backward()
    assert rel_err(xl.grad, (2.0 * ml / ml.sum(1, keepdim=True)).unsqueeze(-1).expand(1, 323, 200)) < 1e-6
    for kind, fn in (('relu', torch.relu), ('gelu', torch.nn.functional.gelu)):
        a = torch.randn(300, 512, device='cuda', generator=g).requires_grad_(True)
        b2 = a.detach().clone().requires_grad_(True)
        y = H.ActFn.apply(a, kind); yr = fn(b2)
        assert rel_err(y.detach(), yr.detach()) < 2e-6
        y.sum().backward(); yr.sum().backward()
        assert rel_err(a.grad, b2.grad) < 2e-6
    W = torch.randn(700, 512, device='cuda', generator=g).requires_grad_(True); bb = torch.randn(700, device='cuda', generator=g).requires_grad_(True)
    xin = torch.randn(B, M, 512, device='cuda', generator=g).requires_grad_(True)
    y = H.LinearNdF32Fn.apply(xin, W, bb)
    W2, b3, x2 = [t.detach().clone().requires_grad_(True) for t in (W, bb, xin)]
    yr = torch.nn.functional.linear(x2, W2, b3)
    assert rel_err(y.detach(), yr.detach()) < 1e-5
    wgt = torch.randn_like(yr)
    (y * wgt).sum().backward(); (yr * wgt).sum().backward()
    assert rel_err(xin.grad, x2.grad) < 1e-5 and rel_err(W.grad, W2.grad) < 1e-5 and rel_err(bb.grad, b3.grad) < 1e-5
    lw = (1 + 0.1 * torch.randn(512, device='cuda', generator=g)).requires_grad_(True); lb = torch.randn(512, device='cuda', generator=g).requires_grad_(True)
    x3 = torch.randn(B * M, 512, device='cuda', generator=g).requires_grad_(True)
    y = H.LayerNormF32Fn.apply(x3, lw, lb, 1e-5)
    lw2, lb2, x4 = [t.detach().clone().requires_grad_(True) for t in (lw, lb, x3)]
    yr = torch.nn.functional.layer_norm(x4, (512,), lw2, lb2, 1e-5)
    assert rel_err(y.detach(), yr.detach()) < 1e-5
    w3 = torch.randn(B * M, 512, device='cuda', generator=g)
    (y * w3).sum().backward(); (yr * w3).sum().backward()
    assert rel_err(x3.grad, x4.grad) < 1e-4 and rel_err(lw.grad, lw2.grad) < 1e-4 and rel_err(lb.grad, lb2.grad) < 1e-4
    t = torch.tensor([1.0, float('nan'), float('inf'), -float('inf')], device='cuda')
    assert H.NanToNumFn.apply(t).tolist() == [1.0, 0.0, 1e4, -1e4]


@pytest.mark.parametrize('n_other', [7, 2300])
def test_cosine_topk_overflow_fallback_is_exact(ops, n_other):
    """Thousands of near-duplicates overflow the candidate list: the query is flagged and the exact fp32 pass resolves it
    (small problems run that pass without reading the flags back; large ones gather the flagged queries on the host side)."""
    from prcv2025reid_amd.retrieval import GalleryIndex
    g = torch.Generator(device='cuda').manual_seed(3)
    D, Ng, k = 512, 30000, 10
    base = torch.nn.functional.normalize(torch.randn(1, D, device='cuda', generator=g), dim=1)
    G = torch.nn.functional.normalize(torch.randn(Ng, D, device='cuda', generator=g), dim=1)
    G[10000:20000] = torch.nn.functional.normalize(base + 1e-3 * torch.randn(10000, D, device='cuda', generator=g), dim=1)
    Q = torch.cat([base, torch.nn.functional.normalize(torch.randn(n_other, D, device='cuda', generator=g), dim=1)])
    idx, sc = GalleryIndex(G, normalized=True).topk(Q, k=k, normalized=True)
    ref = torch.argsort((Q.double() @ G.double().t()).float(), dim=1, descending=True, stable=True)[:, :k]
    sim = Q.double() @ G.double().t()
    for qi, r in (ref != idx.long()).nonzero().tolist():
        a, b = int(ref[qi, r]), int(idx[qi, r])
        assert abs(float(sim[qi, a] - sim[qi, b])) < 2e-7, (qi, r, a, b)


def test_cosine_topk_large_call_resolves_every_overflow_without_readback(ops):
    """Large calls (> 2^26 scores) resolve candidate-list overflows on the device without reading the flags back -- ANY number of them
    (r03 resolved at most 256 per call and left the rest marked -2): 300 flagged queries at the default settings come back exact, the
    chunked form (scratch budget below one call's worth) gives the same lists, and cmc_from_topk refuses an unresolved marker."""
    from prcv2025reid_amd.retrieval import GalleryIndex, cmc_from_topk
    g = torch.Generator(device='cuda').manual_seed(5)
    D, Ng, k, n_over = 512, 30000, 10, 300
    base = torch.nn.functional.normalize(torch.randn(1, D, device='cuda', generator=g), dim=1)
    G = torch.nn.functional.normalize(torch.randn(Ng, D, device='cuda', generator=g), dim=1)
    G[10000:20000] = torch.nn.functional.normalize(base + 1e-3 * torch.randn(10000, D, device='cuda', generator=g), dim=1)
    over = torch.nn.functional.normalize(base + 1e-4 * torch.randn(n_over, D, device='cuda', generator=g), dim=1)
    Q = torch.cat([over[:150], torch.nn.functional.normalize(torch.randn(2100, D, device='cuda', generator=g), dim=1), over[150:]])
    sim = Q.double() @ G.double().t()
    ref = torch.argsort(sim.float(), dim=1, descending=True, stable=True)[:, :k]

    def check(idx):
        assert int((idx < -1).sum()) == 0
        for q, r in (ref != idx.long()).nonzero().tolist():
            a, b = int(ref[q, r]), int(idx[q, r])
            assert abs(float(sim[q, a] - sim[q, b])) < 2e-7, (q, r, a, b)

    # how many queries overflow: the raw pipeline without the exact pass
    Qb = ops.to_t16(Q)
    ws = torch.empty(ops.topk_ws_bytes(Q.shape[0], Ng, k), dtype=torch.uint8, device='cuda')
    raw_i = torch.empty(Q.shape[0], k, dtype=torch.int32, device='cuda'); raw_s = torch.empty(Q.shape[0], k, device='cuda')
    ops.cosine_topk(Qb, ops.to_t16(G), Q, G, k, ws, raw_i, raw_s)
    n_flagged = int((raw_i[:, 0] == -2).sum())
    assert n_flagged >= n_over > 256
    with pytest.raises(ValueError):
        cmc_from_topk(raw_i, torch.zeros(Q.shape[0], dtype=torch.long), torch.zeros(Ng, dtype=torch.long))

    index = GalleryIndex(G, normalized=True)
    idx, sc = index.topk(Q, k=k, normalized=True)
    assert int(index._slots[0]) == n_flagged
    check(idx)
    assert float((sc.double() - sim.gather(1, idx.long())).abs().max()) < 2e-7
    index.exact_scratch_bytes = 1000 * Ng * 4                        # chunks of 1000 / 1000 / 400 queries: each below 2^26 scores, i.e. the
    idx_c, sc_c = index.topk(Q, k=k, normalized=True)               # small-problem exact pass per chunk -- same lists
    check(idx_c)
    assert torch.equal(idx_c, idx)
    index.exact_scratch_bytes = 2300 * Ng * 4                        # 2300 + 100 queries: one large chunk, one small
    idx_d, _ = index.topk(Q, k=k, normalized=True)
    assert torch.equal(idx_d, idx)
    assert 'R@1' in cmc_from_topk(idx, torch.zeros(Q.shape[0], dtype=torch.long), torch.zeros(Ng, dtype=torch.long))


def test_sharded_gallery_single_process(ops):
    """ShardedGalleryIndex with one rank (no process group) == GalleryIndex on the whole gallery; offsets applied."""
    from prcv2025reid_amd.parallel import ShardedGalleryIndex, merge_topk
    from prcv2025reid_amd.retrieval import GalleryIndex
    g = torch.Generator(device='cuda').manual_seed(4)
    Q = torch.randn(33, 512, device='cuda', generator=g); G = torch.randn(3000, 512, device='cuda', generator=g)
    want_i, want_s = GalleryIndex(G).topk(Q, k=10)
    parts_i, parts_s = [], []
    for a, b in ((0, 1000), (1000, 1900), (1900, 3000)):
        sh = ShardedGalleryIndex(G[a:b], a)
        i, s = sh.topk(Q, k=10)
        parts_i.append(i); parts_s.append(s)
    gi, gs = merge_topk(torch.stack(parts_i), torch.stack(parts_s), 10)
    assert torch.equal(gi, want_i.long()) and torch.allclose(gs, want_s)


def test_attention_query_tile_limit(ops):
    """q_tiles=1: the first 32 query rows equal the full kernel; with a cotangent that is zero elsewhere the backward is identical."""
    g = torch.Generator(device='cuda').manual_seed(6)
    n_seq, S, heads = 5, 197, 4
    d = heads * 64
    qkv = (torch.randn(n_seq * S, 3 * d, device='cuda', generator=g)).to(T16())
    o_full = torch.empty(n_seq * S, d, device='cuda', dtype=T16()); lse_full = torch.empty(n_seq, heads, S, device='cuda')
    ops.attn_fwd(qkv, o_full, lse_full, n_seq, S, heads)
    o_lim = torch.zeros_like(o_full); lse_lim = torch.zeros_like(lse_full)
    ops.attn_fwd(qkv, o_lim, lse_lim, n_seq, S, heads, q_tiles=1)
    rows = (torch.arange(n_seq, device='cuda').view(-1, 1) * S + torch.arange(32, device='cuda').view(1, -1)).flatten()
    assert torch.equal(o_lim[rows], o_full[rows]) and torch.equal(lse_lim[:, :, :32], lse_full[:, :, :32])
    assert float(o_lim.float().abs().sum()) == float(o_full[rows].float().abs().sum())        # nothing else was written
    do = torch.zeros(n_seq * S, d, device='cuda', dtype=T16())
    do[torch.arange(n_seq, device='cuda') * S] = torch.randn(n_seq, d, device='cuda', generator=g).to(T16())
    dq_full = torch.empty_like(qkv); dq_lim = torch.full_like(qkv, 7.0)
    delta = torch.empty(n_seq, heads, S, device='cuda'); delta2 = torch.empty_like(delta)
    ops.attn_bwd(qkv, o_full, do, lse_full, dq_full, delta, n_seq, S, heads)
    ops.attn_bwd(qkv, o_lim, do, lse_lim, dq_lim, delta2, n_seq, S, heads, q_tiles=1)
    assert torch.equal(dq_lim, dq_full)


def test_f16_conversion_saturates_finite_overflow():
    """f16 flavor: a finite value beyond IEEE half's range is stored as +-65504 by every kernel (MODE.FP16_OVFL, common.h
    REID_T16_ENTER), infinities and NaNs pass through -- activations far beyond 65 504 give finite operands for the next GEMM
    instead of infinities.  (bf16 has fp32's exponent range: nothing to saturate.)"""
    from prcv2025reid_amd import ops as o, _lib
    _lib.set_flavor('f16')
    try:
        src = torch.tensor([1.0, 65504.0, 65520.0, 7.0e4, 1.0e6, -1.0e6, 3.0e38, float('inf'), float('-inf'), float('nan'), -2.5, 1e-9],
                           device='cuda')
        dst = torch.empty(src.shape, device='cuda', dtype=torch.float16)
        o.cast_f32_bf16(src, dst)
        d = dst.float().cpu()
        assert d[:7].tolist() == [1.0, 65504.0, 65504.0, 65504.0, 65504.0, -65504.0, 65504.0], d
        assert math.isinf(float(d[7])) and float(d[7]) > 0 and math.isinf(float(d[8])) and float(d[8]) < 0 and math.isnan(float(d[9]))
        assert float(d[10]) == -2.5
        # GEMM epilogues (both tiles), LayerNorm output and the GELU epilogue with |values| >> 65504: finite everywhere
        for M, N, K in ((256, 256, 128), (4096, 1536, 128)):
            A = torch.full((M, K), 300.0, device='cuda', dtype=torch.float16)
            A[1::2] = -300.0
            B = torch.full((N, K), 300.0, device='cuda', dtype=torch.float16)
            C = torch.empty(M, N, device='cuda', dtype=torch.float16)
            G = torch.empty(M, N, device='cuda', dtype=torch.float16)
            o.gemm(A, B, C)                                   # 128 * 9e4 = 1.15e7
            assert torch.isfinite(C).all()
            assert float(C[0, 0]) == 65504.0 and float(C[1, 0]) == -65504.0
            o.gemm(A, B, G, act='gelu', C2=C)                 # gelu(1.15e7) = 1.15e7 -> 65504; gelu(-1.15e7) = -0.0
            assert torch.isfinite(G).all() and float(G[0, 0]) == 65504.0 and float(G[1, 0]) == 0.0
        x = torch.randn(64, 768, device='cuda') * 1e6
        y = torch.empty(64, 768, device='cuda', dtype=torch.float16)
        gamma = torch.full((768,), 5.0e4, device='cuda'); beta = torch.zeros(768, device='cuda')
        o.layernorm_fwd(x, gamma, beta, y_bf16=y)             # normalised values up to ~4 * 5e4 = 2e5
        assert torch.isfinite(y).all() and float(y.abs().max()) == 65504.0
    finally:
        _lib.set_flavor('bf16')


@pytest.mark.parametrize('r,G,N,K', [(8, 1, 768, 768), (4, 3, 384, 128), (16, 1, 256, 3072), (32, 1, 256, 768), (24, 3, 384, 128), (64, 1, 128, 128)])
def test_merge_lora_table(ops, r, G, N, K):
    """reid_merge_lora_table: W_eff[mu] = W + s B_mu A_mu per modality (mer_lora.py:80-99 with the adapter folded into the weight),
    rounded once to 16 bits; the transposed stack is the EXACT transpose.  Ranks above 16 (nmod * r > 64: r04) stage one modality's adapter
    rows at a time; the reference accepts any rank."""
    g = torch.Generator(device='cuda').manual_seed(r + G + N)
    nmod = 4
    Rp = ((nmod * r + 31) // 32) * 32
    W = torch.randn(N, K, device='cuda', generator=g) * 0.03
    arena = torch.randn(G * Rp * K + N * Rp + 64, device='cuda', generator=g) * 0.2
    offA, offB = 32, 32 + G * Rp * K
    A = arena[offA:offA + G * Rp * K].view(G * Rp, K); B = arena[offB:offB + N * Rp].view(N, Rp)
    weff = torch.zeros(2 * nmod * N * K + 128, device='cuda', dtype=T16())
    oE, oET = 64, 64 + nmod * N * K
    table = torch.tensor([[W.data_ptr(), offA, offB, oE, oET, N, K, G]], dtype=torch.int64, device='cuda')
    s = 1.0 / r
    ops.merge_lora_table(table, 1, (N // 64) * (K // 64), arena, weff, Rp, r, nmod, s)
    E = weff[oE:oE + nmod * N * K].view(nmod, N, K); ET = weff[oET:oET + nmod * N * K].view(nmod, K, N)
    assert torch.equal(ET, E.transpose(1, 2))
    assert float(weff[:oE].float().abs().max()) == 0 and float(weff[oET + nmod * N * K:].float().abs().max()) == 0
    n_g = N // G
    for mu in range(nmod):
        ref = W.clone()
        for gi in range(G):
            ref[gi * n_g:(gi + 1) * n_g] += s * (B[gi * n_g:(gi + 1) * n_g, mu * r:(mu + 1) * r].double() @
                                                 A[gi * Rp + mu * r: gi * Rp + (mu + 1) * r].double()).float()
        got = E[mu].float()
        # one rounding of the fp32 sum: within half a 16-bit ulp of the reference (+ fp32 summation-order noise)
        ulp = 2.0 ** (-8 if T16() == torch.bfloat16 else -11)
        assert float(((got - ref).abs() / ref.abs().clamp_min(1e-3)).max()) <= ulp * 1.01 + 1e-5, mu


@pytest.mark.parametrize('N,K,epi', [(768, 768, 'res32'), (2304, 768, 'plain16'), (3072, 768, 'gelu2d'), (768, 3072, 'res32'),
                                     (3072, 768, 'mulaux'), (96, 768, 'generic')])
def test_gemm_row_groups(ops, N, K, epi):
    """Row-group form of reid_mer_gemm: rows of group g multiply weight matrix row_group_b[g] of a stack; tiles never straddle
    groups; ragged group sizes (multiples of 197 rows), 128 x 128 tile and both ping-pong tiles, every lean epilogue + the generic one."""
    g = torch.Generator(device='cuda').manual_seed(N + K + len(epi))
    S = 197
    imgs, mus = [10, 14, 9, 7], [2, 0, 3, 1]
    M = sum(imgs) * S
    ends, e = [], 0
    for n in imgs:
        e += n * S; ends.append(e)
    A = bf(torch.randn(M, K, device='cuda', generator=g))
    Wst = bf(torch.randn(4, N, K, device='cuda', generator=g) * 0.03)
    bias = torch.randn(N, device='cuda', generator=g) * 0.1
    ref = torch.empty(M, N, device='cuda')
    lo = 0
    for hi, mu in zip(ends, mus):
        ref[lo:hi] = A[lo:hi].float() @ Wst[mu].float().t() + bias
        lo = hi
    rg = (ends, mus)
    if epi == 'res32':
        R = torch.randn(M, N, device='cuda', generator=g)
        rs = (torch.rand(sum(imgs), device='cuda', generator=g) > 0.3).float() / 0.7
        C = torch.empty(M, N, device='cuda')
        ops.gemm(A, Wst, C, bias=bias, R=R, row_scale=rs, rows_per_img=S, row_groups=rg)
        want = R + ref * rs.repeat_interleave(S).view(-1, 1)
        assert rel_err(C, want) < 3e-5
    elif epi == 'plain16' or epi == 'generic':
        C = torch.empty(M, N, device='cuda', dtype=T16())
        ops.gemm(A, Wst, C, bias=bias, row_groups=rg)
        assert rel_err(C.float(), ref) < 1e-2
        Cf = torch.empty(M, N, device='cuda')
        ops.gemm(A, Wst, Cf, bias=bias, row_groups=rg)
        assert rel_err(Cf, ref) < 3e-5
    elif epi == 'gelu2d':
        C = torch.empty(M, N, device='cuda', dtype=T16()); C2 = torch.empty(M, N, device='cuda', dtype=T16())
        ops.gemm(A, Wst, C, bias=bias, act='gelu_dsave', C2=C2, row_groups=rg)
        x = ref.double()
        cdf = 0.5 * (1 + torch.erf(x / math.sqrt(2))); pdf = torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
        assert rel_err(C.float(), (x * cdf).float()) < 1e-2
        assert rel_err(C2.float(), (cdf + x * pdf).float()) < 1e-2
    else:
        aux = bf(torch.randn(M, N, device='cuda', generator=g))
        C = torch.empty(M, N, device='cuda', dtype=T16())
        ops.gemm(A, Wst, C, act='mul_aux', aux=aux, row_groups=rg)
        assert rel_err(C.float(), (ref - bias) * aux.float()) < 1e-2
    # rows of one group must not see another group's matrix: a single-group call on a slice gives the same bits
    lo, hi, mu = ends[0], ends[1], mus[1]
    Cs = torch.empty(hi - lo, N, device='cuda')
    ops.gemm(A[lo:hi], Wst[mu], Cs, bias=bias)
    Cg = torch.empty(M, N, device='cuda')
    ops.gemm(A, Wst, Cg, bias=bias, row_groups=rg)
    assert rel_err(Cg[lo:hi], Cs) < 1e-6
