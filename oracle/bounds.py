"""Derived parity bounds for the 16-bit operand flavors.  TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke()
and bench.py's parity leg, like the rest of oracle/ -- never by the product path.

Why not "the last measurement x 1.5": a gate sized to a run cannot tell a regression from the floor.  The bounds here follow from
the operand format and the structure of the encoders, and the measured errors are reported next to them by every test.

Model of the error of an MFMA GEMM with 16-bit operands and fp32 accumulation: each operand element carries an independent
relative rounding error, uniform in +-u (u = 2^-8 for bf16: 8 significant bits; 2^-11 for IEEE half), rms u / sqrt(3); a product of
two rounded elements has rms relative error sqrt(2/3) u; the sum over K of products of incoherent signs has the SAME relative rms
error as one product (signal and error both grow like sqrt(K)):

    eps_gemm(flavor) = sqrt(2/3) * u                      bf16 3.2e-3, f16 4.0e-4

A residual branch chains 2 (MLP: fc1, fc2) or 4 (attention: q|k|v, QK^T, PV, out) such products: eps_branch <= 2 eps_gemm.  The
residual stream is the incoherent sum of its 2L branch outputs and its error the incoherent sum of their errors, so the relative
L2 error of an encoder output stays at the branch level whatever the depth; a factor 2 covers the propagation of earlier errors
through later blocks (each block's Jacobian is close to the identity plus one branch):

    rel_l2(encoder feature) <= 4 * eps_gemm               bf16 1.28e-2, f16 1.6e-3      (measured on MI355X: 6.0-7.2e-3, 8e-4)

On a unit-normalised D-vector an error of relative L2 size r is r / sqrt(D) rms per element, and the largest of n = B * D
independent Gaussian elements is sqrt(2 ln n) sigma:

    unit_maxabs(flavor, B, D) = 4 * eps_gemm / sqrt(D) * sqrt(2 ln(B D))

For bf16 at B = 64, D = 512 that is 2.6e-3 (measured 1.1-1.5e-3): the bf16 flavor cannot meet north_star's 1e-3 on embeddings, the
f16 flavor (3.3e-4 by this formula) does and is held to the 1e-3 as written.

The BN-neck (batch-statistics BatchNorm over B samples, then 8 * L2-normalise) is NOT norm-preserving: it subtracts the batch mean
-- the sample-independent 73-96 % of a random-init feature -- and divides by the per-feature spread, so it amplifies a perturbation of
the fused feature by a factor that depends on the batch (17-27x on the B = 6 fixtures with one fused modality, ~3x at B = 64).
``head_amplification`` measures that factor on the oracle's head for the batch at hand (finite perturbation of the encoder outputs,
fp64); the bound on the fused feature / on bn_features / 8 is the encoder-output bound times it.
"""
import math

import torch

UNIT_ROUNDOFF = {'bf16': 2.0 ** -8, 'f16': 2.0 ** -11}
NORTH_STAR_TOL = 1e-3


def eps_gemm(flavor: str) -> float:
    return math.sqrt(2.0 / 3.0) * UNIT_ROUNDOFF[flavor]


def rel_l2_bound(flavor: str) -> float:
    """Relative L2 error bound of an encoder output feature."""
    return 4.0 * eps_gemm(flavor)


def unit_maxabs(flavor: str, B: int, D: int, floor: bool = True) -> float:
    """Bound on max|delta| between unit-normalised [B, D] encoder features of the HIP path and the fp32 reference.  With ``floor``
    the f16 flavor is held to north_star's 1e-3 as written where the derived bound is tighter (it is for every D >= 64)."""
    n = max(2, B * D)
    b = rel_l2_bound(flavor) / math.sqrt(D) * math.sqrt(2.0 * math.log(n))
    return max(b, NORTH_STAR_TOL) if (floor and flavor == 'f16') else b


def head_bounds(flavor: str, B: int, D: int, k_fused: float, k_bn: float):
    """(encoder features, fused pre-BN feature, bn_features / 8) bounds on unit-normalised max|delta| for a batch whose head
    amplifies encoder-output perturbations by (k_fused, k_bn) (head_amplification)."""
    enc = unit_maxabs(flavor, B, D)
    drv = unit_maxabs(flavor, B, D, floor=False)
    return enc, max(enc, k_fused * drv), max(enc, k_bn * drv)


def head_amplification(raw, fmask, state, arch, training: bool, moddrop_keep=None, min_modalities: int = 1, trials: int = 3,
                       rel: float = 1e-3, seed: int = 0, labels=None, loss_kw=None):
    """(k_fused, k_bn): how much the head (SDM module, fusion, BN-neck; oracle/reid_oracle.py head()) amplifies a perturbation of
    the encoder outputs, as  max|delta(out)| / max_m max|delta(unit-normalised raw_m)|  with out = the unit-normalised fused
    pre-BN feature / bn_features / 8.  Random perturbations of relative L2 size ``rel`` per row and modality, fp64, worst of
    ``trials``; never below 1.  With ``labels`` a third value: {loss name: max |delta loss| / max|delta unit raw|} for the three
    losses of oracle compute_loss(**loss_kw) -- the sensitivity of each loss to the same perturbations."""
    from . import reid_oracle as O

    def dbl(v):
        return v.double() if torch.is_tensor(v) and v.dtype.is_floating_point else v

    st = {k: dbl(v) for k, v in state.items() if not k.startswith('clip_encoder.')}
    r0 = {m: torch.as_tensor(v).detach().double() for m, v in raw.items()}
    fm = {m: torch.as_tensor(v) for m, v in fmask.items()}
    unit = lambda t: torch.nn.functional.normalize(t, dim=1)
    with torch.no_grad():
        o0 = O.head(dict(r0), dict(fm), st, arch, training, moddrop_keep, min_modalities)
        g = torch.Generator().manual_seed(seed)
        kf = kb = 1.0
        kl = {}
        L0 = O.compute_loss(o0, labels, **(loss_kw or {})) if labels is not None else None
        for _ in range(trials):
            r1, din = {}, 0.0
            for m, x in r0.items():
                n = torch.randn(x.shape, generator=g, dtype=torch.float64)
                n = n / n.norm(dim=1, keepdim=True) * x.norm(dim=1, keepdim=True) * rel
                r1[m] = x + n
                din = max(din, float((unit(r1[m]) - unit(x)).abs().max()))
            o1 = O.head(r1, dict(fm), st, arch, training, moddrop_keep, min_modalities)
            kf = max(kf, float((unit(o1['features']) - unit(o0['features'])).abs().max()) / din)
            if 'bn_features' in o0:
                kb = max(kb, float((o1['bn_features'] - o0['bn_features']).abs().max()) / O.FEAT_SCALE / din)
            if L0 is not None:
                L1 = O.compute_loss(o1, labels, **(loss_kw or {}))
                for k in ('total_loss', 'ce_loss', 'sdm_loss'):
                    kl[k] = max(kl.get(k, 0.0), abs(float(L1[k]) - float(L0[k])) / din)
    return (kf, kb, kl) if labels is not None else (kf, kb)


def loss_bound(flavor: str, B: int, D: int, k_loss: float) -> float:
    """Bound on |delta loss|: 2 x (sensitivity of the loss to random encoder-output perturbations, head_amplification) x the derived
    encoder bound -- the factor covers the coherent part of operand rounding (weight rounding is the same for every sample, a
    random perturbation is not) -- and never below the encoder bound itself (f16: north_star's 1e-3 as written)."""
    return max(unit_maxabs(flavor, B, D), 2.0 * k_loss * unit_maxabs(flavor, B, D, floor=False))
