"""Explicit forward/backward executor of the two encoders on libreid_hip.so.

Vision (reference: CLIPUnifiedEncoder.encode_vision clip_backbone.py:254-286, MERTransformerBlock
:61-85, MERLinear mer_lora.py:80-99, MERMultiheadAttention :141-231, MERMLP :267-280, PatchEmbed
patch_embeds.py:45-76).  MI355X-first differences from the reference's execution (same arithmetic):

* ALL vision modalities of a batch go through the 12 blocks in ONE pass: valid images are packed
  [n_img, 197, 768] modality by modality.  The per-modality LoRA adapters live INSIDE the weights:
  once per optimizer step ``reid_merge_lora_table`` forms W_eff[mu] = W + (alpha/r) B_mu A_mu for every
  linear and modality (16-bit, both orientations: 1.4 GB of the 288 GB), and every MERLinear is ONE plain
  MFMA GEMM whose row tiles pick the matrix of their modality (row groups of reid_mer_gemm) -- no
  rank-r side computation on the critical path, no K extension.  What the adapter GRADIENTS need
  (T = x A^T in the forward, U = dY B in the backward, dA = U^T x, dB = dY^T T) runs on a second HIP
  stream from tensors the main stream produces anyway.
  4x fewer, 4x larger launches than the reference's one-encoder-pass-per-modality loop.
* q|k|v are one [768 -> 2304] GEMM; LayerNorm-1 is computed once (the reference evaluates it 3x).
* residual stream fp32, MFMA operands bf16, fp32 accumulate; GELU / residual / bias fused in epilogues.
* backward is hand-written: dX GEMMs reuse the same kernel with transposed packs, LoRA gradients are
  reduce-over-rows GEMMs written straight into one flat fp32 gradient arena (one RCCL bucket).

Text (reference: encode_text clip_backbone.py:288-313 -> HF CLIPTextModel): same kernels, causal +
key-padding attention, quick_gelu, EOS pooling; forward only (the tower carries no LoRA and is frozen
by train.py:1418-1425).
"""
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import os

import torch

from . import _lib, ops

LIN = ('qkv', 'out', 'fc1', 'fc2')


def round_up(a, b):
    return (a + b - 1) // b * b


_EXP_SKIP_U = os.environ.get('REID_EXP_SKIP_U', '0') == '1'
_EXP_SKIP_DA = os.environ.get('REID_EXP_SKIP_DA', '0') == '1'


class LoraLayout:
    """Where each (layer, linear) adapter set lives in the flat fp32 arena and in the bf16 pack arena."""

    def __init__(self, arch):
        self.vmods = [m for m in arch['modalities'] if m != 'text']
        self.nmod = len(self.vmods)
        self.r = arch['lora_rank']
        self.Rp = round_up(self.nmod * self.r, 32)
        self.d = arch['vision_hidden_dim']; self.ff = arch['vision_mlp_dim']
        self.L = arch['vision_layers']
        d, ff, Rp = self.d, self.ff, self.Rp
        self.dims = {'qkv': (3, d, 3 * d), 'out': (1, d, d), 'fc1': (1, d, ff), 'fc2': (1, ff, d)}   # G, K(in), N(out)
        off = 0; poff = 0; woff = 0
        self.ent = {}
        for l in range(self.L):
            for nm in LIN:
                G, K, N = self.dims[nm]
                e = dict(G=G, K=K, N=N)
                e['A'] = (off, (G * Rp, K)); off += G * Rp * K
                e['B'] = (off, (N, Rp)); off += N * Rp
                # bf16 packs: A, A^T, B, B^T
                e['pA'] = poff; poff += G * Rp * K
                e['pAT'] = poff; poff += G * Rp * K
                e['pB'] = poff; poff += N * Rp
                e['pBT'] = poff; poff += N * Rp
                # merged weights W + (alpha/r) B_mu A_mu per modality: [nmod, N, K] and the transposes [nmod, K, N]
                e['wE'] = woff; woff += self.nmod * N * K
                e['wET'] = woff; woff += self.nmod * N * K
                self.ent[(l, nm)] = e
        self.size = off
        self.pack_size = poff
        self.weff_size = woff

    def table(self) -> torch.Tensor:
        rows = []
        for (l, nm), e in self.ent.items():
            (oa, (ra, ca)), (ob, (rb, cb)) = e['A'], e['B']
            rows.append([oa, ra, ca, e['pA'], e['pAT']])
            rows.append([ob, rb, cb, e['pB'], e['pBT']])
        return torch.tensor(rows, dtype=torch.int64)

    # reference key <-> arena slice ------------------------------------------------------------
    def ref_slices(self, l: int, ref_lin: str, modality: str):
        """(arena offset/shape info) of lora_A [r, K] and lora_B [N, r] of one reference adapter."""
        nm, g = {'attn.q_proj': ('qkv', 0), 'attn.k_proj': ('qkv', 1), 'attn.v_proj': ('qkv', 2),
                 'attn.out_proj': ('out', 0), 'mlp.fc1': ('fc1', 0), 'mlp.fc2': ('fc2', 0)}[ref_lin]
        e = self.ent[(l, nm)]
        mu = self.vmods.index(modality)
        n_out = e['N'] // e['G']
        return e, g, mu, n_out

    def view_A(self, arena, l, nm):
        o, shp = self.ent[(l, nm)]['A']
        return arena[o:o + shp[0] * shp[1]].view(shp)

    def view_B(self, arena, l, nm):
        o, shp = self.ent[(l, nm)]['B']
        return arena[o:o + shp[0] * shp[1]].view(shp)

    def weff(self, arena, l, nm, transposed=False):
        """Merged weight stack of one linear: [nmod, N, K] (forward operand) or [nmod, K, N] (dX operand)."""
        e = self.ent[(l, nm)]
        N, K = e['N'], e['K']
        o = e['wET'] if transposed else e['wE']
        return arena[o:o + self.nmod * N * K].view((self.nmod, K, N) if transposed else (self.nmod, N, K))

    def pk(self, pack, l, nm, which):
        e = self.ent[(l, nm)]
        G, K, N, Rp = e['G'], e['K'], e['N'], self.Rp
        if which == 'A':
            return pack[e['pA']:e['pA'] + G * Rp * K].view(G * Rp, K)
        if which == 'AT':
            return pack[e['pAT']:e['pAT'] + G * Rp * K].view(K, G * Rp)
        if which == 'B':
            return pack[e['pB']:e['pB'] + N * Rp].view(N, Rp)
        return pack[e['pBT']:e['pBT'] + N * Rp].view(Rp, N)


class Engine:
    """Owns packed bf16 weights and runs the encoders.  ``P`` maps reference names -> fp32 tensors."""

    def __init__(self, arch: dict, P: Dict[str, torch.Tensor], lora_arena: torch.Tensor, device):
        self.arch = arch
        self.P = P
        self.lora_arena = lora_arena
        self.dev = device
        self.lay = LoraLayout(arch)
        self.S = (arch['image_size'] // arch['patch_size']) ** 2 + 1
        self.scaling = arch['lora_alpha'] / arch['lora_rank']
        # f16 operands need the backward pass scaled into half's range (gradients of a mean loss are ~1e-6..1e-3, f16
        # normals start at 6e-5): the incoming cotangent is scaled by a power of two chosen ON DEVICE so that its largest
        # entry is in [256, 512) (64x head-room below 65504), and the fp32 LoRA gradients are unscaled at the end.
        # bf16 has fp32's exponent range and needs none of this.
        self.flavor = _lib.flavor()
        self.loss_scaling = self.flavor == 'f16'
        # The residual-stream gradient of the vision tower (dx between the LayerNorm backward kernels) in IEEE half: those kernels
        # are HBM-bound and move 12 instead of 16 bytes per element.  Needs the same scaling in either flavor.
        self.dx_half = os.environ.get('REID_DX_HALF', '1') != '0'
        self._dense_ver = None
        self._lora_ver = None
        self._lora_pack = None
        self._weff = None
        self._merge_table = None
        self._table = None
        self._side = None
        self._tside = None
        self._consts = {}
        self._rng = torch.Generator(device=device)
        self._rng.manual_seed(777)
        self.pending_drop_scales = None
        self.text_backward_ready = True
        self._text_packed = None
        self.overlap_tn = os.environ.get('REID_TN_STREAM', '1') != '0'
        self.cls_prune = os.environ.get('REID_CLS_PRUNE', '1') != '0'
        # experiment switches, read once (the per-layer code paths test attributes, not the environment)
        self.add_ln = os.environ.get('REID_ADD_LN', '1') != '0'
        self.lora_down_defer = os.environ.get('REID_LORA_DOWN_DEFER', '1') != '0'
        self.lora_fused = os.environ.get('REID_LORA_FUSED', '1') != '0'
        self.lora_da_fused = os.environ.get('REID_LORA_DA_FUSED', '1') != '0'
        self.lora_fused_max_n = 768 if os.environ.get('REID_LORA_FUSED') == '768' else 1 << 30      # (A/B: only the 768-column linears)
        self.W = {}
        self.W32 = {}

    def _const(self, key, make):
        """Small index tensors that depend only on the batch layout: built once per layout and kept on the device (a
        host->device copy per step would also make the step impossible to capture in a HIP graph)."""
        t = self._consts.get(key)
        if t is None:
            if len(self._consts) > 64:
                self._consts.clear()
            t = make().to(self.dev)
            self._consts[key] = t
        return t

    def _text_stream(self):
        if self._tside is None:
            self._tside = torch.cuda.Stream(self.dev)
        return self._tside

    def _side_stream(self):
        """Stream of the adapter-gradient kernels: LOWEST priority, so its workgroups are dispatched only where the main stream has
        none pending (the partly filled last round of a GEMM, gaps between kernels) instead of taking compute units from it."""
        if self._side is None:
            prio = os.environ.get('REID_SIDE_PRIORITY')
            if prio is None:
                try:
                    least, _greatest = torch.cuda.Stream.priority_range()
                except Exception:
                    least = 0
                prio = least
            self._side = torch.cuda.Stream(self.dev, priority=int(prio))
        return self._side

    # ------------------------------------------------------------------------------- packing
    def _bf(self, t):
        return ops.to_bf16(t.detach())

    def _bft(self, t):
        return ops.to_bf16(t.detach().t().contiguous())

    def pack_dense(self):
        P, a, W = self.P, self.arch, {}
        ce = 'clip_encoder.'
        for l in range(a['vision_layers']):
            lp = f'{ce}vision_layers.{l}.'
            # fp32 masters of the four linears: the merge kernel reads them (the 16-bit operands are the merged stacks, pack_lora)
            self.W32[(l, 'qkv')] = torch.cat([P[lp + f'attn.{n}_proj.shared_linear.weight'].detach() for n in 'qkv'], 0).contiguous()
            W[('v', l, 'bqkv')] = torch.cat([P[lp + f'attn.{n}_proj.shared_linear.bias'].detach() for n in 'qkv'], 0).contiguous()
            for nm, ref in (('out', 'attn.out_proj'), ('fc1', 'mlp.fc1'), ('fc2', 'mlp.fc2')):
                self.W32[(l, nm)] = P[lp + ref + '.shared_linear.weight'].detach().contiguous()
        for m in self.lay.vmods:
            w = P[f'{ce}patch_embeds.{m}.proj.weight']
            W[('pe', m)] = self._bf(w.reshape(w.shape[0], -1))
        w = P[ce + 'vision_proj.weight']
        W['vproj'] = self._bf(w); W['vprojT'] = self._bft(w)
        tp = ce + 'clip_model.text_model.'
        for l in range(a['text_layers']):
            lp = f'{tp}encoder.layers.{l}.'
            wq = torch.cat([P[lp + f'self_attn.{n}_proj.weight'].detach() for n in 'qkv'], 0)
            W[('t', l, 'qkv')] = self._bf(wq)
            W[('t', l, 'bqkv')] = torch.cat([P[lp + f'self_attn.{n}_proj.bias'].detach() for n in 'qkv'], 0).contiguous()
            W[('t', l, 'out')] = self._bf(P[lp + 'self_attn.out_proj.weight'])
            W[('t', l, 'fc1')] = self._bf(P[lp + 'mlp.fc1.weight'])
            W[('t', l, 'fc2')] = self._bf(P[lp + 'mlp.fc2.weight'])
            if self.text_trains():                          # transposed packs for the dX products of the text backward
                W[('t', l, 'qkvT')] = self._bft(wq)
                W[('t', l, 'outT')] = self._bft(P[lp + 'self_attn.out_proj.weight'])
                W[('t', l, 'fc1T')] = self._bft(P[lp + 'mlp.fc1.weight'])
                W[('t', l, 'fc2T')] = self._bft(P[lp + 'mlp.fc2.weight'])
        W['tproj'] = self._bf(P[ce + 'text_proj.weight'])
        if self.text_trains():
            W['tprojT'] = self._bft(P[ce + 'text_proj.weight'])
        self.W = W
        self._merge_table = None                              # the fp32 master pointers may have moved

    def pack_lora(self):
        """16-bit copies of the adapters (A, A^T, B, B^T: operands of the gradient-side kernels) and the merged weight stacks
        W_eff[mu] = W + (alpha/r) B_mu A_mu (operands of every MERLinear GEMM): two launches per optimizer step."""
        lay = self.lay
        if self._lora_pack is None:
            self._lora_pack = torch.empty(lay.pack_size, dtype=_lib.t16(), device=self.dev)
            self._table = lay.table().to(self.dev)
        if self._weff is None:
            self._weff = torch.empty(lay.weff_size, dtype=_lib.t16(), device=self.dev)
        if self._merge_table is None:
            rows = []
            for (l, nm), e in lay.ent.items():
                w = self.W32[(l, nm)]
                assert w.shape == (e['N'], e['K']) and w.is_contiguous() and w.dtype == torch.float32
                rows.append([w.data_ptr(), e['A'][0], e['B'][0], e['wE'], e['wET'], e['N'], e['K'], e['G']])
            self._merge_table = torch.tensor(rows, dtype=torch.int64).to(self.dev)
            self._merge_tiles = max((e['N'] // 64) * (e['K'] // 64) for e in lay.ent.values())
        arena = self.lora_arena.detach()
        self.wait_packed()                                   # (a previous pack nobody consumed must not overlap writers of the arena)
        # Both launches go to a stream of their own, forked from the caller's: the merge (~1.7 GB of traffic) then runs beside the
        # patch embedding and the text tower instead of in front of them; the first vision block waits for it (wait_packed).
        main = torch.cuda.current_stream(self.dev)
        ps = self._pack_stream()
        ev = torch.cuda.Event(); ev.record(main); ps.wait_event(ev)
        if self._side is not None:
            # deferred T = x A^T launches of a forward whose backward never ran may still be reading the 16-bit adapter pack
            ev_s = torch.cuda.Event(); ev_s.record(self._side); ps.wait_event(ev_s)
        with torch.cuda.stream(ps):
            ops.pack_bf16_table(arena, self._lora_pack, self._table, self._table.shape[0])
            ops.merge_lora_table(self._merge_table, self._merge_table.shape[0], self._merge_tiles, arena, self._weff, lay.Rp, lay.r,
                                 lay.nmod, self.scaling)
            self._pack_event = torch.cuda.Event(); self._pack_event.record(ps)

    def wait_packed(self):
        """Make the current stream (and the adapter-gradient side stream) wait for the last pack_lora."""
        ev = getattr(self, '_pack_event', None)
        if ev is not None:
            torch.cuda.current_stream(self.dev).wait_event(ev)
            if self._side is not None:
                self._side.wait_event(ev)
            self._pack_event = None

    def _pack_stream(self):
        if getattr(self, '_pstream', None) is None:
            self._pstream = torch.cuda.Stream(self.dev)
        return self._pstream

    def refresh(self):
        dv = sum(p._version for k, p in self.P.items() if k.startswith('clip_encoder.') and p is not self.lora_arena)
        tt = self.text_trains()
        dense_changed = dv != self._dense_ver or tt != self._text_packed
        if dense_changed:
            self.pack_dense(); self._dense_ver = dv; self._text_packed = tt
        lv = self.lora_arena._version
        if lv != self._lora_ver or dense_changed:
            self.pack_lora(); self._lora_ver = lv

    # ------------------------------------------------------------------------------- vision forward
    def drop_path_scales(self, n_img: int, rate: float):
        """Per layer (s_attn, s_mlp), each f32 [n_img] = floor(keep + U) / keep with keep = 1 - rate * l / (L - 1)
        (DropPath, clip_backbone.py:137-141 with the per-block rates of :204); None where the rate is 0."""
        L = self.arch['vision_layers']
        keep = self._const(('dp_keep', L, float(rate)),
                           lambda: torch.tensor([1.0 - rate * l / max(1, L - 1) for l in range(L)], dtype=torch.float32))
        u = torch.rand(L, 2, n_img, device=self.dev, generator=self._rng)          # ONE draw for all 2L branches
        sc = torch.floor(u + keep.view(L, 1, 1)) / keep.view(L, 1, 1)
        return [(None, None) if rate * l / max(1, L - 1) <= 0.0 else (sc[l, 0], sc[l, 1]) for l in range(L)]

    def vision_forward(self, groups: List[Tuple[int, torch.Tensor]], save: bool, drop_scales=None):
        """groups: [(modality index, images f32 [n,3,H,W])] -> (features f32 [n_img, D], saved state).
        ``drop_scales``: per layer (s_attn, s_mlp) per-image DropPath factors or None."""
        a, P, W, lay = self.arch, self.P, self.W, self.lay
        dev = self.dev
        S, d, ff, Rp, r = self.S, lay.d, lay.ff, lay.Rp, lay.r
        heads = a['vision_heads']
        n_img = sum(g[1].shape[0] for g in groups)
        M = n_img * S
        ce = 'clip_encoder.'
        mods = []
        for mu, img in groups:
            mods += [mu] * img.shape[0]
        img_mod = self._const(('img_mod', tuple(mods)), lambda: torch.tensor(mods, dtype=torch.int32))
        f32 = dict(dtype=torch.float32, device=dev); b16 = dict(dtype=_lib.t16(), device=dev)
        x = torch.empty(M, d, **f32)
        pos = P[ce + 'vision_pos_embed']
        start = 0
        for mu, img in groups:
            m = lay.vmods[mu]
            n = img.shape[0]
            wpe = W[('pe', m)]
            cin = wpe.shape[1] // (a['patch_size'] ** 2)
            patches = torch.empty(n * (S - 1), wpe.shape[1], **b16)
            ops.patch_im2col(img.contiguous(), patches, a['patch_size'], cin)
            ops.gemm(patches, wpe, x[start * S:], bias=P[f'{ce}patch_embeds.{m}.proj.bias'], R=pos[1:], r_period=S - 1,
                     c_group=S - 1, c_group_stride=S, c_row_off=1)
            start += n
        ops.cls_rows(P[ce + 'cls_token'].view(-1), pos[0], x, n_img, S)
        mk = dict(img_mod=img_mod, mask_r=r, mask_period=Rp, rows_per_img=S, alpha=self.scaling)
        pk = lambda l, nm, w: lay.pk(self._lora_pack, l, nm, w)
        we = lambda l, nm: lay.weff(self._weff, l, nm)
        # row groups of the merged-weight GEMMs: images are packed modality by modality
        counts = [g[1].shape[0] for g in groups if g[1].shape[0] > 0]
        mus = [g[0] for g in groups if g[1].shape[0] > 0]
        ends = [0]
        for c in counts:
            ends.append(ends[-1] + c)
        rg_full = ([e * S for e in ends[1:]], mus)
        rg_cls = (ends[1:], mus)
        saved = []
        buf = {}
        idx = self._const(('cls_idx', n_img, S), lambda: torch.arange(n_img, dtype=torch.int32) * S)
        idxl = self._const(('cls_idx64', n_img, S), lambda: torch.arange(n_img, dtype=torch.int64) * S)
        main = torch.cuda.current_stream(dev)
        side = self._side_stream() if (save and self.overlap_tn) else None
        self.wait_packed()                                   # merged weights / adapter packs of this step (pack_lora's stream)

        def new(name, shape, kw):
            if save:
                return torch.empty(shape, **kw)
            t = buf.get((name, shape))                       # (the class-row pass of the last block has its own, smaller set)
            if t is None:
                t = buf[(name, shape)] = torch.empty(shape, **kw)
            return t

        deferred = []

        def lora_down(calls):
            """T = mask_modality(x . Acat^T) * (alpha/r) for the block's four linears: needed only by the backward pass (dB = dY^T T).
            Held back until the encoder is through (flush_lora_down): the 48 HBM-bound launches then run on the side stream beneath the
            head / loss / head-backward section -- ~1.5 ms of tiny kernels that leave the chip idle -- instead of taking compute units
            from the forward GEMMs and attention (their inputs are the saved activations, alive until the backward pass)."""
            if side is None:
                for xin_, A_, T_, kw in calls:
                    ops.gemm(xin_, A_, T_, **kw)
                return
            deferred.extend(calls)

        def flush_lora_down():
            if not deferred:
                return
            ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
            with torch.cuda.stream(side):
                for xin_, A_, T_, kw in deferred:
                    ops.gemm(xin_, A_, T_, **kw)
                    # (both normally live until the backward pass; if the graph is dropped instead, the allocator must not hand
                    #  their blocks out while these launches are pending)
                    T_.record_stream(side); xin_.record_stream(side)
            deferred.clear()

        # Residual adds live in the LayerNorm that follows them (reid_add_layernorm_fwd): the out-projection and fc2 GEMMs store their
        # 16-bit branch output, the add + LN kernel streams x once.  LN1 of block l+1 is therefore produced at the end of block l.
        add_ln = self.add_ln
        L = a['vision_layers']
        nxt = None
        for l in range(L):
            lp = f'{ce}vision_layers.{l}.'
            if nxt is None:
                h = new('h', (M, d), b16); mean1 = new('m1', (M,), f32); rstd1 = new('r1', (M,), f32)
                ops.layernorm_fwd(x, P[lp + 'ln1.weight'], P[lp + 'ln1.bias'], y_bf16=h, mean=mean1, rstd=rstd1)
            else:
                h, mean1, rstd1 = nxt
                nxt = None
            qkv = new('qkv', (M, 3 * d), b16)
            ops.gemm(h, we(l, 'qkv'), qkv, bias=W[('v', l, 'bqkv')], row_groups=rg_full)
            o = new('o', (M, d), b16); lse = new('lse', (n_img, heads, S), f32)
            ops.attn_fwd(qkv, o, lse, n_img, S, heads, q_tiles=1 if (self.cls_prune and l == a['vision_layers'] - 1) else 0)
            sa, sm_ = (None, None) if drop_scales is None else drop_scales[l]
            last = self.cls_prune and l == a['vision_layers'] - 1
            if last:
                # Only the class-token row of the last block's output is ever used (clip_backbone.py:281: x[:, 0]), and rows do
                # not mix after the attention core: out-projection, LN2 and the MLP of the LAST block run on the n_img class rows
                # instead of all n_img*197 (same function; the reference computes and discards the other 196/197).
                Mr, rpi, rg = n_img, 1, rg_cls
                xin = x.index_select(0, idxl); oin = o.index_select(0, idxl)
            else:
                Mr, rpi, rg = M, S, rg_full
                xin, oin = x, o
            mkr = dict(img_mod=img_mod, mask_r=r, mask_period=Rp, rows_per_img=rpi, alpha=self.scaling)
            xm = new('xm', (Mr, d), f32)
            h2 = new('h2', (Mr, d), b16); mean2 = new('m2', (Mr,), f32); rstd2 = new('r2', (Mr,), f32)
            if add_ln and not last:
                yb = buf.get('yb')
                if yb is None:
                    # branch output of the out-projection / fc2 GEMM: consumed by the add + LayerNorm kernel, never by an MFMA, so it
                    # is IEEE half (11 significant bits, saturating) in BOTH flavors (r03 stored bf16 in the bf16 flavor: one more
                    # 8-bit rounding per residual branch, bn_features/8 3.8e-3 -> 4.3e-3)
                    yb = buf['yb'] = torch.empty(M, d, dtype=torch.float16, device=dev)
                ops.gemm(oin, we(l, 'out'), yb, bias=P[lp + 'attn.out_proj.shared_linear.bias'], row_groups=rg)
                ops.add_layernorm_fwd(xin, yb, xm, P[lp + 'ln2.weight'], P[lp + 'ln2.bias'], h2, mean2, rstd2, row_scale=sa, rows_per_img=rpi)
            else:
                ops.gemm(oin, we(l, 'out'), xm, bias=P[lp + 'attn.out_proj.shared_linear.bias'], R=xin, row_scale=sa, rows_per_img=rpi,
                         row_groups=rg)
                ops.layernorm_fwd(xm, P[lp + 'ln2.weight'], P[lp + 'ln2.bias'], y_bf16=h2, mean=mean2, rstd=rstd2)
            u = new('u', (Mr, ff), b16) if save else None      # holds gelu'(pre-activation): the backward epilogue is one multiply
            g = new('g', (Mr, ff), b16)
            ops.gemm(h2, we(l, 'fc1'), g, bias=P[lp + 'mlp.fc1.shared_linear.bias'], act='gelu_dsave' if save else 'gelu', C2=u,
                     row_groups=rg)
            T = To = T1 = T2 = None
            if save:
                T = torch.empty(M, 3 * Rp, **b16); To = torch.empty(Mr, Rp, **b16)
                T1 = torch.empty(Mr, Rp, **b16); T2 = torch.empty(Mr, Rp, **b16)
                lora_down([(h, pk(l, 'qkv', 'A'), T, mk), (oin, pk(l, 'out', 'A'), To, mkr), (h2, pk(l, 'fc1', 'A'), T1, mkr),
                           (g, pk(l, 'fc2', 'A'), T2, mkr)])
                if side is not None and not self.lora_down_defer:
                    flush_lora_down()
            xn = torch.empty(Mr, d, **f32) if save else new('xn' + str(l & 1) + ('c' if last else ''), (Mr, d), f32)
            if add_ln and not last and l + 1 < L:
                np_ = f'{ce}vision_layers.{l + 1}.'
                ops.gemm(g, we(l, 'fc2'), yb, bias=P[lp + 'mlp.fc2.shared_linear.bias'], row_groups=rg)
                # (eval keeps two h buffers: the next block's h is written while nothing reads this block's any more, but a fresh
                #  name keeps the lifetime obvious)
                nxt = (new('h' + str((l + 1) & 1), (M, d), b16), new('m1' + str((l + 1) & 1), (M,), f32), new('r1' + str((l + 1) & 1), (M,), f32))
                ops.add_layernorm_fwd(xm, yb, xn, P[np_ + 'ln1.weight'], P[np_ + 'ln1.bias'], nxt[0], nxt[1], nxt[2], row_scale=sm_,
                                      rows_per_img=rpi)
            else:
                ops.gemm(g, we(l, 'fc2'), xn, bias=P[lp + 'mlp.fc2.shared_linear.bias'], R=xm, row_scale=sm_, rows_per_img=rpi,
                         row_groups=rg)
            if save:
                saved.append(dict(x=x, h=h, mean1=mean1, rstd1=rstd1, T=T, qkv=qkv, o=o, lse=lse, To=To, xm=xm, h2=h2,
                                  mean2=mean2, rstd2=rstd2, T1=T1, u=u, g=g, T2=T2, sa=sa, sm=sm_, cls=last, o_rows=oin))
            x = xn
        cls_h = torch.empty(n_img, d, **b16); mf = torch.empty(n_img, **f32); rf = torch.empty(n_img, **f32)
        ops.layernorm_fwd(x, P[ce + 'vision_ln_final.weight'], P[ce + 'vision_ln_final.bias'], y_bf16=cls_h, mean=mf, rstd=rf,
                          row_index=None if self.cls_prune else idx)
        feats = torch.empty(n_img, a['fusion_dim'], **f32)
        ops.gemm(cls_h, W['vproj'], feats)
        if self.lora_down_defer:
            flush_lora_down()
        state = dict(layers=saved, x_final=x, idx=idx, idxl=idxl, mf=mf, rf=rf, img_mod=img_mod, n_img=n_img, cls_h=cls_h,
                     groups=groups, cls_prune=self.cls_prune, rg_full=rg_full, rg_cls=rg_cls) if save else None
        return feats, state

    # ------------------------------------------------------------------------------- vision backward
    def vision_dense_keys(self) -> List[str]:
        """Reference keys of the vision backbone tensors (everything of clip_encoder that is not the text tower, its
        projection, or a LoRA adapter), in a fixed order: the non-LoRA inputs / gradient outputs of VisionEncodeFn."""
        tp = 'clip_encoder.clip_model.'
        return [k for k in self.P if k.startswith('clip_encoder.') and not k.startswith(tp) and '.loras.' not in k
                and k != 'clip_encoder.text_proj.weight' and self.P[k] is not self.lora_arena]

    def vision_backward(self, st, dfeat: torch.Tensor, want_dense: bool = False):
        """dfeat f32 [n_img, D] -> fp32 gradient of the LoRA arena (same layout as the arena); with ``want_dense`` also a
        dict {reference key: fp32 gradient} for the backbone tensors (weights by reduce-over-rows GEMMs dY^T X, biases as
        column sums of dY, LayerNorm affine pairs from the LN backward kernel, position / class embeddings and the patch
        convolutions from the gradient of the embedded sequence) -- the reference's ``freeze_backbone=False`` mode."""
        a, P, W, lay = self.arch, self.P, self.W, self.lay
        dev = self.dev
        S, d, ff, Rp, r = self.S, lay.d, lay.ff, lay.Rp, lay.r
        heads = a['vision_heads']
        n_img = st['n_img']; M = n_img * S
        ce = 'clip_encoder.'
        f32 = dict(dtype=torch.float32, device=dev); b16 = dict(dtype=_lib.t16(), device=dev)
        grad = torch.zeros(lay.size, **f32)               # the dA/dB GEMMs accumulate (beta=1): no per-call fill
        mk = dict(img_mod=st['img_mod'], mask_r=r, mask_period=Rp, rows_per_img=S, alpha=self.scaling)
        pk = lambda l, nm, w: lay.pk(self._lora_pack, l, nm, w)
        gA = lambda l, nm: lay.view_A(grad, l, nm)
        gB = lambda l, nm: lay.view_B(grad, l, nm)
        scale_t = None
        if self.loss_scaling or self.dx_half:
            amax = dfeat.abs().amax().clamp_min(1e-30)
            scale_t = torch.exp2(torch.floor(torch.log2(512.0 / amax))).clamp(2.0 ** -20, 2.0 ** 40)
            dfeat = dfeat * scale_t
        dfb = ops.to_bf16(dfeat)
        gx = dict(dtype=torch.float16 if self.dx_half else torch.float32, device=dev)      # the residual-stream gradient
        dcls = torch.empty(n_img, d, **b16)
        ops.gemm(dfb, W['vprojT'], dcls)
        prune = bool(st.get('cls_prune'))
        idxl = st['idxl']
        if prune:                                           # class rows only until the last block's attention (see vision_forward)
            dx = torch.empty(M, d, **gx); dxb = torch.empty(M, d, **b16)        # first written (all rows) by the last block's LN1 backward
            dx_c = torch.empty(n_img, d, **gx); dxb_c = torch.empty(n_img, d, **b16)
        else:
            dx = torch.zeros(M, d, **gx); dxb = torch.zeros(M, d, **b16)
        dense = {} if want_dense else None
        ones8 = torch.ones(M, 8, **b16) if want_dense else None

        def wgrad(dY, X):                                   # dW [N, K] = dY^T X  (rows reduced on the matrix cores)
            out = torch.empty(dY.shape[1], X.shape[1], **f32)
            ops.gemm_tn(dY, X, out)
            return out

        def colsum(dY):                                     # db [N] = column sums of dY (same kernel against a block of ones)
            out = torch.empty(dY.shape[1], 8, **f32)
            ops.gemm_tn(dY, ones8[:dY.shape[0]], out)
            return out[:, 0].contiguous()

        def ln_grads(key):
            if not want_dense:
                return None, None
            dense[key + '.weight'] = torch.zeros(d, **f32); dense[key + '.bias'] = torch.zeros(d, **f32)
            return dense[key + '.weight'], dense[key + '.bias']

        if want_dense:
            dense[ce + 'vision_proj.weight'] = wgrad(dfb, st['cls_h'])
        dgf, dbf = ln_grads(ce + 'vision_ln_final')
        # dxb always holds the gradient ENTERING the next residual branch: dx times that branch's DropPath factor
        if prune:
            ops.layernorm_bwd(dcls, st['x_final'], P[ce + 'vision_ln_final.weight'], st['mf'], st['rf'], dx_c, dx_bf16=dxb_c,
                              bf16_row_scale=st['layers'][-1]['sm'], rows_per_img=1, dgamma=dgf, dbeta=dbf)
        else:
            ops.layernorm_bwd(dcls, st['x_final'], P[ce + 'vision_ln_final.weight'], st['mf'], st['rf'], dx, dx_bf16=dxb,
                              row_index=st['idx'], bf16_row_scale=st['layers'][-1]['sm'], rows_per_img=S, dgamma=dgf, dbeta=dbf)
        # Scratch.  Everything the SIDE stream reads (the dY of the four linears: dxb, du, dxmb, dqkv) exists twice, used by alternate
        # layers: the main stream may then run a whole layer ahead of the adapter-gradient kernels without overwriting their inputs.
        # U* (= dY . Bcat, the rank-r cotangents) are written and read on the side stream only.
        U2 = torch.empty(M, Rp, **b16); U1 = torch.empty(M, Rp, **b16); Uo = torch.empty(M, Rp, **b16)
        Uq = torch.empty(M, 3 * Rp, **b16)
        du2 = [torch.empty(M, ff, **b16) for _ in range(2)]
        dqkv2 = [torch.empty(M, 3 * d, **b16) for _ in range(2)]
        dxmb2 = [torch.empty(M, d, **b16) for _ in range(2)]
        dxb2 = [dxb, torch.empty(M, d, **b16)]
        dh = torch.empty(M, d, **b16); do = torch.empty(M, d, **b16)
        delta = torch.empty(n_img, heads, S, **f32)
        dxm = torch.empty(M, d, **gx)
        weT = lambda l, nm: lay.weff(self._weff, l, nm, transposed=True)
        rg_full, rg_cls = st['rg_full'], st['rg_cls']
        main = torch.cuda.current_stream(dev)
        side = self._side_stream() if self.overlap_tn else None
        side_done = {}
        L = a['vision_layers']
        # the gradient ENTERING layer l (dx times that layer's MLP DropPath factor, 16-bit) lives in dxb2[l & 1]
        if not prune:
            cur_dxb = dxb2[(L - 1) & 1]
            if cur_dxb is not dxb:
                cur_dxb.copy_(dxb)

        fuse_u_db, fuse_max_n = self.lora_fused, self.lora_fused_max_n
        u_part = torch.empty(M, Rp, **f32) if fuse_u_db else None      # fp32 partial U of fc1's four column blocks (side stream only)
        # Class-row scratch of the pruned last block.  Read and written by the SIDE stream (lora_grads) long after the main stream has
        # moved on, so it must not be released inside the loop: a block freed by the main stream is handed to the main stream's next
        # allocation at once (wgrad / colsum / ln_grads with want_dense), whatever other streams still have pending on it (r03 found
        # this hazard for forward-only calls).  Allocated here, these tensors die when this function returns -- after the join of the
        # side stream into the main stream below has been enqueued, which orders every later main-stream use behind the side kernels.
        cls_tmp = None
        if prune:
            cls_tmp = dict(U2=torch.empty(n_img, Rp, **b16), U1=torch.empty(n_img, Rp, **b16), Uo=torch.empty(n_img, Rp, **b16),
                           du=torch.empty(n_img, ff, **b16), dh=torch.empty(n_img, d, **b16), do=torch.empty(n_img, d, **b16),
                           dxm=torch.empty(n_img, d, **gx), dxmb=torch.empty(n_img, d, **b16))

        def lora_grads(l, calls):
            """Adapter gradients of one linear on the side stream: U = mask(dY . Bcat) * (alpha/r), dB += dY^T T, dA += U^T X.
            ``calls`` = [(dY, BT, U, mask kwargs, [(X operand, Y operand, out), ...])]."""
            def run():
                for dY, BT, U, kw, tns in calls:
                    # tns[0] = (dY, T, dB): with N = 768 output columns (every linear but fc1) U and dB come out of ONE pass over dY
                    same = tns[0][0].data_ptr() == dY.data_ptr() and tns[0][0].shape == dY.shape and tns[0][0].stride() == dY.stride()
                    if fuse_u_db and same and ops.lora_bwd_fused_ok(dY.shape[1], tns[0][1].shape[1]) and dY.shape[1] <= fuse_max_n:
                        ops.lora_bwd_fused(dY, tns[0][1], BT, U, tns[0][2], kw['img_mod'], kw['rows_per_img'], kw['mask_r'], kw['alpha'],
                                           u_partial=u_part[:dY.shape[0]] if dY.shape[1] > 768 else None)
                        rest = tns[1:]
                    else:
                        if not _EXP_SKIP_U:                  # (timing experiment only: REID_EXP_SKIP_U=1 leaves U unwritten -> wrong dA)
                            ops.gemm(dY, BT, U, **kw)
                        rest = tns
                    for xx, yy, out in rest:
                        if _EXP_SKIP_DA:                     # (timing experiment only: REID_EXP_SKIP_DA=1 leaves dA unwritten)
                            continue
                        # (xx, yy, out) = (U [M, G Rp], the linear's input X [M, K], dA [G Rp, K]): one pass over X, one image per workgroup
                        ng = xx.shape[1] // Rp
                        if self.lora_da_fused and xx.shape[1] == ng * Rp and out.shape[0] == ng * Rp and \
                                ops.lora_da_fused_ok(yy.shape[1], Rp, kw['rows_per_img'], kw['mask_r'], ng):
                            ops.lora_da_fused(yy, xx, out, kw['img_mod'], kw['rows_per_img'], kw['mask_r'], n_groups=ng)
                        else:
                            ops.gemm_tn(xx, yy, out, beta=1.0)
            if side is None:
                run()
                return
            ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
            with torch.cuda.stream(side):
                run()

        def layer_done(l):
            if side is not None:
                ev = torch.cuda.Event(); ev.record(side); side_done[l] = ev

        def wait_side(l):
            ev = side_done.pop(l, None)
            if ev is not None:
                main.wait_event(ev)

        for l in reversed(range(L)):
            s = st['layers'][l]
            lp = f'{ce}vision_layers.{l}.'
            b = l & 1
            c = bool(s.get('cls'))                          # this block's MLP / out-projection ran on the class rows only
            if c:
                Mr, rg = n_img, rg_cls
                mkr = dict(img_mod=st['img_mod'], mask_r=r, mask_period=Rp, rows_per_img=1, alpha=self.scaling)
                gy, gyb = dx_c, dxb_c
                U2r, U1r, Uor, dur, dhr, dor, dxmr, dxmbr = (cls_tmp[k_] for k_ in ('U2', 'U1', 'Uo', 'du', 'dh', 'do', 'dxm', 'dxmb'))
                rpi = 1
            else:
                Mr, mkr, gy, gyb, rg = M, mk, dx, dxb2[b], rg_full
                U2r, U1r, Uor, dur, dhr, dor, dxmr, dxmbr, rpi = U2, U1, Uo, du2[b], dh, do, dxm, dxmb2[b], S
            dqkv = dqkv2[b]
            # ---- fc2:  x_next = xm + g W2_eff^T + b2
            lora_grads(l, [(gyb, pk(l, 'fc2', 'BT'), U2r, mkr, [(gyb, s['T2'], gB(l, 'fc2')), (U2r, s['g'], gA(l, 'fc2'))])])
            ops.gemm(gyb, weT(l, 'fc2'), dur, act='mul_aux', aux=s['u'], row_groups=rg)
            if want_dense:
                dense[lp + 'mlp.fc2.shared_linear.weight'] = wgrad(gyb, s['g'])
                dense[lp + 'mlp.fc2.shared_linear.bias'] = colsum(gyb)
            # ---- fc1:  u = h2 W1_eff^T + b1
            lora_grads(l, [(dur, pk(l, 'fc1', 'BT'), U1r, mkr, [(dur, s['T1'], gB(l, 'fc1')), (U1r, s['h2'], gA(l, 'fc1'))])])
            ops.gemm(dur, weT(l, 'fc1'), dhr, row_groups=rg)
            if want_dense:
                dense[lp + 'mlp.fc1.shared_linear.weight'] = wgrad(dur, s['h2'])
                dense[lp + 'mlp.fc1.shared_linear.bias'] = colsum(dur)
            # ---- LN2
            dg2, db2 = ln_grads(lp + 'ln2')
            ops.layernorm_bwd(dhr, s['xm'], P[lp + 'ln2.weight'], s['mean2'], s['rstd2'], dxmr, dx_bf16=dxmbr, dres=gy,
                              bf16_row_scale=s['sa'], rows_per_img=rpi, dgamma=dg2, dbeta=db2)
            # ---- out proj:  xm = x + o Wo_eff^T + bo
            lora_grads(l, [(dxmbr, pk(l, 'out', 'BT'), Uor, mkr, [(dxmbr, s['To'], gB(l, 'out')), (Uor, s['o_rows'], gA(l, 'out'))])])
            ops.gemm(dxmbr, weT(l, 'out'), dor, row_groups=rg)
            if want_dense:
                dense[lp + 'attn.out_proj.shared_linear.weight'] = wgrad(dxmbr, s['o_rows'])
                dense[lp + 'attn.out_proj.shared_linear.bias'] = colsum(dxmbr)
            if c:                                           # back to all rows: zero everywhere but the class rows
                do.zero_(); do.index_copy_(0, idxl, dor)
                dxm.zero_(); dxm.index_copy_(0, idxl, dxmr)
            # ---- attention
            ops.attn_bwd(s['qkv'], s['o'], do, s['lse'], dqkv, delta, n_img, S, heads, q_tiles=1 if c else 0)
            # ---- qkv:  qkv = h Wqkv_eff^T + b (one adapter set per projection)
            bT = pk(l, 'qkv', 'BT')                         # [Rp, 3d]
            gBq = gB(l, 'qkv')                              # [3d, Rp]
            lora_grads(l, [(dqkv[:, g * d:(g + 1) * d], bT[:, g * d:(g + 1) * d], Uq[:, g * Rp:(g + 1) * Rp], mk,
                            [(dqkv[:, g * d:(g + 1) * d], s['T'][:, g * Rp:(g + 1) * Rp], gBq[g * d:(g + 1) * d])] +
                            ([(Uq, s['h'], gA(l, 'qkv'))] if g == 2 else [])) for g in range(3)])
            layer_done(l)
            if l == 0 and not want_dense:
                # nothing below layer 0 trains under the default freeze: the gradient of the embedded sequence (dX of the
                # q|k|v projection and the LN1 backward) would be computed only to be thrown away
                break
            ops.gemm(dqkv, weT(l, 'qkv'), dh, row_groups=rg_full)
            if want_dense:
                gw = wgrad(dqkv, s['h']); gb_ = colsum(dqkv)
                for gi, nm in enumerate('qkv'):
                    dense[lp + f'attn.{nm}_proj.shared_linear.weight'] = gw[gi * d:(gi + 1) * d]
                    dense[lp + f'attn.{nm}_proj.shared_linear.bias'] = gb_[gi * d:(gi + 1) * d]
            # ---- LN1: writes the gradient entering layer l - 1 into the OTHER buffer set; its previous readers on the side stream
            # are the adapter-gradient kernels of layer l + 1 (and this layer's successor will overwrite du / dxmb / dqkv of that set)
            wait_side(l + 1)
            dg1, db1 = ln_grads(lp + 'ln1')
            ops.layernorm_bwd(dh, s['x'], P[lp + 'ln1.weight'], s['mean1'], s['rstd1'], dx, dx_bf16=dxb2[(l - 1) & 1], dres=dxm,
                              bf16_row_scale=st['layers'][l - 1]['sm'] if l > 0 else None, rows_per_img=S, dgamma=dg1, dbeta=db1)
        if side is not None:                                # the gradient arena is complete only when the side stream is
            ev = torch.cuda.Event(); ev.record(side); main.wait_event(ev)
        if want_dense:
            # embedded sequence x0[img, t] = (cls | patch_t) + pos[t]: dx now holds d loss / d x0
            ones_r = torch.ones(1, n_img, **f32)
            dpos = torch.empty(1, S * d, **f32)
            dx = dx.float()
            ops.sgemm(ones_r, dx.view(n_img, S * d), dpos)
            dpos = dpos.view(S, d)
            dense[ce + 'vision_pos_embed'] = dpos
            dense[ce + 'cls_token'] = dpos[0].reshape(1, 1, d).clone()
            start = 0
            acc_w, acc_b = {}, {}
            for mu, img in st['groups']:
                m = lay.vmods[mu]
                n = img.shape[0]
                wpe = W[('pe', m)]
                cin = wpe.shape[1] // (a['patch_size'] ** 2)
                patches = torch.empty(n * (S - 1), wpe.shape[1], **b16)
                ops.patch_im2col(img.contiguous(), patches, a['patch_size'], cin)
                dP = ops.to_bf16(dx.view(n_img, S, d)[start:start + n, 1:, :].reshape(n * (S - 1), d).contiguous())
                gw = wgrad(dP, patches); gb_ = colsum(dP)
                kw, kb = f'{ce}patch_embeds.{m}.proj.weight', f'{ce}patch_embeds.{m}.proj.bias'
                acc_w[kw] = gw if kw not in acc_w else acc_w[kw] + gw
                acc_b[kb] = gb_ if kb not in acc_b else acc_b[kb] + gb_
                start += n
            for k, v in acc_w.items():
                dense[k] = v.view(P[k].shape)
            dense.update(acc_b)
        if scale_t is not None:
            grad.mul_(1.0 / scale_t)
            if want_dense:
                for k in dense:
                    dense[k] = dense[k] * (1.0 / scale_t)
        return (grad, dense) if want_dense else grad

    # ------------------------------------------------------------------------------- text forward
    def text_keys(self) -> List[str]:
        """Reference keys of the text tower + its projection, in a fixed order (inputs / gradient outputs of TextEncodeFn)."""
        tp = 'clip_encoder.clip_model.text_model.'
        return [k for k in self.P if k.startswith(tp)] + ['clip_encoder.text_proj.weight']

    def text_trains(self) -> bool:
        return any(self.P[k].requires_grad for k in self.text_keys())

    def text_forward(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor], save: bool = False):
        """HF CLIP text tower + text_proj (clip_backbone.py:288-313).  ``save``: also return the activations the backward
        pass needs (``(features, state)``); the default returns the features only."""
        a, P, W = self.arch, self.P, self.W
        dev = self.dev
        tp = 'clip_encoder.clip_model.text_model.'
        B, T = input_ids.shape
        td, tff, heads = a['text_hidden_dim'], a['text_mlp_dim'], a['text_heads']
        f32 = dict(dtype=torch.float32, device=dev); b16 = dict(dtype=_lib.t16(), device=dev)
        ids = input_ids.to(dev).long().contiguous()
        x = torch.empty(B * T, td, **f32)
        ops.embed_tokens(P[tp + 'embeddings.token_embedding.weight'].detach(), P[tp + 'embeddings.position_embedding.weight'].detach(),
                         ids, x)
        km = None
        if attention_mask is not None:
            km = attention_mask.to(dev).to(torch.uint8).contiguous()
        M = B * T
        layers = []
        if not save:
            h = torch.empty(M, td, **b16); qkv = torch.empty(M, 3 * td, **b16); o = torch.empty(M, td, **b16)
            g = torch.empty(M, tff, **b16); xm = torch.empty(M, td, **f32); xn = torch.empty(M, td, **f32)
        for l in range(a['text_layers']):
            lp = f'{tp}encoder.layers.{l}.'
            if save:
                h = torch.empty(M, td, **b16); qkv = torch.empty(M, 3 * td, **b16); o = torch.empty(M, td, **b16)
                g = torch.empty(M, tff, **b16); xm = torch.empty(M, td, **f32); xn = torch.empty(M, td, **f32)
                h2 = torch.empty(M, td, **b16); u = torch.empty(M, tff, **b16); lse = torch.empty(B, heads, T, **f32)
                m1 = torch.empty(M, **f32); r1 = torch.empty(M, **f32); m2 = torch.empty(M, **f32); r2 = torch.empty(M, **f32)
            else:
                h2, u, lse, m1, r1, m2, r2 = h, None, None, None, None, None, None
            ops.layernorm_fwd(x, P[lp + 'layer_norm1.weight'], P[lp + 'layer_norm1.bias'], y_bf16=h, mean=m1, rstd=r1)
            ops.gemm(h, W[('t', l, 'qkv')], qkv, bias=W[('t', l, 'bqkv')])
            ops.attn_fwd(qkv, o, lse, B, T, heads, causal=True, key_mask=km)
            ops.gemm(o, W[('t', l, 'out')], xm, bias=P[lp + 'self_attn.out_proj.bias'], R=x)
            ops.layernorm_fwd(xm, P[lp + 'layer_norm2.weight'], P[lp + 'layer_norm2.bias'], y_bf16=h2, mean=m2, rstd=r2)
            ops.gemm(h2, W[('t', l, 'fc1')], g, bias=P[lp + 'mlp.fc1.bias'], act='quick_gelu', C2=u)
            ops.gemm(g, W[('t', l, 'fc2')], xn, bias=P[lp + 'mlp.fc2.bias'], R=xm)
            if save:
                layers.append(dict(x=x, h=h, m1=m1, r1=r1, qkv=qkv, o=o, lse=lse, xm=xm, h2=h2, m2=m2, r2=r2, u=u, g=g))
                x = xn
            else:
                x, xn = xn, x
        eos = (ids == a['text_eos_id']).int().argmax(dim=-1)                  # first EOS (HF pooling rule)
        idx = (torch.arange(B, device=dev) * T + eos).to(torch.int32)
        pooled = torch.empty(B, td, **b16)
        mf = torch.empty(B, **f32) if save else None; rf = torch.empty(B, **f32) if save else None
        ops.layernorm_fwd(x, P[tp + 'final_layer_norm.weight'], P[tp + 'final_layer_norm.bias'], y_bf16=pooled, mean=mf, rstd=rf,
                          row_index=idx)
        feats = torch.empty(B, a['fusion_dim'], **f32)
        ops.gemm(pooled, W['tproj'], feats)
        if not save:
            return feats
        return feats, dict(layers=layers, x_final=x, idx=idx, mf=mf, rf=rf, pooled=pooled, ids=ids, km=km, B=B, T=T)

    def text_backward(self, st, dfeat: torch.Tensor) -> Dict[str, torch.Tensor]:
        """dfeat f32 [B, D] -> {reference key: fp32 gradient} for every tensor of ``text_keys()`` (freeze_backbone=False /
        freeze_text_backbone=False).  Same scheme as the vision backward without LoRA: dX through transposed 16-bit packs,
        dW = dY^T X on the reduce-over-rows GEMM, biases as column sums, LayerNorm pairs from the LN backward kernel, causal
        attention backward, embedding tables by scatter-add / batch sum."""
        a, P, W = self.arch, self.P, self.W
        dev = self.dev
        tp = 'clip_encoder.clip_model.text_model.'
        B, T = st['B'], st['T']
        M = B * T
        td, tff, heads = a['text_hidden_dim'], a['text_mlp_dim'], a['text_heads']
        f32 = dict(dtype=torch.float32, device=dev); b16 = dict(dtype=_lib.t16(), device=dev)
        G: Dict[str, torch.Tensor] = {}
        scale_t = None
        if self.loss_scaling:
            amax = dfeat.abs().amax().clamp_min(1e-30)
            scale_t = torch.exp2(torch.floor(torch.log2(512.0 / amax))).clamp(2.0 ** -20, 2.0 ** 40)
            dfeat = dfeat * scale_t
        ones8 = torch.ones(M, 8, **b16)

        def wgrad(dY, X):
            out = torch.empty(dY.shape[1], X.shape[1], **f32)
            ops.gemm_tn(dY, X, out)
            return out

        def colsum(dY):
            out = torch.empty(dY.shape[1], 8, **f32)
            ops.gemm_tn(dY, ones8[:dY.shape[0]], out)
            return out[:, 0].contiguous()

        def ln_pair(key):
            G[key + '.weight'] = torch.zeros(td, **f32); G[key + '.bias'] = torch.zeros(td, **f32)
            return G[key + '.weight'], G[key + '.bias']

        dfb = ops.to_bf16(dfeat)
        G['clip_encoder.text_proj.weight'] = wgrad(dfb, st['pooled'])
        dpool = torch.empty(B, td, **b16)
        ops.gemm(dfb, W['tprojT'], dpool)
        dx = torch.zeros(M, td, **f32); dxb = torch.zeros(M, td, **b16)
        dg_, db_ = ln_pair(tp + 'final_layer_norm')
        ops.layernorm_bwd(dpool, st['x_final'], P[tp + 'final_layer_norm.weight'], st['mf'], st['rf'], dx, dx_bf16=dxb,
                          row_index=st['idx'], dgamma=dg_, dbeta=db_)
        du = torch.empty(M, tff, **b16); dh = torch.empty(M, td, **b16); do = torch.empty(M, td, **b16)
        dqkv = torch.empty(M, 3 * td, **b16); delta = torch.empty(B, heads, T, **f32)
        dxm = torch.empty(M, td, **f32); dxmb = torch.empty(M, td, **b16)
        for l in reversed(range(a['text_layers'])):
            s = st['layers'][l]
            lp = f'{tp}encoder.layers.{l}.'
            ops.gemm(dxb, W[('t', l, 'fc2T')], du, act='dquick_gelu', aux=s['u'])
            G[lp + 'mlp.fc2.weight'] = wgrad(dxb, s['g']); G[lp + 'mlp.fc2.bias'] = colsum(dxb)
            ops.gemm(du, W[('t', l, 'fc1T')], dh)
            G[lp + 'mlp.fc1.weight'] = wgrad(du, s['h2']); G[lp + 'mlp.fc1.bias'] = colsum(du)
            dg_, db_ = ln_pair(lp + 'layer_norm2')
            ops.layernorm_bwd(dh, s['xm'], P[lp + 'layer_norm2.weight'], s['m2'], s['r2'], dxm, dx_bf16=dxmb, dres=dx,
                              dgamma=dg_, dbeta=db_)
            ops.gemm(dxmb, W[('t', l, 'outT')], do)
            G[lp + 'self_attn.out_proj.weight'] = wgrad(dxmb, s['o']); G[lp + 'self_attn.out_proj.bias'] = colsum(dxmb)
            ops.attn_bwd(s['qkv'], s['o'], do, s['lse'], dqkv, delta, B, T, heads, causal=True, key_mask=st['km'])
            ops.gemm(dqkv, W[('t', l, 'qkvT')], dh)
            gw = wgrad(dqkv, s['h']); gb = colsum(dqkv)
            for gi, nm in enumerate('qkv'):
                G[lp + f'self_attn.{nm}_proj.weight'] = gw[gi * td:(gi + 1) * td]
                G[lp + f'self_attn.{nm}_proj.bias'] = gb[gi * td:(gi + 1) * td]
            dg_, db_ = ln_pair(lp + 'layer_norm1')
            ops.layernorm_bwd(dh, s['x'], P[lp + 'layer_norm1.weight'], s['m1'], s['r1'], dx, dx_bf16=dxb, dres=dxm,
                              dgamma=dg_, dbeta=db_)
        # embeddings: x0[b, t] = tok[ids[b, t]] + pos[t]
        tok = P[tp + 'embeddings.token_embedding.weight']; posw = P[tp + 'embeddings.position_embedding.weight']
        dtok = torch.zeros(tok.shape, **f32)
        ops.scatter_add_rows(dx, st['ids'].reshape(-1).to(torch.int32).contiguous(), dtok)
        dpos = torch.zeros(posw.shape, **f32)
        dsum = torch.empty(1, T * td, **f32)
        ops.sgemm(torch.ones(1, B, **f32), dx.view(B, T * td), dsum)
        dpos[:T] = dsum.view(T, td)
        G[tp + 'embeddings.token_embedding.weight'] = dtok
        G[tp + 'embeddings.position_embedding.weight'] = dpos
        if scale_t is not None:
            for k in G:
                G[k] = G[k] * (1.0 / scale_t)
        return G


class TextEncodeFn(torch.autograd.Function):
    """Autograd boundary of the text executor when the text tower trains: (tokens, text tensors) -> [B, D] features."""

    @staticmethod
    def forward(ctx, engine: Engine, input_ids, attention_mask, *params):
        feats, st = engine.text_forward(input_ids, attention_mask, save=True)
        ctx.engine, ctx.st, ctx.n = engine, st, len(params)
        return feats

    @staticmethod
    def backward(ctx, dfeat):
        _lib.set_flavor(ctx.engine.flavor)
        G = ctx.engine.text_backward(ctx.st, dfeat.contiguous().float())
        ctx.st = None
        keys = ctx.engine.text_keys()
        return (None, None, None) + tuple(G.get(k) if ctx.needs_input_grad[3 + i] else None for i, k in enumerate(keys))


class VisionEncodeFn(torch.autograd.Function):
    """Autograd boundary of the vision executor: (LoRA arena, images, [backbone tensors]) -> per-image features.
    ``dense`` = the tensors of ``engine.vision_dense_keys()`` in that order when the backbone trains (freeze_backbone=False),
    else empty: the default path carries no extra inputs."""

    @staticmethod
    def forward(ctx, engine: Engine, mods: Tuple[int, ...], lora_arena: torch.Tensor, n_images: int, *tensors):
        images, dense = tensors[:n_images], tensors[n_images:]
        need_dense = any(ctx.needs_input_grad[4 + n_images + i] for i in range(len(dense)))
        need = bool(ctx.needs_input_grad[2]) or need_dense
        scales = engine.pending_drop_scales if need else None     # set by the model for this call (training + drop_path > 0)
        engine.pending_drop_scales = None
        feats, st = engine.vision_forward(list(zip(mods, images)), save=need, drop_scales=scales)
        ctx.engine = engine
        ctx.st = st
        ctx.n_images = n_images
        ctx.n_dense = len(dense)
        ctx.need_dense = need_dense
        return feats

    @staticmethod
    def backward(ctx, dfeat):
        _lib.set_flavor(ctx.engine.flavor)
        res = ctx.engine.vision_backward(ctx.st, dfeat.contiguous().float(), want_dense=ctx.need_dense)
        ctx.st = None
        if ctx.need_dense:
            grad, dense = res
            keys = ctx.engine.vision_dense_keys()
            dgr = tuple(dense.get(k) if ctx.needs_input_grad[4 + ctx.n_images + i] else None for i, k in enumerate(keys))
        else:
            grad, dgr = res, (None,) * ctx.n_dense
        return (None, None, grad, None) + (None,) * ctx.n_images + dgr
