"""Step driver of the hot path: what ``train.py::train_epoch_fixed`` (train.py:684-1245) does
around ``model(...)`` / ``compute_loss`` -- accumulation, gradient sanitising, (adaptive) clipping, AdamW with the
reference's parameter groups, the warm-up + cosine LambdaLR -- without its host synchronisations.

The reference pays one ``.item()`` per parameter tensor for the gradient norm and one masked assignment per tensor for
the sanitiser (~1.2 k tensors).  Here the trainable set is a few flat fp32 buffers (the LoRA arena is one tensor) held
in a device table, and a step is three launches of libreid_hip (csrc/optim.hip); the adaptive rule's history and
percentile live on the device, so nothing is read back unless the caller asks for the statistics.

Semantics kept from the reference (file:line):
  * gradients accumulate over ``accum_steps`` micro-batches with ``loss / accum_steps`` (train.py:833-834, 895, 975);
  * non-finite gradient entries are zeroed before the norm (train.py:85-96);
  * adaptive clip: norm recorded when ``batch_idx % 200 == 0``, max_norm = clamp(1.15 * p70(last 10), 0.5, 3.0) once
    more than ten norms exist, else 1.0 (train.py:981-1001); fixed clip 0.5 otherwise (:1002-1008);
  * AdamW(param_groups, weight_decay) with torch defaults betas (0.9, 0.999), eps 1e-8 (train.py:1460);
  * LambdaLR warm-up + cosine per EPOCH, one lambda for every group (train.py:1249-1262, 1495-1501);
  * the head-LR "drop" of train.py:1602-1612 looks for 'classifier' in the group name, but the group is called
    'classification_head' (model.py:716), so it never fires; ``apply_head_lr`` reproduces that (and says so).
One deliberate difference: a non-finite LOSS does not skip the backward pass through a host check
(train.py:872-879 reads the loss with ``.item()``); its gradients are non-finite and are zeroed by the sanitiser.
"""
import ctypes as C
import math
from typing import Any, Dict, List, Optional

import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr

NORM_EVERY = 200          # train.py:1120


def warmup_cosine_lambda(total_epochs: int, warmup_epochs: int, start_factor: float = 0.01, min_factor: float = 0.01):
    """epoch -> LR scale (train.py:1249-1262)."""
    assert 0.0 < start_factor <= 1.0 and 0.0 < min_factor <= 1.0

    def lmbda(epoch: int) -> float:
        if epoch < warmup_epochs:
            return start_factor + (1.0 - start_factor) * (epoch + 1) / max(1, warmup_epochs)
        T = max(1, total_epochs - warmup_epochs)
        t = max(0, epoch - warmup_epochs)
        return min_factor + (1.0 - min_factor) * 0.5 * (1.0 + math.cos(math.pi * t / T))
    return lmbda


class _Entry(C.Structure):
    _fields_ = [('p', C.c_void_p), ('g', C.c_void_p), ('m', C.c_void_p), ('v', C.c_void_p), ('n', C.c_int64),
                ('lr', C.c_float), ('wd', C.c_float)]


class FusedAdamW:
    """AdamW over ``param_groups`` (the list ``model.get_learnable_params()`` returns, filtered to ``requires_grad``).

    Gradients live in persistent buffers (``p.grad`` is created once and zeroed by the step kernel), so the device table
    is built once and only rewritten when a learning rate changes.
    """

    def __init__(self, param_groups: List[Dict[str, Any]], weight_decay: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8):
        self.param_groups = []
        for g in param_groups:
            ps = [p for p in g['params'] if p.requires_grad]
            if ps:
                ng = dict(g); ng['params'] = ps; ng.setdefault('weight_decay', weight_decay); ng['initial_lr'] = ng['lr']
                self.param_groups.append(ng)
        if not self.param_groups:
            raise ValueError('no trainable parameters')
        self.betas, self.eps = betas, eps
        self.step_count = 0
        self.params = [p for g in self.param_groups for p in g['params']]
        dev = self.params[0].device
        if dev.type != 'cuda':
            raise _lib.ReidHipError('FusedAdamW runs on the HIP device only (no CPU path)')
        self.dev = dev
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.data_ptr() % 16:
                raise ValueError('trainable tensors must be contiguous, 16-byte aligned fp32')
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        assert lib().reid_opt_entry_bytes() == C.sizeof(_Entry)
        n = len(self.params)
        self.ws = torch.zeros(lib().reid_opt_ws_floats(n), device=dev)
        self.state = torch.zeros(lib().reid_opt_state_floats(), device=dev)
        self._table = torch.empty(n * C.sizeof(_Entry), dtype=torch.uint8, device=dev)
        self._write_table()

    def _write_table(self):
        ents = (_Entry * len(self.params))()
        i = 0
        for g in self.param_groups:
            for p in g['params']:
                if p.grad is None or p.grad.data_ptr() % 16 or not p.grad.is_contiguous():
                    raise ValueError('gradient buffers must stay allocated (do not zero_grad(set_to_none=True))')
                ents[i] = _Entry(p.data_ptr(), p.grad.data_ptr(), self.exp_avg[i].data_ptr(), self.exp_avg_sq[i].data_ptr(),
                                 p.numel(), float(g['lr']), float(g['weight_decay']))
                i += 1
        host = torch.frombuffer(bytearray(bytes(ents)), dtype=torch.uint8)
        self._table.copy_(host)
        self._grad_ptrs = [p.grad.data_ptr() for p in self.params]

    def set_lrs(self, scale_fn=None, epoch: Optional[int] = None):
        """LambdaLR.step(): lr = initial_lr * scale_fn(epoch) for every group."""
        if scale_fn is not None:
            for g in self.param_groups:
                g['lr'] = g['initial_lr'] * scale_fn(epoch)
        self._write_table()

    def apply_head_lr(self, epoch: int, head_lr: float = 3e-3, head_lr_warmup_epochs: int = 2) -> bool:
        """train.py:1602-1612.  Returns whether any group changed (never, with the reference's group names)."""
        changed = False
        if epoch >= head_lr_warmup_epochs:
            for g in self.param_groups:
                if 'classifier' in g.get('name', '') and g['lr'] != head_lr:
                    g['lr'] = head_lr; changed = True
        if changed:
            self._write_table()
        return changed

    def zero_grad(self):
        for p in self.params:
            p.grad.zero_()

    def step(self, adaptive_clip: bool = True, record_norm=False, fixed_max_norm: float = 0.5, zero_grad: bool = True,
             device_counters: bool = False):
        """sanitise -> norm -> clip coefficient -> AdamW (+ gradient clear).  No host synchronisation.
        ``device_counters``: the step number (bias correction) and the record-every-200 decision are taken from counters in
        the device state instead of host arguments, so the three launches can be captured once and replayed as a HIP graph
        (``record_norm`` is then the period, e.g. 200)."""
        if [p.grad.data_ptr() if p.grad is not None else 0 for p in self.params] != self._grad_ptrs:
            self._write_table()                              # someone replaced a .grad tensor
        n = len(self.params)
        self.step_count += 1
        check(lib().reid_opt_sumsq(ptr(self._table), n, ptr(self.ws), stream_ptr()))
        rec = -int(record_norm) if device_counters else int(bool(record_norm))
        check(lib().reid_opt_clip(ptr(self.ws), n, ptr(self.state), int(adaptive_clip), C.c_float(fixed_max_norm), rec, stream_ptr()))
        coef = C.c_void_p(self.state.data_ptr() + 8)         # state[2]
        check(lib().reid_opt_adamw(ptr(self._table), n, coef, C.c_float(self.betas[0]), C.c_float(self.betas[1]),
                                   C.c_float(self.eps), 0 if device_counters else self.step_count, int(zero_grad), ptr(self.state),
                                   stream_ptr()))
        # the kernel wrote through raw pointers: tell autograd / the engine's pack cache (engine.refresh keys on _version)
        for p in self.params:
            torch.autograd.graph.increment_version(p)

    def stats(self) -> Dict[str, float]:
        """Host copy of the last step's statistics (synchronises)."""
        s = self.state.cpu()
        return dict(grad_norm=float(s[1]), clip_coef=float(s[2]), max_norm=float(s[3]), non_finite=int(s[4]),
                    recorded_norms=int(s[5]))


class StepDriver:
    """The body of ``train_epoch_fixed`` for one process: forward, loss, backward, (all-reduce), optimizer step.

    ``module`` is the drop-in model or its ``parallel.DataParallel`` wrapper (same ``forward`` / ``compute_loss``).
    """

    def __init__(self, module, optimizer: FusedAdamW, accum_steps: int = 1, adaptive_clip: bool = True, dp=None):
        self.module, self.opt, self.accum_steps, self.adaptive_clip = module, optimizer, max(1, accum_steps), adaptive_clip
        self.dp = dp
        self.batch_idx = 0

    def start_epoch(self, epoch: int, scale_fn=None):
        model = getattr(self.module, 'model', self.module)
        model.train()
        model.set_epoch(epoch)
        self.batch_idx = 0
        # the reference builds grad_norms = [] inside train_epoch_fixed (train.py:698): its ">10 recorded norms" adaptive-clip rule
        # restarts every epoch.  Same here: history count (state[5]) and the device-side batch counter (state[19]) go back to 0
        # (two 4-byte device fills, no host sync).
        self.opt.state[5:6].zero_()
        self.opt.state[19:20].zero_()
        if scale_fn is not None:
            self.opt.set_lrs(scale_fn, epoch - 1)            # LambdaLR counts epochs from 0, the loop from 1

    def step(self, images, texts, modality_masks, labels) -> Dict[str, torch.Tensor]:
        bi = self.batch_idx
        outputs = self.module.forward(images=images, texts=texts, modality_masks=modality_masks, return_features=False)
        loss_dict = self.module.compute_loss(outputs, labels)
        (loss_dict['total_loss'] / self.accum_steps).backward()
        if (bi + 1) % self.accum_steps == 0:
            if self.dp is not None:
                self.dp.reduce_grads()
            self.opt.step(adaptive_clip=self.adaptive_clip, record_norm=(bi % NORM_EVERY == 0), zero_grad=True)
        self.batch_idx += 1
        return loss_dict


class GraphedStep:
    """The whole step of ``StepDriver`` (forward, losses, backward with its second stream, the three optimizer launches)
    captured ONCE as a HIP graph and replayed: ~900 kernel launches become one graph launch, which removes the host-side
    dispatch gaps between the many small kernels of the head and the step driver (measured with rocprofv3: 5-6 ms of idle
    GPU per 52 ms step in eager mode).

    NOTE: construction runs ``warmup`` REAL optimizer steps on the sample batch (lazy initialisation, routing plan, allocator
    pools) before the capture -- parameters, moments and the step count advance by that many steps.

    Constraints of a captured step: single process (collectives are left to the eager path), ``accum_steps == 1``, static
    shapes -- the batch layout AND the modality-mask pattern are part of the graph (masked rows are compacted on the host
    side of the routing plan), so ``step`` checks the pattern and refuses a different one (build another GraphedStep, or
    use the eager ``StepDriver`` for ragged batches).  Inputs are copied into the graph's static buffers on every call.
    """

    def __init__(self, driver: StepDriver, images, tokens, modality_masks, labels, warmup: int = 2):
        if driver.dp is not None and getattr(driver.dp, 'world', 1) > 1:
            raise ValueError('GraphedStep is single-process; use StepDriver for data-parallel runs')
        if driver.accum_steps != 1:
            raise ValueError('GraphedStep captures one optimizer step per micro-batch (accum_steps == 1)')
        if not isinstance(tokens, dict):
            raise ValueError('pre-tokenised text ({input_ids, attention_mask} on the device) is required')
        self.driver, self.opt = driver, driver.opt
        dev = self.opt.dev
        self.images = {m: t.to(dev).float().clone() for m, t in images.items()}
        self.tokens = {k: v.to(dev).clone() for k, v in tokens.items()}
        self.labels = labels.to(dev).clone()
        self.masks = {m: (t.detach().cpu() if torch.is_tensor(t) else torch.as_tensor(t)).float().clone()
                      for m, t in modality_masks.items()}
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):                    # eager warm-up: lazy initialisation, routing plan, allocator pools
            for _ in range(max(1, warmup)):
                self._run(False)
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        self.opt.state[16] = float(self.opt.step_count)  # device-side step / batch counters take over from the host's
        self.opt.state[19] = float(driver.batch_idx)
        self.graph = torch.cuda.CUDAGraph()
        # The training regularisers draw from this package's own device generators (DropPath per image, dropout multipliers):
        # registered with the graph, their Philox offsets advance on every replay, so each replayed step gets fresh masks.
        model = getattr(driver.module, 'model', driver.module)
        for gen in (getattr(model, '_rng_head', None), getattr(getattr(model, 'engine', None), '_rng', None)):
            if gen is not None and gen.device.type == 'cuda':
                self.graph.register_generator_state(gen)
        # Modality dropout decides on the HOST which modalities to drop (one draw each, as the reference does): a captured step
        # would replay the captured decision for ever.
        cfg = getattr(model, 'config', None)
        if (getattr(model, 'training', False) and float(getattr(cfg, 'modality_dropout', 0.0)) > 0
                and getattr(model, 'current_epoch', 0) > int(getattr(cfg, 'modality_dropout_warmup_epochs', 3))):
            raise ValueError('modality dropout is active (host-side draw per step): use the eager StepDriver')
        with torch.cuda.graph(self.graph):
            self.out = self._run(True)
        self.opt.step_count -= 1                         # the capture pass recorded the launches, it did not execute them
        driver.batch_idx -= 1

    def _run(self, device_counters: bool):
        d = self.driver
        outputs = d.module.forward(images=self.images, texts=self.tokens, modality_masks=self.masks, return_features=False)
        L = d.module.compute_loss(outputs, self.labels)
        L['total_loss'].backward()
        self.opt.step(adaptive_clip=d.adaptive_clip,
                      record_norm=NORM_EVERY if device_counters else (d.batch_idx % NORM_EVERY == 0),
                      zero_grad=True, device_counters=device_counters)
        d.batch_idx += 1
        return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in L.items()}

    def step(self, images, tokens, modality_masks, labels):
        for m, t in modality_masks.items():
            h = (t.detach().cpu() if torch.is_tensor(t) else torch.as_tensor(t)).float()
            if m in self.masks and not torch.equal(h, self.masks[m]):
                raise ValueError(f'modality mask of {m!r} differs from the captured pattern')
        for m, t in images.items():
            if t.data_ptr() != self.images[m].data_ptr():
                self.images[m].copy_(t, non_blocking=True)
        for k, v in tokens.items():
            if v.data_ptr() != self.tokens[k].data_ptr():
                self.tokens[k].copy_(v, non_blocking=True)
        if labels.data_ptr() != self.labels.data_ptr():
            self.labels.copy_(labels, non_blocking=True)
        self.graph.replay()
        self.opt.step_count += 1
        self.driver.batch_idx += 1
        for p in self.opt.params:
            torch.autograd.graph.increment_version(p)
        return self.out
