"""Input side of the hot path: the strict P x K batch sampler and the collate that produce
exactly what ``model.forward`` / ``compute_loss`` consume, plus a feeder that stages batches into HBM off the step.

Reference: ``ModalAwarePKBatchSampler_Strict`` (datasets/dataset.py:1327-1464), ``compatible_collate_fn`` (:1467-1606),
``infer_modalities_of_sample`` (:188-255), ``canon_mod`` (:62-76).  The sampler reproduces the reference's draw sequence
(same calls to ``random`` in the same order), so with the same seed it yields the SAME index lists -- pinned by
tests/golden/pipeline_cases.json, which make_golden.py writes by executing the reference's own code.
The dataset class itself (file layout of ORBench, PIL decoding, torchvision transforms) is out of scope: anything that
yields the reference's sample dictionaries works.
"""
import random as _random
import threading
import queue
from typing import Any, Dict, Iterable, Iterator, List, Optional, Sequence, Set

import torch

CANON_DS = {'vis': 'vis', 'rgb': 'vis', 'visible': 'vis', 'v': 'vis', 'nir': 'nir', 'ir': 'nir', 'infrared': 'nir',
            'sk': 'sk', 'sketch': 'sk', 'cp': 'cp', 'cpencil': 'cp', 'colorpencil': 'cp', 'coloredpencil': 'cp',
            'txt': 'text', 'text': 'text', 'caption': 'text'}
IMG_MODALITIES = {'vis', 'nir', 'sk', 'cp'}
ALL_MODALITIES = IMG_MODALITIES | {'text'}
MODALITIES = ['vis', 'nir', 'sk', 'cp']


def canon_mod(name) -> str:
    """dataset.py:72-76."""
    if name is None:
        return ''
    k = str(name).lower().strip()
    return CANON_DS.get(k, k)


def _truthy(x) -> bool:
    """dataset.py:78-93."""
    if x is None:
        return False
    if isinstance(x, (list, tuple, set, dict)):
        return len(x) > 0
    if isinstance(x, bool):
        return x
    if isinstance(x, (int, float)):
        return float(x) > 0.5
    if isinstance(x, str):
        return len(x.strip()) > 0
    if torch.is_tensor(x):
        return int(x.nelement()) > 0 and (bool(x.abs().sum() > 1e-6) if x.dtype.is_floating_point else True)
    return True


def infer_modalities(sample: dict, include_text: bool = True) -> Set[str]:
    """Modalities a sample dictionary provides (dataset.py:224-255)."""
    mods = set()
    mm = sample.get('modality_mask') or sample.get('modal_mask') or sample.get('mods')
    if isinstance(mm, dict):
        for k, v in mm.items():
            m = canon_mod(k)
            if m in IMG_MODALITIES and _truthy(v):
                mods.add(m)
    imgs = sample.get('images') or sample.get('paths') or sample.get('imgs')
    if isinstance(imgs, dict):
        for k, v in imgs.items():
            m = canon_mod(k)
            if m in IMG_MODALITIES and _truthy(v):
                mods.add(m)
    primary = sample.get('modality') or sample.get('mode') or sample.get('mod')
    if primary:
        m = canon_mod(primary)
        if m in IMG_MODALITIES:
            mods.add(m)
    if include_text:
        if _truthy(sample.get('text_description')) or _truthy(sample.get('text')) or _truthy(sample.get('caption')):
            mods.add('text')
        elif isinstance(imgs, dict) and _truthy(imgs.get('text')):
            mods.add('text')
    return {m for m in mods if m in (ALL_MODALITIES if include_text else IMG_MODALITIES)}


class StrictPKBatchSampler:
    """P identities x K instances per batch, K//2 drawn from the identity's vis samples and the rest from its non-vis ones
    (dataset.py:1327-1464).  ``samples``: the dataset's sample dictionaries (``person_id`` / ``pid`` / ``label`` + the
    fields ``infer_modalities`` reads).  ``rng``: a ``random.Random`` (default: the global ``random`` module, like the
    reference)."""

    def __init__(self, samples: Sequence[dict], num_ids_per_batch: int = 4, num_instances: int = 4, allow_id_reuse: bool = True,
                 indices: Optional[Sequence[int]] = None, rng=None):
        self.P, self.K = int(num_ids_per_batch), int(num_instances)
        self.allow_id_reuse = bool(allow_id_reuse)
        self.rng = rng if rng is not None else _random
        self.indices = list(range(len(samples))) if indices is None else list(indices)
        self.pid_to_mod_idxs: Dict[int, Dict[str, List[int]]] = {}
        self.pids = set()
        for i in self.indices:
            s = samples[i]
            pid = int(s.get('person_id') or s.get('pid') or s.get('label') or -1) if isinstance(s, dict) else -1
            if pid < 0:
                continue
            self.pids.add(pid)
            mods_img = infer_modalities(s, include_text=False)
            mods_all = infer_modalities(s, include_text=True)
            d = self.pid_to_mod_idxs.setdefault(pid, {'vis': [], 'nonvis': []})
            if 'vis' in mods_img:
                d['vis'].append(i)
            if bool(mods_img & {'nir', 'sk', 'cp'}) or ('text' in mods_all):
                d['nonvis'].append(i)
        self.strong_ids = [pid for pid, d in self.pid_to_mod_idxs.items() if d['vis'] and d['nonvis']]
        self.soft_ids = [pid for pid in self.pids if pid not in self.strong_ids]
        total = sum(min(len(self.pid_to_mod_idxs[p]['vis']), len(self.pid_to_mod_idxs[p]['nonvis'])) for p in self.strong_ids)
        self._len_est = max(1, total // max(1, self.P * self.K))

    def __len__(self):
        return int(self._len_est) if self.allow_id_reuse else max(1, len(self.strong_ids) // self.P)

    def __iter__(self) -> Iterator[List[int]]:
        R = self.rng
        strong_pool, soft_pool = list(self.strong_ids), list(self.soft_ids)
        while True:
            if len(strong_pool) >= self.P:
                cur = R.sample(strong_pool, self.P) if not self.allow_id_reuse else R.choices(strong_pool, k=self.P)
            else:
                need = self.P - len(strong_pool)
                fill = (R.sample(soft_pool, min(need, len(soft_pool))) if not self.allow_id_reuse
                        else R.choices(soft_pool, k=need)) if soft_pool else []
                cur = list(strong_pool) + fill
                if not cur:
                    break
            out: List[int] = []
            for pid in cur:
                d = self.pid_to_mod_idxs.get(pid, {'vis': [], 'nonvis': []})
                vis_pool = d['vis'] if d['vis'] else d['nonvis']
                non_pool = d['nonvis'] if d['nonvis'] else d['vis']
                k_vis = self.K // 2
                k_non = self.K - k_vis
                out += R.sample(vis_pool, k_vis) if len(vis_pool) >= k_vis else R.choices(vis_pool or non_pool, k=k_vis)
                out += R.sample(non_pool, k_non) if len(non_pool) >= k_non else R.choices(non_pool or vis_pool, k=k_non)
            if len(out) != self.P * self.K:
                continue
            yield out
            if not self.allow_id_reuse:
                for pid in set(cur):
                    if pid in strong_pool:
                        strong_pool.remove(pid)
                    elif pid in soft_pool:
                        soft_pool.remove(pid)
                if len(strong_pool) < 1 and len(soft_pool) < 1:
                    break


def collate(batch: List[dict], image_size: int = 224) -> Dict[str, Any]:
    """``compatible_collate_fn`` (dataset.py:1467-1606): person_id [B], text_description list, images {m: [B,3,H,W]} with zero
    placeholders, modality_mask {m: f32 [B]} = (image present AND non-zero AND the sample's own mask > 0.5) / non-empty text,
    and the canonical primary modality of every sample."""
    if not batch:
        return {}
    first = batch[0]
    out: Dict[str, Any] = {}
    if 'person_id' in first:
        out['person_id'] = torch.stack([torch.as_tensor(s['person_id']) for s in batch])
    texts = []
    if 'text_description' in first:
        for s in batch:
            td = s.get('text_description', [''])
            texts.append(td[0] if isinstance(td, list) and len(td) > 0 else (td if isinstance(td, str) else ''))
    elif 'text_descriptions' in first:
        for s in batch:
            td = s.get('text_descriptions', [''])
            texts.append(td[0] if isinstance(td, list) and len(td) > 0 else '')
    else:
        texts = [''] * len(batch)
    out['text_description'] = texts
    real = {m: [] for m in MODALITIES + ['text']}
    for s in batch:
        for m in MODALITIES:
            ok = False
            imgs = s.get('images')
            if isinstance(imgs, dict) and isinstance(imgs.get(m), torch.Tensor):
                t = imgs[m]
                ok = t.numel() > 0 and bool(t.abs().sum() > 1e-6)
            mm = s.get('modality_mask')
            if isinstance(mm, dict):
                o = mm.get(m, 0.0)
                if isinstance(o, bool):
                    ok = ok and o
                elif isinstance(o, (float, int)):
                    ok = ok and (float(o) > 0.5)
            real[m].append(1.0 if ok else 0.0)
        td = s.get('text_description', s.get('text_descriptions', ['']))
        if isinstance(td, list):
            ok = len(td) > 0 and isinstance(td[0], str) and len(td[0].strip()) > 0
        elif isinstance(td, str):
            ok = len(td.strip()) > 0
        else:
            ok = False
        real['text'].append(1.0 if ok else 0.0)
    images = {}
    if isinstance(first.get('images'), dict):
        for m in MODALITIES:
            ts = []
            for s in batch:
                t = s['images'].get(m) if isinstance(s.get('images'), dict) else None
                ts.append(t if isinstance(t, torch.Tensor) else torch.zeros(3, image_size, image_size))
            images[m] = torch.stack(ts)
    else:
        for m in MODALITIES:
            if isinstance(first.get(m), torch.Tensor):
                images[m] = torch.stack([s[m] for s in batch])
    out['images'] = images
    out['modality_mask'] = {m: torch.tensor(real[m], dtype=torch.float) for m in MODALITIES + ['text']}

    def primary(s):
        if 'modality' in s:
            v = s['modality']
        elif 'mod' in s:
            v = s['mod']
        else:
            best, bm = None, -1
            mm = s.get('modality_mask')
            if isinstance(mm, dict):
                for name, val in mm.items():
                    val = float(val) if not isinstance(val, bool) else (1.0 if val else 0.0)
                    if val > bm:
                        bm, best = val, name
            v = best if best else s.get('meta', {}).get('modality', 'vis')
        return canon_mod(str(v))
    out['modality'] = [primary(s) for s in batch]
    return out


class DeviceFeeder:
    """Background producer: sampler -> dataset[i] -> collate -> tokeniser -> pinned staging -> HBM on a copy stream.

    Yields batches whose ``images`` / ``tokens`` / ``person_id`` are device tensors ready for ``StepDriver.step`` (masks stay
    on the host: the model's routing plan reads them there).  ``depth`` batches are prepared ahead; a batch's H2D copies are
    ordered before its consumer by an event the consumer stream waits on, so the training stream never waits for the host.
    """

    def __init__(self, dataset, batch_sampler: Iterable[List[int]], tokenizer, device, depth: int = 2, max_length: int = 77):
        self.dataset, self.sampler, self.tok = dataset, batch_sampler, tokenizer
        self.dev = torch.device(device)
        self.max_length = max_length
        self.q: "queue.Queue" = queue.Queue(maxsize=max(1, depth))
        self.copy_stream = torch.cuda.Stream(self.dev) if self.dev.type == 'cuda' else None
        self._t = threading.Thread(target=self._work, daemon=True)
        self._started = False

    def _stage(self, t: torch.Tensor) -> torch.Tensor:
        if self.copy_stream is None:
            return t
        return t.pin_memory().to(self.dev, non_blocking=True)

    def _work(self):
        try:
            for idxs in self.sampler:
                b = collate([self.dataset[i] for i in idxs])
                tok = self.tok(b['text_description'], return_tensors='pt', padding=True, truncation=True, max_length=self.max_length)
                ev = None
                if self.copy_stream is not None:
                    with torch.cuda.stream(self.copy_stream):
                        b['images'] = {m: self._stage(t.float()) for m, t in b['images'].items()}
                        b['tokens'] = {k: self._stage(v) for k, v in tok.items()}
                        b['person_id'] = self._stage(b['person_id'])
                        ev = torch.cuda.Event(); ev.record(self.copy_stream)
                else:
                    b['tokens'] = dict(tok)
                self.q.put((b, ev))
        finally:
            self.q.put(None)

    def __iter__(self):
        if not self._started:
            self._t.start(); self._started = True
        while True:
            item = self.q.get()
            if item is None:
                return
            b, ev = item
            if ev is not None:
                torch.cuda.current_stream(self.dev).wait_event(ev)
            yield b
