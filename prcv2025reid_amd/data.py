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


# ---------------------------------------------------------------------------------------------------------------------
# What a sample provides.
#
# Behaviour pinned by the reference (dataset.py:78-93, 224-255) and restated here as DATA: which dictionary fields are
# probed, in which order, and what "has content" means per value type.  The probing itself is one generic loop.

# value type -> "has content" predicate, first match wins (bool is an int in Python, so it takes the numeric rule:
# True -> 1.0 > 0.5)
_CONTENT_RULES = (
    (type(None), lambda v: False),
    ((list, tuple, set, dict), lambda v: len(v) > 0),
    ((int, float), lambda v: float(v) > 0.5),
    (str, lambda v: bool(v.strip())),
    (torch.Tensor, lambda v: v.nelement() > 0 and (not v.dtype.is_floating_point or bool(v.abs().sum() > 1e-6))),
)


def has_content(value) -> bool:
    for types, rule in _CONTENT_RULES:
        if isinstance(value, types):
            return bool(rule(value))
    return True                                              # unknown objects (paths, PIL images, ...) count as present


# sample fields, in probing order; within a family the first field holding a truthy value is the one that is read
_FIELDS = {
    'mask': ('modality_mask', 'modal_mask', 'mods'),        # {modality name: flag}
    'images': ('images', 'paths', 'imgs'),                   # {modality name: tensor / path / list}
    'primary': ('modality', 'mode', 'mod'),                  # one modality name
    'text': ('text_description', 'text', 'caption'),         # caption(s)
}
_PID_FIELDS = ('person_id', 'pid', 'label')


def _probe(sample: dict, family: str):
    for key in _FIELDS[family]:
        value = sample.get(key)
        if value:
            return value
    return None


def infer_modalities(sample: dict, include_text: bool = True) -> Set[str]:
    """Canonical modalities a sample dictionary provides: every image modality named by a non-empty entry of its mask or
    image container or by its primary-modality field, plus 'text' when a caption field (or ``images['text']``) has content."""
    named: List[str] = []
    containers = [_probe(sample, 'mask'), _probe(sample, 'images')]
    for box in containers:
        if isinstance(box, dict):
            named += [key for key, value in box.items() if has_content(value)]
    primary = _probe(sample, 'primary')
    if primary:
        named.append(primary)
    found = {canon_mod(n) for n in named} & IMG_MODALITIES
    if include_text:
        captioned = any(has_content(sample.get(key)) for key in _FIELDS['text'])
        if captioned or (isinstance(containers[1], dict) and has_content(containers[1].get('text'))):
            found.add('text')
    return found


def _identity_of(sample) -> int:
    if not isinstance(sample, dict):
        return -1
    for key in _PID_FIELDS:
        value = sample.get(key)
        if value:                                            # (a zero / missing id falls through to the next field)
            return int(value)
    return -1


# ---------------------------------------------------------------------------------------------------------------------
# Strict P x K batches.
#
# Contract (dataset.py:1327-1464, pinned by pipeline_cases.json): a batch is P identities x K instances; every identity
# contributes K // 2 indices from its "has a vis image" list and K - K // 2 from its "has anything else" list (a list that
# is empty borrows the other one); identities that own both lists are preferred, the rest only fill up.  The DRAW
# SEQUENCE on the random generator is part of the contract (same seed -> same index lists as the reference): one draw for
# the identities, then per identity one draw per side, without replacement when the list is long enough.
#
# Here the bookkeeping is a small table of identities built in one pass and a per-batch draw plan; the iterator only
# executes plans.

class _Identity:
    """Index lists of one person: samples showing a vis image / samples showing any other modality (incl. text)."""
    __slots__ = ('pid', 'vis', 'other')

    def __init__(self, pid: int):
        self.pid, self.vis, self.other = pid, [], []

    @property
    def paired(self) -> bool:
        return bool(self.vis) and bool(self.other)

    def sides(self):
        """(list the vis quota is drawn from, list the non-vis quota is drawn from): an empty side borrows the other."""
        return (self.vis or self.other), (self.other or self.vis)


def _draw(rng, population: Sequence[int], count: int, replace: bool) -> List[int]:
    return rng.choices(population, k=count) if replace else rng.sample(population, count)


class StrictPKBatchSampler:
    """Yields index lists of length P*K.  ``samples``: the dataset's sample dictionaries (identity under ``person_id`` /
    ``pid`` / ``label`` + the fields ``infer_modalities`` reads); ``indices``: the subset to draw from (default all);
    ``rng``: a ``random.Random`` (default: the global ``random`` module, which is what the reference draws from).

    ``paired_ids`` / ``unpaired_ids`` list the identities with / without both sides, in first-seen / set order (the order
    the identity draw sees).  Deliberate difference: when fewer than P identities can be chosen (too few identities, no filler
    left) the reference retries the same short batch forever -- how many identities a draw returns does not depend on the
    random numbers -- whereas this iterator stops there."""

    def __init__(self, samples: Sequence[dict], num_ids_per_batch: int = 4, num_instances: int = 4, allow_id_reuse: bool = True,
                 indices: Optional[Sequence[int]] = None, rng=None):
        self.P, self.K = int(num_ids_per_batch), int(num_instances)
        self.quota = (self.K // 2, self.K - self.K // 2)     # (vis side, other side): the odd instance goes to the other side
        self.reuse = bool(allow_id_reuse)
        self.rng = _random if rng is None else rng
        self.indices = list(range(len(samples))) if indices is None else list(indices)
        self.table: Dict[int, _Identity] = {}
        seen = set()                                         # a set, so the unpaired order below is the reference's set order
        for i in self.indices:
            pid = _identity_of(samples[i])
            if pid < 0:
                continue
            seen.add(pid)
            who = self.table.setdefault(pid, _Identity(pid))
            image_mods = infer_modalities(samples[i], include_text=False)
            if 'vis' in image_mods:
                who.vis.append(i)
            if (image_mods - {'vis'}) or 'text' in infer_modalities(samples[i], include_text=True):
                who.other.append(i)
        self.paired_ids = [pid for pid, who in self.table.items() if who.paired]
        self.unpaired_ids = [pid for pid in seen if not self.table[pid].paired]
        pairable = sum(min(len(self.table[pid].vis), len(self.table[pid].other)) for pid in self.paired_ids)
        self._epoch_batches = max(1, pairable // max(1, self.P * self.K)) if self.reuse else max(1, len(self.paired_ids) // self.P)

    def __len__(self) -> int:
        return self._epoch_batches

    # -- one batch = one plan -------------------------------------------------------------------------------------
    def _pick_identities(self, paired: List[int], unpaired: List[int]) -> List[int]:
        if len(paired) >= self.P:
            return _draw(self.rng, paired, self.P, self.reuse)
        missing = self.P - len(paired)
        if not unpaired:
            return list(paired)
        return list(paired) + _draw(self.rng, unpaired, missing if self.reuse else min(missing, len(unpaired)), self.reuse)

    def _pick_instances(self, who: _Identity) -> List[int]:
        picked: List[int] = []
        for side, fallback, count in zip(who.sides(), reversed(who.sides()), self.quota):
            enough = len(side) >= count
            picked += _draw(self.rng, side if enough else (side or fallback), count, replace=not enough)
        return picked

    def __iter__(self) -> Iterator[List[int]]:
        paired, unpaired = list(self.paired_ids), list(self.unpaired_ids)
        while paired or unpaired:
            chosen = self._pick_identities(paired, unpaired)
            batch = [i for pid in chosen for i in self._pick_instances(self.table[pid])]
            if len(batch) != self.P * self.K:
                return                                       # (see the class docstring)
            yield batch
            if not self.reuse:                               # every identity at most once per epoch
                for pid in set(chosen):
                    (paired if pid in paired else unpaired).remove(pid)


# ---------------------------------------------------------------------------------------------------------------------
# Collate: list of sample dictionaries -> the batch dictionary model.forward / compute_loss consume (dataset.py:1467-1606).
# Validity is computed for the whole batch at once: one stacked |x| reduction per modality instead of one per sample.

def _caption_of(sample: dict, key: str, accept_plain: bool) -> str:
    value = sample.get(key, [''])
    if isinstance(value, list) and value:
        return value[0]
    return value if (accept_plain and isinstance(value, str)) else ''


def _caption_present(sample: dict) -> bool:
    value = sample.get('text_description', sample.get('text_descriptions', ['']))
    if isinstance(value, list):
        value = value[0] if value else None
    return isinstance(value, str) and bool(value.strip())


def _own_flag(sample: dict, modality: str) -> bool:
    """The sample's own say about a modality: a missing entry vetoes, a bool / number decides, anything else abstains."""
    own = sample.get('modality_mask')
    if not isinstance(own, dict):
        return True
    flag = own.get(modality, 0.0)
    if isinstance(flag, (bool, int, float)):
        return float(flag) > 0.5
    return True


def _primary_name(sample: dict) -> str:
    for key in ('modality', 'mod'):
        if key in sample:
            return canon_mod(str(sample[key]))
    own = sample.get('modality_mask')
    best = None
    if isinstance(own, dict) and own:
        weights = {name: float(flag) for name, flag in own.items()}
        top = max(weights.values())
        if top > -1:
            best = next(name for name, w in weights.items() if w == top)      # first entry holding the maximum
    return canon_mod(str(best if best else sample.get('meta', {}).get('modality', 'vis')))


def collate(batch: List[dict], image_size: int = 224) -> Dict[str, Any]:
    """person_id [B], text_description (list of str), images {m: [B,3,H,W]} with zero placeholders for missing images,
    modality_mask {m: f32 [B]} -- an image counts iff its tensor is non-zero (sum |x| > 1e-6) AND the sample's own mask
    agrees; text iff the caption is non-blank -- and the canonical primary modality of every sample."""
    if not batch:
        return {}
    first, B = batch[0], len(batch)
    out: Dict[str, Any] = {}
    if 'person_id' in first:
        out['person_id'] = torch.stack([torch.as_tensor(s['person_id']) for s in batch])
    caption_key = next((k for k in ('text_description', 'text_descriptions') if k in first), None)
    out['text_description'] = [''] * B if caption_key is None else \
        [_caption_of(s, caption_key, accept_plain=caption_key == 'text_description') for s in batch]

    nested = [s['images'] if isinstance(s.get('images'), dict) else {} for s in batch]
    given = {m: [box.get(m) if isinstance(box.get(m), torch.Tensor) else None for box in nested] for m in MODALITIES}
    mask: Dict[str, torch.Tensor] = {}
    for m in MODALITIES:
        lit = torch.zeros(B, dtype=torch.bool)
        rows = [b for b, t in enumerate(given[m]) if t is not None and t.numel() > 0]
        for shape in {given[m][b].shape for b in rows}:      # (one reduction per distinct shape: normally exactly one)
            same = [b for b in rows if given[m][b].shape == shape]
            energy = torch.stack([given[m][b] for b in same]).abs().flatten(1).sum(1)
            lit[same] = energy > 1e-6
        agree = torch.tensor([_own_flag(s, m) for s in batch])
        mask[m] = (lit & agree).float()
    mask['text'] = torch.tensor([_caption_present(s) for s in batch]).float()

    if isinstance(first.get('images'), dict):
        blank = torch.zeros(3, image_size, image_size)
        out['images'] = {m: torch.stack([blank if t is None else t for t in given[m]]) for m in MODALITIES}
    else:                                                    # images stored at the root of the sample dictionaries
        out['images'] = {m: torch.stack([s[m] for s in batch]) for m in MODALITIES if isinstance(first.get(m), torch.Tensor)}
    out['modality_mask'] = mask
    out['modality'] = [_primary_name(s) for s in batch]
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
