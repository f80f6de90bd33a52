"""Text -> token ids for the CLIP text tower.

The reference tokenises on the host every step with
``CLIPTokenizer.from_pretrained(NAME)`` (models/clip_backbone.py:171,297-303):
``padding=True, truncation=True, max_length=77, return_tensors='pt'``; pad id =
EOS id (49407), BOS 49406.  The BPE vocabulary is a download and is not present
offline, so this module provides

* :class:`HashTokenizer` -- same call signature and output layout
  (``input_ids``/``attention_mask`` int64 ``[B, T]``, T = longest in batch,
  BOS ... EOS then EOS padding), word ids from a CRC of the whitespace token.
  It is what "random-token text" (BASELINE.json config 1) means here and what
  the golden fixtures were generated with;
* :func:`load_tokenizer` -- uses a real ``CLIPTokenizer`` when ``name_or_path``
  is a local directory holding vocab.json/merges.txt, else HashTokenizer.

The model also accepts pre-tokenised input (a dict with ``input_ids`` and
``attention_mask``) so a caller can tokenise once, off the hot path.
"""
import os
import zlib
from typing import Dict, List, Sequence, Union

import torch


class HashTokenizer:
    def __init__(self, vocab_size: int = 49408, bos_id: int = None, eos_id: int = None, max_length: int = 77):
        self.vocab_size = vocab_size
        self.bos_token_id = vocab_size - 2 if bos_id is None else bos_id
        self.eos_token_id = vocab_size - 1 if eos_id is None else eos_id
        self.pad_token_id = self.eos_token_id
        self.model_max_length = max_length

    def word_id(self, w: str) -> int:
        return 1 + zlib.crc32(w.encode('utf-8')) % (min(self.bos_token_id, self.eos_token_id) - 1)

    def encode(self, text: str, max_length: int) -> List[int]:
        ids = [self.word_id(w) for w in text.strip().split()]
        ids = ids[:max(0, max_length - 2)]
        return [self.bos_token_id] + ids + [self.eos_token_id]

    def __call__(self, texts: Union[str, Sequence[str]], return_tensors: str = 'pt', padding=True,
                 truncation: bool = True, max_length: int = None) -> Dict[str, torch.Tensor]:
        if isinstance(texts, str):
            texts = [texts]
        L = max_length or self.model_max_length
        rows = [self.encode(t, L) for t in texts]
        T = L if padding == 'max_length' else max(len(r) for r in rows)
        ids = torch.full((len(rows), T), self.pad_token_id, dtype=torch.int64)
        am = torch.zeros((len(rows), T), dtype=torch.int64)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = torch.tensor(r, dtype=torch.int64)
            am[i, :len(r)] = 1
        return {'input_ids': ids, 'attention_mask': am}


def load_tokenizer(name_or_path: str, vocab_size: int = 49408, bos_id: int = None, eos_id: int = None,
                   max_length: int = 77):
    if name_or_path and os.path.isdir(name_or_path) and \
            os.path.exists(os.path.join(name_or_path, 'vocab.json')):
        from transformers import CLIPTokenizer  # local files only; never a hub fetch
        return CLIPTokenizer.from_pretrained(name_or_path, local_files_only=True)
    return HashTokenizer(vocab_size, bos_id, eos_id, max_length)
