"""Seeded synthetic inputs (there are no datasets, vocabularies or checkpoints offline).

Shapes and statistics follow SURVEY.md section 8(d): post-normalisation images ~ N(0,1), captions
of L token ids with `[ENC]`-overwritten first id and `[SEP]`=102 last, label matrices shaped like
the reference's `K_labels` (data_utils.py:166-179, 290-305) with a fraction of rows holding no
positive (those rows exercise the skip rule of validate_stage2.py:95/123 and :239/258).
"""
from __future__ import annotations

import zlib
from typing import List, Sequence

import numpy as np
import torch

ENC_TOKEN_ID = 30523  # blip.py:188-190: '[DEC]' -> 30522, '[ENC]' -> 30523
CLS_ID, SEP_ID, PAD_ID = 101, 102, 0


def image(image_id: int, size: int = 224, chans: int = 3) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed(1000 + int(image_id))
    return torch.randn((chans, size, size), generator=g, dtype=torch.float32)


def images(ids: Sequence[int], size: int = 224) -> torch.Tensor:
    return torch.stack([image(i, size) for i in ids])


def scene_image(image_id: int, size: int = 224) -> torch.Tensor:
    """A structured synthetic image: per-image colour offset + three oriented sinusoidal gratings + a little noise.
    Unlike `image` (i.i.d. noise, statistically identical for every id) its ViT tokens carry a strong per-image
    signature, so candidates of one query receive well separated logits even with random-init weights - the inputs
    of the rank-order fixtures (tests/golden/rank224.npz).  Pure numpy float64 arithmetic, then one cast."""
    import math
    rng = np.random.RandomState(7000 + int(image_id))
    lin = np.linspace(-1.0, 1.0, size)
    yy, xx = np.meshgrid(lin, lin, indexing="ij")
    img = np.zeros((3, size, size), dtype=np.float64) + rng.randn(3, 1, 1) * 1.5
    for _ in range(3):
        theta, freq, phase = rng.rand() * math.pi, rng.uniform(1.0, 12.0), rng.rand() * 2.0 * math.pi
        amp = rng.randn(3, 1, 1) * 0.6
        img += amp * np.sin(freq * math.pi * (xx * math.cos(theta) + yy * math.sin(theta)) + phase)
    return torch.from_numpy(img.astype(np.float32)) + 0.2 * image(image_id, size)


def scene_images(ids: Sequence[int], size: int = 224) -> torch.Tensor:
    return torch.stack([scene_image(i, size) for i in ids])


def caption_ids(query_id: int, length: int = 32) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed(2000 + int(query_id))
    ids = torch.randint(1000, 30000, (length,), generator=g, dtype=torch.int64)
    ids[0] = ENC_TOKEN_ID
    ids[-1] = SEP_ID
    return ids


def label_matrix(n_queries: int, k: int, seed: int = 0, miss_rate: float = 0.1) -> np.ndarray:
    """(Q, K) bool, at most one positive per row, rank geometric-ish, `miss_rate` rows empty."""
    rng = np.random.RandomState(seed)
    lab = np.zeros((n_queries, k), dtype=bool)
    for q in range(n_queries):
        if rng.rand() < miss_rate:
            continue
        lab[q, min(int(rng.geometric(min(1.0, 4.0 / k))) - 1, k - 1)] = True
    return lab


class _Encoding:
    def __init__(self, input_ids: torch.Tensor, attention_mask: torch.Tensor):
        self.input_ids = input_ids
        self.attention_mask = attention_mask

    def to(self, device):
        self.input_ids = self.input_ids.to(device)
        self.attention_mask = self.attention_mask.to(device)
        return self


class HashTokenizer:
    """Deterministic stand-in for the reference's `BertTokenizer` (blip.py:186-191).

    `bert-base-uncased`'s WordPiece vocabulary cannot be fetched offline, so tests and the
    benchmark tokenise with a word hash: `[CLS] h(w1) ... h(wn) [SEP]`, ids in [1000, 30000),
    `padding='longest'` semantics (pad id 0, mask 0).  It exposes what the reference's callers
    use: `__call__(text, padding=..., return_tensors='pt')` -> `.input_ids/.attention_mask/.to()`
    and `.enc_token_id` (blip_stage2.py:113-114).  A real `BertTokenizer` is a drop-in.
    """

    enc_token_id = ENC_TOKEN_ID

    def __init__(self, max_length: int = 512):
        self.max_length = max_length

    @staticmethod
    def _word_id(word: str) -> int:
        return 1000 + zlib.crc32(word.lower().encode()) % 29000

    def encode(self, text: str) -> List[int]:
        words = text.split()[: self.max_length - 2]
        return [CLS_ID] + [self._word_id(w) for w in words] + [SEP_ID]

    def __call__(self, text, padding="longest", return_tensors="pt"):
        if isinstance(text, str):
            text = [text]
        rows = [self.encode(t) for t in text]
        width = max(len(r) for r in rows)
        ids = torch.full((len(rows), width), PAD_ID, dtype=torch.int64)
        mask = torch.zeros((len(rows), width), dtype=torch.int64)
        for i, r in enumerate(rows):
            ids[i, : len(r)] = torch.tensor(r)
            mask[i, : len(r)] = 1
        return _Encoding(ids, mask)


def caption_text(query_id: int, n_words: int = 30) -> str:
    """A caption of `n_words` pseudo-words (-> n_words + 2 tokens under HashTokenizer)."""
    rng = np.random.RandomState(3000 + int(query_id))
    return " ".join("w%05d" % rng.randint(0, 99999) for _ in range(n_words))
