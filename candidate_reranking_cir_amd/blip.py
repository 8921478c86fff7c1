"""Shared helpers with the reference's names (src/blip.py): `init_tokenizer` and `create_vit`.

`init_tokenizer` (blip.py:186-191) needs the `bert-base-uncased` WordPiece vocabulary, which cannot be downloaded in an
offline build: pass a local directory holding `vocab.txt` (or rely on a populated HuggingFace cache); without either the deterministic
`synthetic.HashTokenizer` double is returned when `allow_fallback=True` (what tests and the benchmark use).
"""
from __future__ import annotations

import os
from typing import Optional

from .config import VitGeometry
from .synthetic import HashTokenizer


def init_tokenizer(vocab_file: Optional[str] = None, allow_fallback: bool = False):
    """BertTokenizer + '[DEC]' (bos) + '[ENC]' with `.enc_token_id`, exactly as blip.py:186-191.

    `vocab_file` is a WordPiece `vocab.txt` (or a directory holding one); without it the `bert-base-uncased` entry of
    the local HuggingFace cache is used (no download is attempted).  The two added tokens take the ids vocab_size and
    vocab_size + 1 (30522 / 30523 for bert-base-uncased)."""
    try:
        from transformers import BertTokenizer
        if vocab_file is not None:
            path = os.path.join(vocab_file, "vocab.txt") if os.path.isdir(vocab_file) else vocab_file
            if not os.path.isfile(path):
                raise FileNotFoundError(path)
            import inspect
            if "vocab_file" in inspect.signature(BertTokenizer.__init__).parameters:          # transformers 4.x (reference pins 4.25)
                tok = BertTokenizer(vocab_file=path, do_lower_case=True)                      # bert-base-uncased settings
            else:                                                                             # transformers 5.x: token -> id mapping
                with open(path, encoding="utf-8") as fh:
                    vocab = {line.rstrip("\n"): i for i, line in enumerate(fh)}
                tok = BertTokenizer(vocab=vocab, do_lower_case=True)
        else:
            tok = BertTokenizer.from_pretrained("bert-base-uncased", local_files_only=True)
            if tok.vocab_size < 30000:   # transformers >= 5 builds an EMPTY tokenizer instead of failing when nothing is cached
                raise FileNotFoundError("bert-base-uncased vocabulary (30522 entries) not found in the local cache")
    except Exception as exc:  # no vocabulary available offline
        if allow_fallback:
            return HashTokenizer()
        raise RuntimeError("bert-base-uncased vocabulary not available offline: pass vocab_file=... "
                           "(or allow_fallback=True for the synthetic HashTokenizer)") from exc
    tok.add_special_tokens({"bos_token": "[DEC]"})
    tok.add_special_tokens({"additional_special_tokens": ["[ENC]"]})
    tok.enc_token_id = tok.convert_tokens_to_ids("[ENC]")   # == additional_special_tokens_ids[0] in transformers 4.25
    return tok


def create_vit(vit: str, image_size: int, use_grad_checkpointing: bool = False, ckpt_layer: int = 0, drop_path_rate: float = 0):
    """Geometry of the reference's ViT factory (blip.py:194-209); gradient checkpointing / DropPath are training-only."""
    g = VitGeometry.named(vit, image_size)
    return g, g.width
