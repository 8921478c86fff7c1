"""`init_tokenizer` (blip.py:186-191) with a REAL WordPiece `BertTokenizer` built from a local vocabulary
(no network): [DEC] / [ENC] take the ids vocab_size / vocab_size + 1, callers overwrite ids[:, 0] with `enc_token_id`
(blip_stage2.py:113-114, blip_stage1.py:72-73), `padding='longest'` pads a ragged batch, and the batched validation loop
groups captions by token count with the real tokenizer's calling convention (single strings return plain lists)."""
import os

import numpy as np
import pytest
import torch

from candidate_reranking_cir_amd import blip, synthetic
from candidate_reranking_cir_amd import validate_stage2 as V
from candidate_reranking_cir_amd.blip_stage2 import encode_text

VOCAB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vocab_small.txt")


@pytest.fixture(scope="module")
def tok():
    return blip.init_tokenizer(vocab_file=VOCAB)


def test_small_vocab_special_tokens(tok):
    n = sum(1 for _ in open(VOCAB))
    assert tok.vocab_size == n
    assert tok.convert_tokens_to_ids("[DEC]") == n and tok.bos_token_id == n           # blip.py:188
    assert tok.convert_tokens_to_ids("[ENC]") == n + 1 == tok.enc_token_id              # blip.py:189-190
    assert (tok.pad_token_id, tok.unk_token_id, tok.cls_token_id, tok.sep_token_id) == (0, 100, 101, 102)   # bert-base-uncased layout


def test_wordpiece_and_longest_padding(tok):
    enc = tok(["Is red and has long sleeves", "the dress"], padding="longest", return_tensors="pt")
    ids, mask = enc.input_ids, enc.attention_mask
    assert ids.shape == mask.shape == (2, 9)                                             # [CLS] 6 words + '##s' [SEP]
    assert mask.tolist() == [[1] * 9, [1, 1, 1, 1, 0, 0, 0, 0, 0]]
    assert ids[0, 0] == 101 and ids[0, -1] == 102 and ids[1, 3] == 102 and ids[1, 4] == 0
    pieces = tok.convert_ids_to_tokens(ids[0].tolist())
    assert pieces[1:-1] == ["is", "red", "and", "has", "long", "sleeve", "##s"]         # lower-cased, word pieces
    assert tok.convert_ids_to_tokens(tok(["zzzz"], return_tensors="pt").input_ids[0].tolist())[1] == "[UNK]"


def test_encode_text_overwrites_first_id_with_enc(tok):
    ids, mask = encode_text(tok, ["the dress", "is red and has long sleeves"], "cpu")
    assert ids.dtype == torch.int64 and mask.dtype == torch.int64
    assert ids[:, 0].tolist() == [tok.enc_token_id] * 2                                  # blip_stage2.py:114
    assert ids[0, 3] == 102 and mask[0].sum() == 4 and mask[1].sum() == 9


def test_full_size_vocabulary_gives_the_reference_ids(tmp_path):
    """A 30522-entry vocabulary (size of bert-base-uncased; synthetic word pieces, generated here) reproduces the ids the
    reference's checkpoints were trained with: [DEC] -> 30522, [ENC] -> 30523 = the 30524-row embedding table."""
    toks = ["[PAD]"] + ["[unused%d]" % i for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    toks += ["[unused%d]" % i for i in range(99, 993)] + ["w%05d" % i for i in range(30522 - 104 - 894)]
    assert len(toks) == 30522
    (tmp_path / "vocab.txt").write_text("\n".join(toks) + "\n")
    tok = blip.init_tokenizer(vocab_file=str(tmp_path))                                  # a directory holding vocab.txt
    assert tok.vocab_size == 30522 and tok.bos_token_id == 30522 and tok.enc_token_id == 30523 == synthetic.ENC_TOKEN_ID
    assert len(tok) == 30524


def test_missing_vocabulary_is_loud(tmp_path):
    with pytest.raises(RuntimeError, match="not available offline"):
        blip.init_tokenizer(vocab_file=str(tmp_path / "nope.txt"))
    assert isinstance(blip.init_tokenizer(vocab_file=str(tmp_path / "nope.txt"), allow_fallback=True), synthetic.HashTokenizer)


def test_length_buckets_with_the_real_tokenizer(tok):
    """`tokenizer(one string)` of a real BertTokenizer returns python lists; the bucket code must not assume
    tensors.  Buckets = token counts incl. [CLS]/[SEP] and word pieces."""
    caps = ["the dress", "is red and has long sleeves", "a shirt", "red"]
    ds = V.RelativeValSet(ref_index=np.zeros(4, dtype=int), cand_index=np.zeros((4, 2), dtype=int), labels=np.ones((4, 2), dtype=bool), captions=caps)
    assert V._length_buckets(ds, tok, range(4)) == {4: [0, 2], 9: [1], 3: [3]}
    # and the synthetic double gives its own counts through the same code
    assert V._length_buckets(ds, synthetic.HashTokenizer(), range(4)) == {4: [0, 2], 8: [1], 3: [3]}


def test_models_have_no_silent_tokenizer_fallback():
    """With no cached bert-base-uncased the default tokenizer is None and strings are refused loudly; ids still work."""
    from candidate_reranking_cir_amd.blip_stage2 import default_tokenizer
    t = default_tokenizer()
    if t is not None:                                                                   # a populated HF cache: the real thing
        assert t.enc_token_id == 30523
        return
    with pytest.raises(RuntimeError, match="no tokenizer"):
        encode_text(None, ["a caption"], "cpu")
    ids, mask = encode_text(None, {"input_ids": torch.tensor([[101, 5, 102]]), "attention_mask": torch.ones(1, 3, dtype=torch.long)}, "cpu")
    assert ids[0, 0] == 30523
