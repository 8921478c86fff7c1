"""Stage-I retrieval and the top-K file - the build's counterpart of the reference's src/validate.py
(SURVEY.md section 8(f) row 2): rank the whole index by cosine distance to the fused query feature, compute
Recall@k, and write / read the top-K file that stage II consumes.

Reference arithmetic (validate.py:57-64, 202-226): `distances = 1 - predicted @ index.T` ->
`argsort` ascending -> names; CIRR removes the reference image from each row, derives subset labels from
the 6-member groups, asserts exactly one positive per row.  File schema (validate.py:86-93, 255-262;
read back at data_utils.py:166-179, 290-305): `sorted_index_names (Q,K) str`, `target_names`, `index_names`,
`labels (Q,K) bool`, `split`, plus `dress_types` (FashionIQ) or `group_labels (Q,5)` (CIRR).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch

from . import ops
from .blip_stage2 import encode_text
from .validate_stage2 import RelativeValSet


@torch.no_grad()
def extract_index_features(images: torch.Tensor, model_stage1, batch_size: int = 64):
    """utils.py:57-72 (blip_stage1 branch): ViT tokens of every index image and their normalised 256-d pooled
    features.  Returns (tokens (n, N, D) in the compute dtype, pooled (n, 256) fp32)."""
    toks, pooled = [], []
    for i in range(0, images.shape[0], batch_size):
        t32, p = model_stage1.img_embed(images[i:i + batch_size].to(model_stage1.device), return_pool_and_normalized=True)
        toks.append(ops.gather_rows(t32, None, model_stage1.token_dtype))
        pooled.append(p)
    return torch.cat(toks), torch.cat(pooled)


@torch.no_grad()
def generate_val_predictions(model_stage1, ref_index: np.ndarray, captions: Sequence[str], index_tokens: torch.Tensor,
                             batch_size: int = 32) -> torch.Tensor:
    """validate.py:100-147 / 281-330: fused, normalised query features (Q, 256); captions are padded per batch of 32
    (`padding='longest'`), like the reference's DataLoader batches."""
    dev = model_stage1.device
    out = []
    for s in range(0, len(captions), batch_size):
        rows = list(range(s, min(s + batch_size, len(captions))))
        ids, mask = encode_text(model_stage1.tokenizer, [captions[i] for i in rows], dev)
        ref = ops.gather_rows(index_tokens, torch.as_tensor(ref_index[rows], device=dev))
        heads = model_stage1.engines()[2]
        z = model_stage1.z_t(ref, ids, mask)
        out.append(ops.l2_normalize(ops.linear_f32(z.last_hidden_state[:, 0, :], heads["tw"], heads["tb"])))
    return torch.cat(out)


@torch.no_grad()
def rank_index(predicted: torch.Tensor, index_pooled: torch.Tensor) -> torch.Tensor:
    """(Q, n_index) int64: index rows by ascending `1 - predicted @ index.T` (validate.py:57-58, 202-203)."""
    neg_dist = ops.linear_f32(predicted, index_pooled.contiguous(), None, mode=2)   # -(1 - q.i), exact negation
    return ops.argsort_desc(neg_dist)                                    # ascending distance; ties -> lower index


def recall_at(labels: np.ndarray, k: int) -> float:
    lab = torch.tensor(labels)
    return (torch.sum(lab[:, :k]) / len(lab)).item() * 100


def fiq_topk(sorted_rows: np.ndarray, target_index: np.ndarray, index_names: List[str], k: int, split: str, dress_type: str):
    """FashionIQ: labels, (R@10, R@50) and the top-K dict of validate.py:60-95."""
    labels = sorted_rows == target_index[:, None]
    assert (labels.sum(1) == 1).all()                                      # validate.py:64
    names = np.array(index_names)
    top = dict(sorted_index_names=names[sorted_rows[:, :k]], target_names=[index_names[i] for i in target_index],
               index_names=list(index_names), labels=torch.tensor(labels[:, :k]), split=split, dress_types=dress_type)
    return (recall_at(labels, 10), recall_at(labels, 50)), top


def cirr_topk(sorted_rows: np.ndarray, ref_index: np.ndarray, target_index: np.ndarray, group_index: np.ndarray,
              index_names: List[str], k: int, split: str):
    """CIRR: drop the reference image from each ranking, labels, subset labels, the 7 metrics and the top-K dict
    (validate.py:205-264).  `group_index` (Q, 6) holds the full groups incl. the reference, as in the dataset."""
    q_n, n_idx = sorted_rows.shape
    keep = sorted_rows != ref_index[:, None]                               # validate.py:207-210
    rows = sorted_rows[keep].reshape(q_n, n_idx - 1)
    labels = rows == target_index[:, None]
    group_mask = (rows[..., None] == group_index[:, None, :]).sum(-1).astype(bool)   # validate.py:219
    group_labels = labels[group_mask].reshape(q_n, -1)
    assert (labels.sum(1) == 1).all() and (group_labels.sum(1) == 1).all()           # validate.py:225-226
    names = np.array(index_names)
    top = dict(sorted_index_names=names[rows[:, :k]], target_names=[index_names[i] for i in target_index],
               index_names=list(index_names), labels=torch.tensor(labels[:, :k]), group_labels=torch.tensor(group_labels), split=split)
    metrics = (recall_at(group_labels, 1), recall_at(group_labels, 2), recall_at(group_labels, 3),
               recall_at(labels, 1), recall_at(labels, 5), recall_at(labels, 10), recall_at(labels, 50))
    return metrics, top


def save_topk(path: str, top: dict) -> None:
    torch.save(top, path)


def load_topk(path: str, k: int, ref_index: np.ndarray, captions: Optional[List[str]] = None,
              group_index: Optional[np.ndarray] = None, target_index: Optional[np.ndarray] = None) -> RelativeValSet:
    """Read a top-K file (ours or the authors') into the tensor form stage II consumes: names -> rows of `index_names`
    (data_utils.py:166-179, 290-305 keep the first K columns the same way)."""
    f = torch.load(path, weights_only=False)
    assert k <= f["sorted_index_names"].shape[-1]                           # data_utils.py:169, 293
    row_of = {n: i for i, n in enumerate(f["index_names"])}
    names = np.asarray(f["sorted_index_names"])[:, :k]
    cand = np.vectorize(row_of.__getitem__, otypes=[np.int64])(names)
    labels = np.asarray(f["labels"])[:, :k].astype(bool)
    if target_index is None:
        target_index = np.array([row_of[n] for n in f["target_names"]], dtype=np.int64)
    return RelativeValSet(ref_index=np.asarray(ref_index), cand_index=cand, labels=labels, captions=captions,
                          group_index=group_index, target_index=target_index)
