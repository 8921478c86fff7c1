"""CIRR test-split submission through stage II - counterpart of the reference's
src/cirr_test_submission_stage2.py.

No labels on the test split: every query is scored (no skip rule, :111-178); the server files are
`{"version": "rc2", "metric": "recall", pair_id: [50 names]}` and
`{"version": "rc2", "metric": "recall_subset", pair_id: [3 names]}` (:50-71, :92-108), written with sort_keys.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from . import ops
from .validate_stage2 import RelativeValSet, generate_val_predictions


@torch.no_grad()
def generate_cirr_test_dicts(blip_model, model_stage1, ds: RelativeValSet, index_features: torch.Tensor, index_names: Sequence[str],
                             pair_ids: Sequence[int], query_batch: int = 8, kv_bank=None) -> Tuple[Dict[str, List[str]], Dict[str, List[str]]]:
    """Top-50 global and top-3 subset predictions per pair id (cirr_test_submission_stage2.py:74-108).
    `ds.labels` is ignored: all queries are scored; `ds.group_index` holds the 5 non-reference members."""
    all_true = RelativeValSet(ref_index=ds.ref_index, cand_index=ds.cand_index, labels=np.ones_like(ds.cand_index, dtype=bool),
                              captions=ds.captions, input_ids=ds.input_ids, attention_mask=ds.attention_mask,
                              group_index=ds.group_index, target_index=ds.target_index)
    logits, glogits = generate_val_predictions(blip_model, model_stage1, all_true, index_features, query_batch=query_batch, kv_bank=kv_bank)
    names = np.asarray(index_names)
    order = ops.argsort_desc(logits).cpu().numpy()
    sorted_names = np.take_along_axis(names[ds.cand_index], order, axis=1)
    gorder = ops.argsort_desc(glogits).cpu().numpy()
    sorted_group = np.take_along_axis(names[ds.group_index], gorder, axis=1)
    rec = {str(int(p)): row[:50].tolist() for p, row in zip(pair_ids, sorted_names)}
    sub = {str(int(p)): row[:3].tolist() for p, row in zip(pair_ids, sorted_group)}
    return rec, sub


def write_submissions(folder: str, file_name: str, pairid_to_predictions: dict, pairid_to_group_predictions: dict) -> Tuple[str, str]:
    """cirr_test_submission_stage2.py:50-71."""
    submission = {"version": "rc2", "metric": "recall"}
    group_submission = {"version": "rc2", "metric": "recall_subset"}
    submission.update(pairid_to_predictions)
    group_submission.update(pairid_to_group_predictions)
    os.makedirs(folder, exist_ok=True)
    p1 = os.path.join(folder, f"recall_submission_{file_name}.json")
    p2 = os.path.join(folder, f"recall_subset_submission_{file_name}.json")
    with open(p1, "w+") as fh:
        json.dump(submission, fh, sort_keys=True)
    with open(p2, "w+") as fh:
        json.dump(group_submission, fh, sort_keys=True)
    return p1, p2
