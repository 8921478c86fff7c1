"""Stage-II scoring loop and Recall@k - the build's counterpart of the reference's
src/validate_stage2.py:33-298, batched over queries and shardable over GPUs.

Semantics kept from the reference (pinned by tests/golden/tiny_loop.npz, produced by the
reference's own loop):
  * a query is scored only if its `K_labels` row holds a positive, otherwise its row of logits is
    filled with -99999.99 (validate_stage2.py:95/123, 239/258);
  * candidates are scored in the order of the top-K file (`cand_index` row order) (:115, :251);
  * FashionIQ joins its two captions as "Cap1 and cap2" (:97-100) - `fiq_caption`;
  * CIRR additionally scores the 5 non-reference group members with the same z_t (:261-269);
  * metrics: argsort descending -> gather labels -> 100 * sum(labels[:, :k]) / Q (:53-62, :174-200).
What changes is the schedule: the reference runs one query per step (batch 1, :104, :242); here
`query_batch` queries go through stage I together and all their candidates form one stage-II
batch, z_t is computed once per query (the reference recomputes it for the CIRR subset, :264), and
names are integer rows of the index-feature bank (no Python dict of tensors, no O(Q^2) vstack).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from .blip_stage2 import encode_text

SKIP_FILL = -99999.99


@dataclass
class RelativeValSet:
    """Tensor form of a 'relative' validation split with its top-K file (data_utils.py:166-179,
    290-305): every name is an integer row of the index-feature bank."""
    ref_index: np.ndarray                 # (Q,)   reference image of each query
    cand_index: np.ndarray                # (Q, K) stage-I top-K candidates, best first
    labels: np.ndarray                    # (Q, K) bool, K_labels
    captions: Optional[List[str]] = None  # one caption per query (FashionIQ: already joined)
    input_ids: Optional[torch.Tensor] = None       # or pre-tokenised (Q, L)
    attention_mask: Optional[torch.Tensor] = None
    group_index: Optional[np.ndarray] = None       # (Q, 5) CIRR subset members without the reference
    target_index: Optional[np.ndarray] = None      # (Q,)   CIRR target_hard

    @property
    def K(self) -> int:
        return self.cand_index.shape[1]

    def __len__(self) -> int:
        return self.cand_index.shape[0]


def fiq_caption(cap1: str, cap2: str) -> str:
    """validate_stage2.py:97-100."""
    return f"{cap1.strip('.?, ').capitalize()} and {cap2.strip('.?, ')}"


@torch.no_grad()
def extract_index_features(images: torch.Tensor, model, batch_size: int = 64, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """ViT over the whole index (utils.py:43-55), sized from the model instead of the reference's
    hard-coded (n, 577, 768).  Returns the bank in the model's 16-bit compute dtype by default."""
    outs = []
    for i in range(0, images.shape[0], batch_size):
        chunk = images[i:i + batch_size].to(model.device)
        outs.append(model.img_embed(chunk) if dtype == torch.float32 else model.img_embed16(chunk))
    return torch.cat(outs)


def _tokenise(ds: RelativeValSet, tokenizer, rows: Sequence[int], length: int, device):
    if ds.input_ids is not None:
        sel = list(rows)
        enc = {"input_ids": ds.input_ids[sel][:, :length], "attention_mask": ds.attention_mask[sel][:, :length]}
        return encode_text(tokenizer, enc, device)
    return encode_text(tokenizer, [ds.captions[i] for i in rows], device)


def _length_buckets(ds: RelativeValSet, tokenizer, rows: Sequence[int]):
    """Group queries by token count so a batch never pads (the reference always runs batch 1 with
    padding='longest', i.e. no padding: blip_stage2.py:113).  Pre-tokenised input is right-padded."""
    if ds.input_ids is not None:
        lens = ds.attention_mask[list(rows)].sum(1).tolist()
    else:
        # (works for any tokenizer with the HF call convention: a real BertTokenizer returns plain lists for a single string)
        lens = [int(tokenizer([ds.captions[i]], padding="longest", return_tensors="pt").attention_mask.sum()) for i in rows]
    buckets = {}
    for r, n in zip(rows, lens):
        buckets.setdefault(int(n), []).append(r)
    return buckets


def _check_bank_rows(ds: RelativeValSet, n_index: int):
    """Names are integer rows of the index bank: validate them ONCE per dataset on the host (the device kernels clamp or
    trust their indices; a name the reference could not find raises KeyError in its dict lookup, validate_stage2.py:115)."""
    for what, arr in (("ref_index", ds.ref_index), ("cand_index", ds.cand_index), ("group_index", ds.group_index)):
        if arr is None or len(arr) == 0:
            continue
        a = np.asarray(arr)
        if a.min() < 0 or a.max() >= n_index:
            raise IndexError(f"{what} holds row {int(a.min() if a.min() < 0 else a.max())} outside the index bank of {n_index} images")


@torch.no_grad()
def generate_val_predictions(blip_model, model_stage1, ds: RelativeValSet, index_features: torch.Tensor,
                             query_batch: int = 8, rows: Optional[Sequence[int]] = None, kv_bank: Optional[list] = None):
    """Logits (len(rows), K) [and (len(rows), 5) subset logits when `ds.group_index` is set].
    `rows` selects a shard of the queries (default: all).  `kv_bank` (blip_model.build_kv_bank(index_features))
    re-uses the per-image cross-attention K/V across all queries instead of re-projecting every candidate."""
    dev = blip_model.device
    rows = list(range(len(ds))) if rows is None else list(rows)
    _check_bank_rows(ds, index_features.shape[0] if kv_bank is None else kv_bank[0].shape[0])
    pos = {r: i for i, r in enumerate(rows)}
    k = ds.K
    logits = torch.full((len(rows), k), SKIP_FILL, dtype=torch.float32, device=dev)
    glogits = torch.empty((len(rows), ds.group_index.shape[1]), dtype=torch.float32, device=dev) if ds.group_index is not None else None
    has_pos = ds.labels.any(axis=1)
    # which queries need a forward at all: positives in the top-K (skip rule) or a subset to score
    active = [r for r in rows if has_pos[r] or ds.group_index is not None]
    for length, bucket in sorted(_length_buckets(ds, blip_model.tokenizer, active).items()):
        for s in range(0, len(bucket), query_batch):
            qs = bucket[s:s + query_batch]
            ids, mask = _tokenise(ds, blip_model.tokenizer, qs, length, dev)
            ref = ops.gather_rows(index_features, torch.as_tensor(ds.ref_index[qs], device=dev))
            z = model_stage1.z_t(ref, ids, mask)                           # once per query
            cand_rows, qidx, slots = [], [], []
            for j, q in enumerate(qs):
                if has_pos[q]:
                    cand_rows.append(ds.cand_index[q]); qidx += [j] * k; slots.append((pos[q], 0, k))
                if ds.group_index is not None:
                    g = ds.group_index[q]
                    cand_rows.append(g); qidx += [j] * len(g); slots.append((pos[q], 1, len(g)))
            bank_rows = torch.as_tensor(np.concatenate(cand_rows), device=dev)
            if kv_bank is None:
                out = blip_model.score(z.last_hidden_state, ids, mask, ops.gather_rows(index_features, bank_rows),
                                       torch.as_tensor(qidx, device=dev))
            else:
                out = blip_model.score(z.last_hidden_state, ids, mask, None, torch.as_tensor(qidx, device=dev),
                                       kv_bank=kv_bank, cand_rows=bank_rows)
            o = 0
            for row, which, n in slots:
                (glogits if which else logits)[row] = out[o:o + n]
                o += n
    # One check at the end (a device reduction, no per-batch synchronisation): the fp16-stored residual stream has fp16's range;
    # a checkpoint whose pre-LayerNorm sums exceed 65504 would turn into inf -> NaN logits instead of an error.
    bad = ~torch.isfinite(logits).all() if glogits is None else ~(torch.isfinite(logits).all() & torch.isfinite(glogits).all())
    if bool(bad):
        raise FloatingPointError("non-finite logits: a value left fp16's range (65504) - in the 16-bit modes the residual stream's storage: call "
                                 "model.set_stream_dtype(torch.float32) on both models; in 'text32' (fp32 stream already) an fp16 operand term of "
                                 "the text side: call model.set_precision('exact') on both models (DESIGN.md section 2)")
    return (logits, glogits) if glogits is not None else logits


# ------------------------------------------------------------------------------------------------ reference signatures
def relative_val_set_from_dataset(relative_val_dataset, index_names: Sequence[str]):
    """The reference's 'relative' validation dataset (any object with its duck type: `K`, `K_labels`, `__len__`, and items
    laid out as data_utils.py:204-208 (FashionIQ: reference, target, [cap1, cap2], top-K names, K_labels) or :332-336 (CIRR:
    reference, target_hard, caption, 6 group members incl. the reference, top-K names, K_labels, K_group_labels)) ->
    (RelativeValSet, reference names, target names, group members without the reference or None).  Names become rows of
    `index_names` ONCE here (the reference looks every name up in a dict per query, validate_stage2.py:91/115, 232/251);
    a name that is not in the index raises KeyError, as there."""
    row = {str(n): i for i, n in enumerate(index_names)}
    if len(row) != len(index_names):
        raise ValueError("index_names holds duplicates")
    n_q, k = len(relative_val_dataset), int(relative_val_dataset.K)
    refs, targets, caps, members_no_ref = [], [], [], []
    ref_index, cand_index = np.empty(n_q, dtype=np.int64), np.empty((n_q, k), dtype=np.int64)
    labels = np.empty((n_q, k), dtype=bool)
    cirr = None
    for q in range(n_q):
        item = relative_val_dataset[q]
        if cirr is None:
            if len(item) not in (5, 7):
                raise TypeError(f"a stage-II relative-val item has 5 (FashionIQ) or 7 (CIRR) fields, got {len(item)} "
                                "(was the dataset built with load_topk= / K= ?)")
            cirr = len(item) == 7
        if cirr:
            ref, tgt, cap, members, k_names, k_lab, _ = item
            caps.append(str(cap))
            members_no_ref.append([str(m) for m in members if str(m) != str(ref)])         # validate_stage2.py:266, 274
        else:
            ref, tgt, cap, k_names, k_lab = item
            caps.append(fiq_caption(str(cap[0]), str(cap[1])))                             # validate_stage2.py:97-100
        refs.append(str(ref)); targets.append(str(tgt))
        ref_index[q] = row[str(ref)]
        cand_index[q] = [row[str(n)] for n in k_names]
        labels[q] = np.asarray(k_lab, dtype=bool)
    group_index = target_index = None
    if cirr:
        if any(len(m) != 5 for m in members_no_ref):
            raise ValueError("every CIRR subset has 5 members besides the reference (validate_stage2.py:186)")
        group_index = np.array([[row[m] for m in ms] for ms in members_no_ref], dtype=np.int64).reshape(n_q, 5)
        target_index = np.array([row[t] for t in targets], dtype=np.int64)
    ds = RelativeValSet(ref_index=ref_index, cand_index=cand_index, labels=labels, captions=caps, group_index=group_index, target_index=target_index)
    return ds, refs, targets, (members_no_ref if cirr else None)


def _bank16(blip_model, index_features: torch.Tensor) -> torch.Tensor:
    """The reference hands fp32 index features (utils.py:43-55); the scoring path reads the 16-bit bank: one conversion launch."""
    dt = getattr(blip_model, "token_dtype", blip_model.compute_dtype)
    feats = index_features.to(blip_model.device)
    return feats if feats.dtype == dt else ops.gather_rows(feats, None, dt)


def generate_fiq_val_predictions(blip_model, model_stage1, relative_val_dataset, *args, **kw):
    """Two call forms.  Native: (blip_model, model_stage1, RelativeValSet, index_features, query_batch=..., rows=..., kv_bank=...)
    -> logits (Q, K).  The reference's own (validate_stage2.py:69-71; what compute_fiq_val_metrics and stage2_train.py reach):
    (blip_model, model_stage1, relative_val_dataset, index_names, index_features) with the reference's dataset duck type
    -> (predicted_logits (Q, K), target_names), validate_stage2.py:129."""
    if isinstance(relative_val_dataset, RelativeValSet):
        return generate_val_predictions(blip_model, model_stage1, relative_val_dataset, *args, **kw)
    index_names, index_features = _names_and_features(args, kw)
    ds, _, targets, _ = relative_val_set_from_dataset(relative_val_dataset, index_names)
    logits = generate_val_predictions(blip_model, model_stage1, ds, _bank16(blip_model, index_features), **kw)
    return logits, targets


def generate_cirr_val_predictions(blip_model, model_stage1, relative_val_dataset, *args, **kw):
    """Native form as above -> (logits (Q, K), subset logits (Q, 5)).  Reference form (validate_stage2.py:209-211) ->
    (predicted_logits, group_predicted_logits, reference_names, target_names, group_members_noRef), validate_stage2.py:278."""
    if isinstance(relative_val_dataset, RelativeValSet):
        return generate_val_predictions(blip_model, model_stage1, relative_val_dataset, *args, **kw)
    index_names, index_features = _names_and_features(args, kw)
    ds, refs, targets, members = relative_val_set_from_dataset(relative_val_dataset, index_names)
    if ds.group_index is None:
        raise TypeError("generate_cirr_val_predictions needs CIRR items (7 fields, data_utils.py:332-336)")
    logits, glogits = generate_val_predictions(blip_model, model_stage1, ds, _bank16(blip_model, index_features), **kw)
    return logits, glogits, refs, targets, members


def _names_and_features(args, kw):
    """(index_names, index_features) of a reference-form call, positional or by keyword."""
    names = args[0] if len(args) > 0 else kw.pop("index_names")
    feats = args[1] if len(args) > 1 else kw.pop("index_features")
    if len(args) > 2:
        raise TypeError("reference form: (blip_model, model_stage1, relative_val_dataset, index_names, index_features)")
    if isinstance(names, torch.Tensor) or not isinstance(feats, torch.Tensor):
        raise TypeError("reference form takes index_names (list of str) BEFORE index_features (tensor), validate_stage2.py:69-71")
    return list(names), feats


# ------------------------------------------------------------------------------------------------ metrics
def sorted_labels(logits: torch.Tensor, labels: np.ndarray) -> torch.Tensor:
    """argsort(desc) on the GPU kernel, then np.take_along_axis like validate_stage2.py:53-57."""
    order = ops.argsort_desc(logits).cpu().numpy() if logits.is_cuda else torch.argsort(logits, dim=-1, descending=True, stable=True).numpy()
    return torch.tensor(np.take_along_axis(labels, order, axis=1))


def recall_at(lab: torch.Tensor, k: int) -> float:
    return (torch.sum(lab[:, :k]) / len(lab)).item() * 100


def compute_fiq_val_metrics(*args, **kw) -> Tuple[float, float]:
    """(R@10, R@50), validate_stage2.py:53-66.  Native form: (logits, RelativeValSet).  Reference form (validate_stage2.py:33-37,
    called stage2_train.py:270): (relative_val_dataset, blip_model, model_stage1, index_features, index_names)."""
    if isinstance(args[0], torch.Tensor):
        logits, ds = args
        lab = sorted_labels(logits, ds.labels)
        return recall_at(lab, 10), recall_at(lab, 50)
    relative_val_dataset, blip_model, model_stage1, index_features, index_names = _metric_args(args, kw)
    ds, _, _, _ = relative_val_set_from_dataset(relative_val_dataset, index_names)
    logits = generate_val_predictions(blip_model, model_stage1, ds, _bank16(blip_model, index_features), **kw)
    return compute_fiq_val_metrics(logits, ds)


def compute_cirr_val_metrics(*args, **kw):
    """(Rs@1, Rs@2, Rs@3, R@1, R@5, R@10, R@50), validate_stage2.py:174-206.  Native form: (logits, group_logits, RelativeValSet).
    Reference form (validate_stage2.py:153-156, called stage2_train.py:513): (relative_val_dataset, blip_model, model_stage1,
    index_features, index_names)."""
    if isinstance(args[0], torch.Tensor):
        logits, group_logits, ds = args
        lab = sorted_labels(logits, ds.labels)
        glab = sorted_labels(group_logits, ds.group_index == ds.target_index[:, None])
        return (recall_at(glab, 1), recall_at(glab, 2), recall_at(glab, 3),
                recall_at(lab, 1), recall_at(lab, 5), recall_at(lab, 10), recall_at(lab, 50))
    relative_val_dataset, blip_model, model_stage1, index_features, index_names = _metric_args(args, kw)
    ds, _, _, _ = relative_val_set_from_dataset(relative_val_dataset, index_names)
    if ds.group_index is None:
        raise TypeError("compute_cirr_val_metrics needs CIRR items (7 fields, data_utils.py:332-336)")
    logits, glogits = generate_val_predictions(blip_model, model_stage1, ds, _bank16(blip_model, index_features), **kw)
    return compute_cirr_val_metrics(logits, glogits, ds)


def _metric_args(args, kw):
    names = ("relative_val_dataset", "blip_model", "model_stage1", "index_features", "index_names")
    if len(args) > len(names):
        raise TypeError("reference form: (relative_val_dataset, blip_model, model_stage1, index_features, index_names)")
    vals = list(args) + [kw.pop(n) for n in names[len(args):]]
    return vals
