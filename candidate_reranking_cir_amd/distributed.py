"""Query-level data parallelism for the stage-II scoring loop.

Queries are independent units (each carries its own K candidates), so they are split into contiguous
per-rank blocks with NO data-path collective; the only exchange is one all-gather of the per-rank
`(ceil(Q/world), K [+ 5])` score blocks (and, with `with_indices`, one of their argsort indices) at the end.  The last
block is padded with skip rows (-99999.99) so every rank contributes an equally sized tensor.
Backend: `nccl` (= RCCL over xGMI) on GPUs, `gloo` in the CPU tests.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple, Union

import torch
import torch.distributed as dist

SKIP_FILL = -99999.99


def shard_bounds(n_queries: int, rank: int, world: int) -> Tuple[int, int, int]:
    """Contiguous block [lo, hi) of rank `rank` and the common padded block length."""
    per = -(-n_queries // world)
    lo = min(rank * per, n_queries)
    hi = min(lo + per, n_queries)
    return lo, hi, per


def balanced_order(active: Sequence[bool]) -> list:
    """Permutation that interleaves scored and skipped queries so that contiguous blocks carry about
    the same number of queries that actually need a forward pass (skip rows cost nothing)."""
    act = [i for i, a in enumerate(active) if a]
    idle = [i for i, a in enumerate(active) if not a]
    n = len(active)
    out, ia, ii = [], 0, 0
    for pos in range(n):
        # keep the running share of active queries proportional
        want_active = (ia * n) <= (pos * len(act)) if act else False
        if (want_active and ia < len(act)) or ii >= len(idle):
            out.append(act[ia]); ia += 1
        else:
            out.append(idle[ii]); ii += 1
    return out


def _argsort_desc(scores: torch.Tensor) -> torch.Tensor:
    """Per-row descending argsort, ties to the lower index: the HIP kernel on device tensors (cir_topk_desc), torch's
    stable sort for the CPU tests."""
    if scores.is_cuda:
        from . import ops
        return ops.argsort_desc(scores)
    return torch.argsort(scores, dim=-1, descending=True, stable=True)


def sharded_scores(score_rows: Callable[[Sequence[int]], Union[torch.Tensor, Tuple[torch.Tensor, ...]]], n_queries: int,
                   k: Union[int, Sequence[int]], device: torch.device, order: Optional[Sequence[int]] = None,
                   with_indices: bool = False):
    """Run `score_rows(rows)` on this rank's block of queries and all-gather the result.

    `score_rows(rows)` returns one fp32 tensor `(len(rows), k)` or a tuple of them `(len(rows), k_i)` - e.g.
    `generate_val_predictions` on a CIRR split returns `(logits (., K), subset logits (., 5))`; `k` is then the tuple of
    widths.  The blocks of a rank are laid side by side in ONE `(ceil(Q/world), sum k_i)` tensor padded with skip rows
    (-99999.99), so the exchange is a single all-gather (plus one for the indices when `with_indices`: the per-row
    descending argsort of the FIRST tensor, computed on the owning rank - SURVEY 8(e): scores and indices).
    Returns the full `(n_queries, k_i)` matrices on every rank in dataset order, in the structure `score_rows` uses
    [, and the `(n_queries, k_0)` int64 order].  `order` is an optional permutation (e.g. `balanced_order`) applied
    before blocking and undone after the gather."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    order = list(range(n_queries)) if order is None else list(order)
    widths = [int(k)] if isinstance(k, int) else [int(w) for w in k]
    lo, hi, per = shard_bounds(n_queries, rank, world)
    block = torch.full((per, sum(widths)), SKIP_FILL, dtype=torch.float32, device=device)
    if hi > lo:
        out = score_rows(order[lo:hi])
        parts = (out,) if isinstance(out, torch.Tensor) else tuple(out)
        if len(parts) != len(widths) or any(p.shape != (hi - lo, w) for p, w in zip(parts, widths)):
            raise ValueError(f"score_rows returned shapes {[tuple(p.shape) for p in parts]}, expected widths {widths} for {hi - lo} rows")
        block[: hi - lo] = torch.cat([p.to(device=device, dtype=torch.float32) for p in parts], dim=1)
    idx_block = _argsort_desc(block[:, : widths[0]].contiguous()) if with_indices else None
    if not dist.is_initialized():
        gathered, gathered_idx = block, idx_block
    else:       # (also with a one-rank group: the collective then still runs - RCCL communicator, device buffers - as on N ranks)
        gathered = torch.empty((world * per, sum(widths)), dtype=torch.float32, device=device)
        dist.all_gather_into_tensor(gathered, block)
        gathered_idx = None
        if with_indices:
            gathered_idx = torch.empty((world * per, widths[0]), dtype=torch.int64, device=device)
            dist.all_gather_into_tensor(gathered_idx, idx_block)
    inv = torch.as_tensor(order, dtype=torch.int64, device=device)              # (int64 also when the list is empty)
    full = torch.empty((n_queries, sum(widths)), dtype=torch.float32, device=device)
    full[inv] = gathered[:n_queries]
    pieces = tuple(full.split(widths, dim=1))
    result = pieces[0] if isinstance(k, int) else tuple(p.contiguous() for p in pieces)
    if not with_indices:
        return result
    full_idx = torch.empty((n_queries, widths[0]), dtype=torch.int64, device=device)
    full_idx[inv] = gathered_idx[:n_queries]
    return result, full_idx
