"""Query-level data parallelism for the stage-II scoring loop.

Queries are independent units (each carries its own K candidates), so they are split into contiguous
per-rank blocks with NO data-path collective; the only exchange is one all-gather of the per-rank
`(ceil(Q/world), K)` score blocks (and, optionally, their argsort indices) at the end.  The last
block is padded with skip rows (-99999.99) so every rank contributes an equally sized tensor.
Backend: `nccl` (= RCCL over xGMI) on GPUs, `gloo` in the CPU tests.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

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


def sharded_scores(score_rows: Callable[[Sequence[int]], torch.Tensor], n_queries: int, k: int,
                   device: torch.device, order: Optional[Sequence[int]] = None) -> torch.Tensor:
    """Run `score_rows(rows) -> (len(rows), k) fp32` on this rank's block and all-gather the result.

    Returns the full (n_queries, k) matrix on every rank, rows in dataset order.  `order` is an optional
    permutation (e.g. `balanced_order`) applied before blocking; it is undone after the gather."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    order = list(range(n_queries)) if order is None else list(order)
    lo, hi, per = shard_bounds(n_queries, rank, world)
    block = torch.full((per, k), SKIP_FILL, dtype=torch.float32, device=device)
    if hi > lo:
        block[: hi - lo] = score_rows(order[lo:hi]).to(device=device, dtype=torch.float32)
    if world == 1:
        gathered = block
    else:
        gathered = torch.empty((world * per, k), dtype=torch.float32, device=device)
        dist.all_gather_into_tensor(gathered, block)
    gathered = gathered[:n_queries]
    out = torch.empty_like(gathered)
    out[torch.as_tensor(order, dtype=torch.int64, device=device)] = gathered   # (int64 also when the list is empty)
    return out
