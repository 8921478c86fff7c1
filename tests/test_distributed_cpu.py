"""N>1 path on CPU: world_size-2 gloo processes shard the queries, score their block with a stand-in
scorer and all-gather; the result must equal the single-process matrix (incl. skip rows and padding)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from candidate_reranking_cir_amd import distributed as D


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_scores(rows, k, active):
    """Deterministic stand-in for the GPU scorer: depends only on (query, candidate)."""
    out = torch.full((len(rows), k), D.SKIP_FILL)
    for i, q in enumerate(rows):
        if active[q]:
            out[i] = torch.sin(torch.arange(k, dtype=torch.float32) * 0.37 + q)
    return out


def _worker(rank, world, port, n_q, k, active, use_balance, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        order = D.balanced_order(active) if use_balance else None
        full = D.sharded_scores(lambda rows: _fake_scores(rows, k, active), n_q, k, torch.device("cpu"), order)
        # CIRR shape: (logits, subset logits) tuple, plus the gathered argsort of the first
        (a, b), idx = D.sharded_scores(lambda rows: (_fake_scores(rows, k, active), _fake_scores(rows, 5, active) * 2.0),
                                       n_q, (k, 5), torch.device("cpu"), order, with_indices=True)
        ret[rank] = (full.numpy(), a.numpy(), b.numpy(), idx.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_q,use_balance", [(7, False), (8, True), (1, False), (13, True)])
def test_two_rank_gather_equals_single_process(n_q, use_balance):
    k = 5
    rng = np.random.RandomState(n_q)
    active = (rng.rand(n_q) > 0.3).tolist()
    expect = _fake_scores(list(range(n_q)), k, active).numpy()
    world = 2
    ret = mp.Manager().dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_q, k, active, use_balance, ret), nprocs=world, join=True)
    expect_b = _fake_scores(list(range(n_q)), 5, active).numpy() * 2.0
    expect_idx = torch.argsort(torch.tensor(expect), dim=-1, descending=True, stable=True).numpy()
    for r in range(world):
        full, a, b, idx = ret[r]
        np.testing.assert_array_equal(full, expect)
        np.testing.assert_array_equal(a, expect)
        np.testing.assert_array_equal(b, expect_b)
        np.testing.assert_array_equal(idx, expect_idx)


def _fast_scores(rows, k, active, gain=1.0):
    """Vectorised stand-in scorer (depends only on (query, candidate)); skipped queries give the fill row."""
    q = torch.as_tensor(list(rows), dtype=torch.float32)[:, None]
    out = torch.sin(torch.arange(k, dtype=torch.float32)[None, :] * 0.37 + q) * gain
    act = torch.as_tensor([bool(active[r]) for r in rows])
    out[~act] = D.SKIP_FILL
    return out


# BASELINE configs[2] / [3] / [4] shapes and the degenerate "fewer queries than ranks" case: (queries, widths)
EIGHT_RANK_CASES = [(4181, (100, 5)), (6016, (100,)), (512, (200, 5)), (5, (100, 5))]


def _worker8(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        got = []
        for n_q, widths in EIGHT_RANK_CASES:
            active = (np.random.RandomState(n_q).rand(n_q) > 0.17).tolist()
            order = D.balanced_order(active)
            lo, hi, per = D.shard_bounds(n_q, rank, world)
            mine = sum(active[q] for q in order[lo:hi])
            if len(widths) == 1:
                full, idx = D.sharded_scores(lambda rows: _fast_scores(rows, widths[0], active), n_q, widths[0], torch.device("cpu"), order, with_indices=True)
                parts = (full,)
            else:
                parts, idx = D.sharded_scores(lambda rows: (_fast_scores(rows, widths[0], active), _fast_scores(rows, widths[1], active, 2.0)),
                                              n_q, widths, torch.device("cpu"), order, with_indices=True)
            # every rank checks the whole matrix itself (returning 8 x 6016 x 105 floats through a manager would dominate the test)
            ok = True
            for p, (w, gain) in zip(parts, zip(widths, (1.0, 2.0))):
                ok &= bool(torch.equal(p, _fast_scores(range(n_q), w, active, gain)))
            ok &= bool(torch.equal(idx, torch.argsort(parts[0], dim=-1, descending=True, stable=True)))
            got.append((ok, mine, hi - lo, per))
        ret[rank] = got
    finally:
        dist.destroy_process_group()


def test_eight_rank_gather_on_the_benchmark_configs():
    """SURVEY 8(e) at the rank count it names: EIGHT gloo processes, the query counts of BASELINE configs[2] (CIRR val, 4181 x (100 + 5)),
    configs[3] (FashionIQ, 6016 x 100), configs[4] (512 x (200 + 5)) and 5 queries on 8 ranks (three ranks own nothing), balanced
    order + indices: every rank ends with the single-process matrices bit for bit, blocks differ by at most 2 scored queries."""
    world = 8
    ret = mp.Manager().dict()
    mp.spawn(_worker8, args=(world, _free_port(), ret), nprocs=world, join=True)
    for ci, (n_q, widths) in enumerate(EIGHT_RANK_CASES):
        loads = [ret[r][ci][1] for r in range(world)]
        assert all(ret[r][ci][0] for r in range(world)), (n_q, widths)
        assert sum(ret[r][ci][2] for r in range(world)) == n_q and max(loads) - min(loads) <= 2, (n_q, loads)
        assert len({ret[r][ci][3] for r in range(world)}) == 1            # one padded block length: equal-sized contributions


def test_empty_and_tiny_query_sets_single_process():
    """No queries at all (an empty split) and fewer queries than ranks are valid inputs."""
    import torch
    out = D.sharded_scores(lambda rows: torch.zeros((len(rows), 3)), n_queries=0, k=3, device=torch.device("cpu"))
    assert out.shape == (0, 3)
    out = D.sharded_scores(lambda rows: torch.full((len(rows), 2), 7.0), n_queries=1, k=2, device=torch.device("cpu"))
    assert out.tolist() == [[7.0, 7.0]]
    (a, b), idx = D.sharded_scores(lambda rows: (torch.tensor([[1.0, 3.0, 2.0]]), torch.zeros((1, 5))), 1, (3, 5), torch.device("cpu"), with_indices=True)
    assert a.tolist() == [[1.0, 3.0, 2.0]] and b.shape == (1, 5) and idx.tolist() == [[1, 2, 0]]
    with pytest.raises(ValueError, match="expected widths"):          # a tuple where one tensor was declared
        D.sharded_scores(lambda rows: (torch.zeros((1, 3)), torch.zeros((1, 5))), 1, 3, torch.device("cpu"))


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 4181):
        for world in (1, 2, 8):
            seen = []
            for r in range(world):
                lo, hi, per = D.shard_bounds(n, r, world)
                assert hi - lo <= per
                seen += list(range(lo, hi))
            assert seen == list(range(n))


def test_balanced_order_is_a_permutation_and_balances():
    rng = np.random.RandomState(0)
    active = (rng.rand(4181) > 0.17).tolist()
    order = D.balanced_order(active)
    assert sorted(order) == list(range(len(active)))
    per = -(-len(active) // 8)
    loads = [sum(active[i] for i in order[r * per:(r + 1) * per]) for r in range(8)]
    assert max(loads) - min(loads) <= 2
