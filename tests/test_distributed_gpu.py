"""The N>1 path with the REAL scorer: two gloo ranks share the one GPU of the box, each runs
`validate_stage2.generate_val_predictions` (HIP kernels through the C ABI) on its block of queries via
`distributed.sharded_scores` (tuple of logits + CIRR subset logits, skip rows, `balanced_order`, gathered indices), and every
rank must end up with exactly the matrix a single process computes.  (RCCL refuses two ranks on one device, and gloo
gathers host tensors: the exchange goes through `device=cpu` here; on an 8-GPU node the same call runs with
backend nccl and `device=model.device`.)"""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup():
    from candidate_reranking_cir_amd import synthetic, validate_stage2 as V
    from tests import helpers as H
    from tests.test_model_gpu import build_models
    z = H.load("tiny_loop.npz")
    g, v = H.geometry(json.loads(str(z["bert_cfg"])), json.loads(str(z["vit_cfg"])))
    m2, m1 = build_models(g, v, int(z["seed"]), str(z["profile"]), torch.bfloat16, torch.device("cuda"))
    bank = V.extract_index_features(H.fixture_images(z, range(14), v.image_size), m2)
    ds = V.RelativeValSet(ref_index=z["refs"], cand_index=z["cand_idx"], labels=z["labels"], captions=[str(c) for c in z["cirr_caps"]],
                          group_index=z["groups"], target_index=z["targets"])
    return V, m2, m1, bank, ds


def _worker(rank, world, port, use_balance, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from candidate_reranking_cir_amd import distributed as D
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        V, m2, m1, bank, ds = _setup()
        order = D.balanced_order(ds.labels.any(axis=1).tolist()) if use_balance else None
        (logits, glogits), idx = D.sharded_scores(
            lambda rows: V.generate_val_predictions(m2, m1, ds, bank, query_batch=3, rows=rows),
            n_queries=len(ds), k=(ds.K, ds.group_index.shape[1]), device=torch.device("cpu"), order=order, with_indices=True)
        ret[rank] = (logits.numpy(), glogits.numpy(), idx.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_balance", [False, True])
def test_two_ranks_real_scorer_equals_single_process(use_balance):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    V, m2, m1, bank, ds = _setup()
    logits, glogits = V.generate_val_predictions(m2, m1, ds, bank, query_batch=3)
    expect, gexpect = logits.cpu().numpy(), glogits.cpu().numpy()
    skipped = ~ds.labels.any(axis=1)
    assert skipped.any() and np.all(expect[skipped] == np.float32(-99999.99))
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), use_balance, ret), nprocs=world, join=True)
    from candidate_reranking_cir_amd import ops
    eidx = ops.argsort_desc(logits).cpu().numpy()
    for r in range(world):
        a, b, idx = ret[r]
        # rows are independent and a row's arithmetic does not depend on its batch: bit-identical to the single-process run
        np.testing.assert_array_equal(a, expect)
        np.testing.assert_array_equal(b, gexpect)
        np.testing.assert_array_equal(idx, eidx)


def _rccl_worker(rank, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from candidate_reranking_cir_amd import distributed as D, ops
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    V, m2, m1, bank, ds = _setup()
    logits, glogits = V.generate_val_predictions(m2, m1, ds, bank, query_batch=3)      # no process group yet: plain path
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        order = D.balanced_order(ds.labels.any(axis=1).tolist())
        (a, b), idx = D.sharded_scores(lambda rows: V.generate_val_predictions(m2, m1, ds, bank, query_batch=3, rows=rows),
                                       n_queries=len(ds), k=(ds.K, ds.group_index.shape[1]), device=dev, order=order, with_indices=True)
        assert a.is_cuda and b.is_cuda and idx.is_cuda and idx.dtype == torch.int64
        ret["equal"] = bool(torch.equal(a, logits) and torch.equal(b, glogits) and torch.equal(idx, ops.argsort_desc(logits)))
        ret["backend"] = dist.get_backend()
    finally:
        dist.destroy_process_group()


def test_rccl_one_rank_group_gathers_device_tensors():
    """`sharded_scores(..., device=cuda, with_indices=True)` under an initialised NCCL (= RCCL) group of ONE rank: the
    communicator is created and both all_gather_into_tensor calls run on device tensors - the code path of the 8-GPU run."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ret = mp.Manager().dict()
    mp.spawn(_rccl_worker, args=(_free_port(), ret), nprocs=1, join=True)
    assert ret["backend"] == "nccl" and ret["equal"]
