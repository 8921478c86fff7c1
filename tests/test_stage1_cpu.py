"""Stage-I retrieval, top-K file schema and CIRR test dicts: the CPU oracle and the host-side file code
against what the REAL reference produced (tests/golden/stage1_tiny.npz: features, the top-K files the
reference wrote itself, its submission dicts)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cir_oracle as O
from candidate_reranking_cir_amd import synthetic
from candidate_reranking_cir_amd import validate as V1
from tests import helpers as H


@pytest.fixture(scope="module")
def s1():
    z = H.load("stage1_tiny.npz")
    g, v = H.geometry(json.loads(str(z["bert_cfg"])), json.loads(str(z["vit_cfg"])))
    sd2, sd1 = H.state_dicts(g, v, int(z["seed"]), str(z["profile"]))
    with torch.no_grad():
        tokens, pooled = O.stage1_img_embed(sd1, synthetic.images(range(14), v.image_size))
    return z, g, v, sd2, sd1, tokens, pooled


def test_oracle_stage1_features_and_ranking(s1):
    z, g, v, sd2, sd1, tokens, pooled = s1
    np.testing.assert_allclose(pooled.numpy(), z["pooled"], atol=2e-5)
    np.testing.assert_allclose(tokens[:, :3, :8].numpy(), z["tokens_slice"], atol=2e-5)
    caps = [str(c) for c in z["cirr_caps"]]
    ids, mask = H.tokenize(caps)                                  # one padded batch (the reference batches 32)
    assert (mask == 0).any()
    with torch.no_grad():
        pred = O.stage1_query_features(sd1, tokens[torch.as_tensor(z["refs"])], ids, mask)
    np.testing.assert_allclose(pred.numpy(), z["cirr_pred"], atol=2e-5)
    rows = O.cirr_drop_reference(O.rank_index(torch.tensor(z["cirr_pred"]), torch.tensor(z["pooled"])), z["refs"])
    names = z["index_names"][rows[:, : int(z["k"])]]
    assert (names == z["cirr_file_names"]).all()                  # == the file the reference wrote


def test_topk_file_roundtrip_matches_reference_file(s1, tmp_path):
    z = s1[0]
    k = int(z["k"])
    names = [str(n) for n in z["index_names"]]
    ranked = O.rank_index(torch.tensor(z["cirr_pred"]), torch.tensor(z["pooled"]))
    group6 = np.concatenate([z["refs"][:, None], z["groups"]], axis=1)
    metrics, top = V1.cirr_topk(ranked, z["refs"], z["targets"], group6, names, k, "val")
    np.testing.assert_allclose(metrics, z["cirr_metrics"], atol=1e-4)
    assert (top["sorted_index_names"] == z["cirr_file_names"]).all()
    assert (top["labels"].numpy() == z["cirr_file_labels"]).all()
    assert (top["group_labels"].numpy() == z["cirr_file_group_labels"]).all() and top["split"] == str(z["cirr_file_split"])
    ranked_f = O.rank_index(torch.tensor(z["fiq_pred"]), torch.tensor(z["pooled"]))
    fm, ftop = V1.fiq_topk(ranked_f, z["targets"], names, k, "val", "dress")
    np.testing.assert_allclose(fm, z["fiq_metrics"], atol=1e-4)
    assert (ftop["sorted_index_names"] == z["fiq_file_names"]).all() and (ftop["labels"].numpy() == z["fiq_file_labels"]).all()
    assert ftop["target_names"] == [str(t) for t in z["fiq_file_targets"]] and ftop["dress_types"] == str(z["fiq_file_dress"])
    # write + read back into the stage-II dataset form
    path = str(tmp_path / "cirr_top.pt")
    V1.save_topk(path, top)
    ds = V1.load_topk(path, k - 1, z["refs"], captions=[str(c) for c in z["cirr_caps"]], group_index=z["groups"])
    assert ds.K == k - 1 and (z["index_names"][ds.cand_index] == z["cirr_file_names"][:, : k - 1]).all()
    assert (ds.labels == z["cirr_file_labels"][:, : k - 1]).all() and (ds.target_index == z["targets"]).all()


def test_oracle_cirr_test_dicts(s1):
    z, g, v, sd2, sd1, tokens1, pooled = s1
    names = z["index_names"]
    row_of = {str(n): i for i, n in enumerate(names)}
    cand_idx = np.vectorize(row_of.__getitem__)(z["cirr_file_names"])
    with torch.no_grad():
        feats2 = O.img_embed(sd2, synthetic.images(range(14), v.image_size))
        logits, glogits = [], []
        for q, cap in enumerate(z["cirr_caps"]):
            ids, mask = H.tokenize([str(cap)])
            out = O.score_queries(sd2, sd1, feats2, [int(z["refs"][q])], cand_idx[q:q + 1], np.ones((1, cand_idx.shape[1]), dtype=bool),
                                  ids, mask, group_index=z["groups"][q:q + 1])
            logits.append(out[0]); glogits.append(out[1])
    rec, sub = O.cirr_test_dicts(torch.cat(logits), torch.cat(glogits), names[cand_idx], names[z["groups"]], z["pair_ids"])
    assert json.dumps(rec, sort_keys=True) == str(z["test_recall_json"])
    assert json.dumps(sub, sort_keys=True) == str(z["test_subset_json"])


def test_submission_writer(tmp_path):
    from candidate_reranking_cir_amd import cirr_test_submission_stage2 as S
    p1, p2 = S.write_submissions(str(tmp_path), "t", {"12": ["a", "b"]}, {"12": ["c"]})
    assert json.load(open(p1)) == {"version": "rc2", "metric": "recall", "12": ["a", "b"]}
    assert json.load(open(p2)) == {"version": "rc2", "metric": "recall_subset", "12": ["c"]}
