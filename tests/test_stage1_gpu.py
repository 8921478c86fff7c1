"""Stage-I retrieval and CIRR test dicts on the HIP path, against the reference's
goldens (tests/golden/stage1_tiny.npz) and the CPU oracle."""
import json

import numpy as np
import pytest
import torch

from candidate_reranking_cir_amd import synthetic
from tests import helpers as H
from tests.test_model_gpu import build_models, margin_order_ok

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def s1():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    z = H.load("stage1_tiny.npz")
    g, v = H.geometry(json.loads(str(z["bert_cfg"])), json.loads(str(z["vit_cfg"])))
    m2, m1 = build_models(g, v, int(z["seed"]), str(z["profile"]), torch.float16, torch.device("cuda"))
    return z, g, v, m2, m1


def test_fp32_heads_and_ranking_kernels_exact(s1):
    """cir_linear_f32 (mode 2) + cir_topk_desc on the reference's own features reproduce the reference's ranking."""
    from candidate_reranking_cir_amd import ops, validate as V1
    z = s1[0]
    pred, pooled = torch.tensor(z["cirr_pred"]).cuda(), torch.tensor(z["pooled"]).cuda()
    ranked = V1.rank_index(pred, pooled).cpu().numpy()
    ref_dist = 1 - z["cirr_pred"] @ z["pooled"].T
    assert (np.sort(ref_dist, axis=1) == np.take_along_axis(ref_dist, ranked, axis=1)).all() or \
        np.allclose(np.sort(ref_dist, axis=1), np.take_along_axis(ref_dist, ranked, axis=1), atol=1e-6)
    group6 = np.concatenate([z["refs"][:, None], z["groups"]], axis=1)
    metrics, top = V1.cirr_topk(ranked, z["refs"], z["targets"], group6, [str(n) for n in z["index_names"]], int(z["k"]), "val")
    assert (top["sorted_index_names"] == z["cirr_file_names"]).all()
    np.testing.assert_allclose(metrics, z["cirr_metrics"], atol=1e-4)
    x = torch.randn(37, 50).cuda(); w = torch.randn(9, 50).cuda(); b = torch.randn(9).cuda()
    torch.testing.assert_close(ops.linear_f32(x, w, b), x @ w.T + b, atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(ops.l2_normalize(x), torch.nn.functional.normalize(x, dim=-1), atol=1e-6, rtol=1e-6)
    big = torch.randn(3, 6346).cuda()                       # FashionIQ shirt index size: needs the 8192-wide sort
    assert torch.equal(ops.argsort_desc(big).cpu(), torch.argsort(big.cpu(), dim=-1, descending=True, stable=True))


def test_stage1_features_and_topk_end_to_end(s1):
    from candidate_reranking_cir_amd import validate as V1
    z, g, v, m2, m1 = s1
    tokens, pooled = V1.extract_index_features(synthetic.images(range(14), v.image_size), m1)
    e_pool = np.abs(pooled.cpu().numpy() - z["pooled"]).max()
    pred = V1.generate_val_predictions(m1, z["refs"], [str(c) for c in z["cirr_caps"]], tokens)
    e_pred = np.abs(pred.cpu().numpy() - z["cirr_pred"]).max()
    from candidate_reranking_cir_amd.validate_stage2 import fiq_caption
    fpred = V1.generate_val_predictions(m1, z["refs"], [fiq_caption(str(a), str(b)) for a, b in z["fiq_caps"]], tokens)
    e_fpred = np.abs(fpred.cpu().numpy() - z["fiq_pred"]).max()
    print(f"\n[stage-I fp16] pooled {e_pool:.2e}  cirr query feature {e_pred:.2e}  fiq query feature {e_fpred:.2e}")
    assert e_pool < 2e-3 and e_pred < 2e-3 and e_fpred < 2e-3          # unit-norm 256-d features, fp16 operands
    ranked = V1.rank_index(pred, pooled).cpu().numpy()
    ref_dist = 1 - z["cirr_pred"] @ z["pooled"].T
    ours = 1 - pred.cpu().numpy() @ pooled.cpu().numpy().T
    for q in range(len(ranked)):                                        # order kept wherever the reference gap is real
        assert margin_order_ok(-ours[q], -ref_dist[q], 2e-3)


def test_cirr_test_dicts_hip_vs_reference(s1):
    """Submission dicts from the HIP path: identical to the reference's wherever its logit gaps exceed the fp16 bound."""
    from candidate_reranking_cir_amd import cirr_test_submission_stage2 as S, validate_stage2 as V
    from oracle import cir_oracle as O
    z, g, v, m2, m1 = s1
    names = [str(n) for n in z["index_names"]]
    row_of = {n: i for i, n in enumerate(names)}
    cand_idx = np.vectorize(row_of.__getitem__)(z["cirr_file_names"])
    bank = V.extract_index_features(synthetic.images(range(14), v.image_size), m2)
    ds = V.RelativeValSet(ref_index=z["refs"], cand_index=cand_idx, labels=np.zeros_like(cand_idx, dtype=bool),
                          captions=[str(c) for c in z["cirr_caps"]], group_index=z["groups"])
    rec, sub = S.generate_cirr_test_dicts(m2, m1, ds, bank, names, z["pair_ids"], query_batch=3)
    ref_rec, ref_sub = json.loads(str(z["test_recall_json"])), json.loads(str(z["test_subset_json"]))
    assert set(rec) == set(ref_rec) and all(sorted(rec[p]) == sorted(ref_rec[p]) for p in rec)     # same candidate sets
    # reference logits (oracle == reference to 2e-5) decide which positions have a real margin
    sd2, sd1 = H.state_dicts(g, v, int(z["seed"]), str(z["profile"]))
    agree = total = 0
    with torch.no_grad():
        feats = O.img_embed(sd2, synthetic.images(range(14), v.image_size))
        for q, cap in enumerate(z["cirr_caps"]):
            ids, mask = H.tokenize([str(cap)])
            lg, glg = O.score_queries(sd2, sd1, feats, [int(z["refs"][q])], cand_idx[q:q + 1], np.ones((1, cand_idx.shape[1]), dtype=bool),
                                      ids, mask, group_index=z["groups"][q:q + 1])
            pid = str(int(z["pair_ids"][q]))
            order = np.argsort(-lg[0].numpy(), kind="stable")
            gaps = np.abs(np.diff(lg[0].numpy()[order]))
            for pos in range(len(order)):
                clear = (pos == 0 or gaps[pos - 1] > 4e-4) and (pos == len(order) - 1 or gaps[pos] > 4e-4)
                if clear:
                    total += 1
                    agree += rec[pid][pos] == ref_rec[pid][pos]
    print(f"\n[test dicts] positions with a clear reference margin: {agree}/{total} identical")
    assert total > 0 and agree == total
    assert all(len(v) == 3 for v in sub.values()) and set(sub) == set(ref_sub)
