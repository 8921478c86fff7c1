"""Reference-signature entry points of validate_stage2 (host side, no GPU): the reference's dataset duck type -> RelativeValSet.

stage2_train.py:34 imports `compute_cirr_val_metrics, compute_fiq_val_metrics` from validate_stage2 and calls them as
(relative_val_dataset, blip_model, model_stage1, index_features, index_names) (:270, :513); generate_*_val_predictions take
(blip_model, model_stage1, relative_val_dataset, index_names, index_features) (validate_stage2.py:69-71, 209-211)."""
import numpy as np
import pytest
import torch

from candidate_reranking_cir_amd import validate_stage2 as V
from tests import helpers as H


def _case():
    z = H.load("tiny_loop.npz")
    names = ["n%03d" % (7 * i % 100) for i in range(14)]                 # names are NOT their row numbers
    return z, names


def test_dataset_adapter_fiq():
    z, names = _case()
    ds0 = H.DuckFIQ(names, z["refs"], z["targets"], z["fiq_caps"], z["cand_idx"], z["labels"])
    ds, refs, targets, members = V.relative_val_set_from_dataset(ds0, names)
    assert members is None and ds.group_index is None and ds.K == 6 and len(ds) == 8
    np.testing.assert_array_equal(ds.ref_index, z["refs"])
    np.testing.assert_array_equal(ds.cand_index, z["cand_idx"])
    np.testing.assert_array_equal(ds.labels, z["labels"])
    assert refs == [names[i] for i in z["refs"]] and targets == [names[i] for i in z["targets"]]
    assert ds.captions == [H.fiq_caption(p) for p in z["fiq_caps"]]       # "Cap1 and cap2", validate_stage2.py:97-100


@pytest.mark.parametrize("ref_slot", [0, 3, 5])
def test_dataset_adapter_cirr(ref_slot):
    z, names = _case()
    ds0 = H.DuckCIRR(names, z["refs"], z["targets"], z["cirr_caps"], z["cand_idx"], z["labels"], z["groups"], ref_slot)
    ds, refs, targets, members = V.relative_val_set_from_dataset(ds0, names)
    np.testing.assert_array_equal(ds.group_index, z["groups"])          # the reference image is dropped wherever it sits
    np.testing.assert_array_equal(ds.target_index, z["targets"])
    assert members == [[names[j] for j in row] for row in z["groups"]] and ds.captions == [str(c) for c in z["cirr_caps"]]


def test_reference_form_metrics_on_fixture_logits(monkeypatch):
    """compute_*_val_metrics in the reference's call form == the reference's recall tuples when the scorer returns the
    reference's logits (the scorer itself is checked on the GPU: tests/test_model_gpu.py::test_reference_signatures)."""
    z, names = _case()

    class Model:
        device, compute_dtype = torch.device("cpu"), torch.float32
    calls = []

    def fake_gen(blip_model, model_stage1, ds, index_features, **kw):
        calls.append(kw)
        if ds.group_index is not None:
            return torch.tensor(z["cirr_logits"]), torch.tensor(z["cirr_group_logits"])
        return torch.tensor(z["fiq_logits"])
    monkeypatch.setattr(V, "generate_val_predictions", fake_gen)
    feats = torch.zeros((14, 2, 4))
    fiq = H.DuckFIQ(names, z["refs"], z["targets"], z["fiq_caps"], z["cand_idx"], z["labels"])
    cirr = H.DuckCIRR(names, z["refs"], z["targets"], z["cirr_caps"], z["cand_idx"], z["labels"], z["groups"])
    np.testing.assert_allclose(V.compute_fiq_val_metrics(fiq, Model(), None, feats, names), z["fiq_metrics"], atol=1e-4)
    np.testing.assert_allclose(V.compute_cirr_val_metrics(cirr, Model(), None, index_features=feats, index_names=names, query_batch=3),
                               z["cirr_metrics"], atol=1e-4)
    assert calls[-1] == {"query_batch": 3}
    lg, tn = V.generate_fiq_val_predictions(Model(), None, fiq, names, feats)
    assert tn == [names[i] for i in z["targets"]] and lg.shape == (8, 6)
    lg, gl, rn, tn, mem = V.generate_cirr_val_predictions(Model(), None, cirr, names, feats)
    assert rn == [names[i] for i in z["refs"]] and len(mem) == 8 and all(len(m) == 5 for m in mem) and gl.shape == (8, 5)


def test_reference_form_argument_errors():
    z, names = _case()
    fiq = H.DuckFIQ(names, z["refs"], z["targets"], z["fiq_caps"], z["cand_idx"], z["labels"])

    class Model:
        device, compute_dtype = torch.device("cpu"), torch.float32
    with pytest.raises(TypeError):                                        # features and names swapped
        V.generate_fiq_val_predictions(Model(), None, fiq, torch.zeros((14, 2, 4)), names)
    with pytest.raises(KeyError):                                         # a top-K name missing from the index
        V.relative_val_set_from_dataset(fiq, names[:-1] + ["other"])
    with pytest.raises(TypeError):                                        # FashionIQ items into the CIRR entry point
        V.generate_cirr_val_predictions(Model(), None, fiq, names, torch.zeros((14, 2, 4)))
