"""Per-query error of the exact (fp32) mode and of the default mode on tests/golden/outlier224_wide.npz (round 5 diagnostic):
which queries carry the 4e-3 disagreement between two fp32 implementations (the reference's CPU BLAS order vs the f32 MFMA chain)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from scipy.stats import kendalltau
from candidate_reranking_cir_amd import synthetic, validate_stage2 as V
from tests import helpers as H
from tests.test_model_gpu import build_models

z = H.load(sys.argv[1] if len(sys.argv) > 1 else "outlier224_wide.npz")
g, v = H.geometry(H.FULL_BERT, dict(image_size=224))
dev = torch.device("cuda")
m2, m1 = build_models(g, v, int(z["seed"]), str(z["profile"]), torch.float16, dev)
imgs = synthetic.scene_images(range(int(z["n_index"])), 224)
ds = V.RelativeValSet(ref_index=z["refs"], cand_index=z["cand"], labels=z["labels"], captions=[str(c) for c in z["caps"]],
                      group_index=z["groups"], target_index=z["targets"])
ref = z["logits"]; act = z["labels"].any(1)
out = {}
for mode in ("exact", "f16"):
    for m in (m2, m1):
        m.set_precision(mode)
    bank = V.extract_index_features(imgs, m2, batch_size=64)
    lt, gt = V.generate_cirr_val_predictions(m2, m1, ds, bank, query_batch=4)
    lg = lt.cpu().numpy()
    rows = []
    for q in np.where(act)[0]:
        e = lg[q] - ref[q]
        rows.append(dict(q=int(q), max_abs=float(np.abs(e).max()), mean_shift=float(e.mean()), centred_max=float(np.abs(e - e.mean()).max()),
                         sigma=float(ref[q].std()), lo=float(ref[q].min()), hi=float(ref[q].max()), tau=float(kendalltau(lg[q], ref[q]).statistic),
                         exact=float((np.argsort(-lg[q], kind="stable") == np.argsort(-ref[q], kind="stable")).mean())))
        print(mode, rows[-1], flush=True)
    out[mode] = rows
print(json.dumps(out))
