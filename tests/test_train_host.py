"""Host-side logic of the training step that needs no GPU: slab ordering (stacked q|k|v projections), the row-chunk choice of
the weight gradients, flat-buffer detection of train.AdamW, bench.py's flop model."""
import torch

from candidate_reranking_cir_amd import config, weights
from candidate_reranking_cir_amd.train import AdamW, NlvrTrainer, _row_split


def _trained_names():
    g, v = config.BertGeometry(), config.VitGeometry(image_size=64, depth=1)
    names = list(weights.nlvr_param_spec(g, v))
    tr = NlvrTrainer.__new__(NlvrTrainer)
    return [n for n in names if tr._trained(n)], weights.nlvr_param_spec(g, v)


def test_trained_set_and_slab_order():
    names, spec = _trained_names()
    assert len(names) == 572                                     # the parameters the reference's step gives a gradient (train768.npz)
    assert not any("token_type" in n or "pooler" in n or n.startswith("visual_encoder.") for n in names)
    order = NlvrTrainer._order(names)
    assert sorted(order) == sorted(names) and len(set(order)) == len(order)
    pos = {n: i for i, n in enumerate(order)}
    for layer in (0, 6, 11):
        p = f"text_encoder.encoder.layer.{layer}."
        # the q | k | v of branch 0, then of branch 1 (one stacked Linear each, the twins a constant stride apart), then the biases alike
        i = pos[p + "attention.self0.query.weight"]
        assert [order[i + j] for j in range(12)] == [p + f"attention.self{b}.{x}.{y}" for y in ("weight", "bias") for b in (0, 1) for x in ("query", "key", "value")]
        i = pos[p + "crossattention.self0.key.weight"]
        assert [order[i + j] for j in range(8)] == [p + f"crossattention.self{b}.{x}.{y}" for y in ("weight", "bias") for b in (0, 1) for x in ("key", "value")]
        for stem in ("attention.output.dense", "crossattention.output.dense"):
            i = pos[p + stem + "0.weight"]
            assert [order[i + j] for j in range(4)] == [p + stem + f"{b}.{y}" for y in ("weight", "bias") for b in (0, 1)]
        i = pos[p + "crossattention.self0.query.weight"]
        assert [order[i + j] for j in range(4)] == [p + f"crossattention.self{b}.query.{y}" for y in ("weight", "bias") for b in (0, 1)]
    # every slice of a stacked group is a multiple of 8 elements: adjacency survives the slab's 8-element padding
    assert all(torch.Size(spec[n][0]).numel() % 8 == 0 for n in order if ".self" in n)


def test_row_split():
    assert _row_split(8192, 768, 768) == 16 and _row_split(9232, 768, 768) == 16          # 36 tiles x 16 chunks
    assert _row_split(8192, 3072, 768) == 8 and _row_split(8192, 2304, 768) == 8           # FFN / stacked q|k|v
    assert _row_split(176, 768, 768) == 1 and _row_split(16, 2, 768) == 1                   # too few rows to split
    for rows in (275, 1000, 8191, 12288):
        nb = _row_split(rows, 768, 768)
        assert rows % nb == 0 and (nb == 1 or rows // nb >= 128)


def test_adamw_flat_detection():
    flat = torch.zeros(8 * 5)
    a, b = flat[0:12].view(3, 4), flat[16:40].view(24)                                       # 12 -> padded 16, 24 -> 24: tiles the storage
    assert AdamW._flat_range([a, b]) == (flat.untyped_storage().data_ptr(), 40)
    assert AdamW._flat_range([a, torch.zeros(24)]) is None                                   # different storages
    assert AdamW._flat_range([a]) is None                                                     # does not cover the storage
    assert AdamW._flat_range([a, flat[16:40].view(4, 6).t()]) is None                        # not contiguous


def test_adamw_group_plan_and_gradient_match():
    """The cached flat layout of a parameter group and the per-step check that the gradients are slices of ONE buffer laid out alike
    (round 6: one pointer comparison per tensor instead of re-deriving the layout of 572 tensors three times per step)."""
    flat, gflat = torch.zeros(40), torch.zeros(40)
    ps = [torch.nn.Parameter(torch.zeros(1)) for _ in range(2)]
    ps[0].data, ps[1].data = flat[0:12].view(3, 4), flat[16:40].view(6, 4)
    opt = AdamW(ps, lr=1e-3)
    plan = opt._plan(ps)
    assert plan == (flat.untyped_storage().data_ptr(), 40, [0, 16]) and opt._plan(ps) is plan          # cached
    ps[0].grad, ps[1].grad = gflat[0:12].view(3, 4), gflat[16:40].view(6, 4)
    assert AdamW._grads_match(ps, plan) == gflat.data_ptr()
    ps[1].grad = torch.zeros(6, 4)                                                                     # a gradient from elsewhere
    assert AdamW._grads_match(ps, plan) is None
    ps[1].grad = gflat[16:40].view(4, 6).t()                                                           # right place, not contiguous
    assert AdamW._grads_match(ps, plan) is None
    other = torch.zeros(48)                                                                            # same offsets inside a LARGER buffer
    ps[0].grad, ps[1].grad = other[0:12].view(3, 4), other[16:40].view(6, 4)
    assert AdamW._grads_match(ps, plan) is None
    lone = [torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(7))]                     # separate storages: no plan
    assert AdamW(lone)._plan(lone) is None
    assert opt.t == 0 and opt.skipped_steps == 0                                                       # no device state before the first step


def test_stage1_dropout_site_seeds():
    """train_med.site_seed: one 62-bit counter-based seed per (call, layer, site) - distinct across the sites of a call, a pure function of
    its arguments (the host-side mask regeneration of tests/test_train_med_gpu.py depends on it)."""
    from candidate_reranking_cir_amd import train_med as M
    sites = (M.SITE_EMB, M.SITE_SELF_ATTN, M.SITE_SELF_OUT, M.SITE_CROSS_ATTN, M.SITE_CROSS_OUT, M.SITE_FFN_OUT)
    assert sites == (0, 1, 2, 3, 4, 5)
    seeds = {M.site_seed(12345, layer, s) for layer in range(12) for s in sites}
    assert len(seeds) == 72 and all(0 <= x < 2 ** 62 for x in seeds)
    assert M.site_seed(12345, 3, 2) == M.site_seed(12345, 3, 2) != M.site_seed(12346, 3, 2)
    assert M.site_seed(2 ** 62 - 1, 11, 5) < 2 ** 62


def test_train_flop_model():
    import bench
    b, l, n, d = 2, 8, 17, 768
    fwd, bwd = bench.train_gflop(b, l, n)
    r = b * b * l
    per_branch = 6 * 2 * r * d * d + 16 * r * d * d + 4 * r * l * d + 4 * r * n * d + 4 * (b * n) * d * d
    want = 12 * 2 * per_branch + 6 * 4 * r * d * d + 2 * b * b * (2 * d * d + 2 * d)
    assert abs(fwd * 1e9 - want) < 1e-6 * want and 1.8 * fwd < bwd < 2.0 * fwd


def test_cosine_lr_schedule_drives_adamw():
    """utils.cosine_lr_schedule's contract (utils.py:216-221): writes param_group['lr']; train.AdamW applies that value."""
    import math
    import torch
    from candidate_reranking_cir_amd.train import cosine_lr_schedule
    opt = AdamW([torch.nn.Parameter(torch.zeros(4))], lr=2e-5)
    assert opt.lr == 2e-5 and opt.param_groups[0]["lr"] == 2e-5
    lr = cosine_lr_schedule(opt, 5, 50, 2e-5, 0.0)
    assert lr == opt.lr == opt.param_groups[0]["lr"] and abs(lr - 1e-5 * (1 + math.cos(math.pi * 0.1))) < 1e-12
    assert cosine_lr_schedule(opt, 50, 50, 2e-5, 1e-6) == 1e-6
    ref = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(4))], lr=2e-5)
    assert cosine_lr_schedule(ref, 5, 50, 2e-5, 0.0) == lr and ref.param_groups[0]["lr"] == lr
