"""`img_embed` in .train() mode WITHOUT a graph (frozen ViT under torch.no_grad(), stage2_train.py:183-190): the reference's DropPath modules
(timm, vit.py:98-109; drop_path_rate 0.1 from blip_stage2.py:37) are in training mode there and drop each sample's residual branches -
round 6 does the same (`VitEngine.forward_drop_path`); rounds 3-5 ran the inference engine."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from candidate_reranking_cir_amd import synthetic
    from candidate_reranking_cir_amd.config import BertGeometry, VitGeometry
    from candidate_reranking_cir_amd.blip_stage2 import BLIP_NLVR
    torch.manual_seed(0)
    vit = VitGeometry(image_size=64, patch_size=16, width=768, depth=6, num_heads=12)
    m = BLIP_NLVR(BertGeometry(num_hidden_layers=2), vit_geometry=vit, tokenizer=synthetic.HashTokenizer()).to("cuda")
    for p in m.visual_encoder.parameters():
        p.requires_grad_(False)                               # the frozen image encoder of the default training run (stage2_train.py:98)
    return m


def test_drop_path_engine_against_its_definition(model):
    from candidate_reranking_cir_amd import synthetic
    m = model.eval()
    eng = m.engines(text=False)[0]
    geo = m.vit_geometry
    imgs = synthetic.scene_images(range(6), 64).cuda()
    ones = torch.ones((geo.depth, 2, 6), device="cuda")
    plain, _ = eng.forward(imgs, want32=True)
    kept = eng.forward_drop_path(imgs, ones)
    assert (kept - plain).abs().max().item() < 2e-2            # same encoder (fp32 instead of fp16 stream storage)
    # sample 2 dropped everywhere: its tokens are LayerNorm(patch embedding + cls / pos) - no block touched them; sample 4 keeps only block 0's
    # attention branch at scale 1 / keep: a different result from both; every other sample unchanged
    sc = ones.clone()
    sc[:, :, 2] = 0.0
    sc[:, :, 4] = 0.0
    sc[0, 0, 4] = 1.25
    out = eng.forward_drop_path(imgs, sc)
    others = [0, 1, 3, 5]
    assert torch.equal(out[others], kept[others])
    sd = m.state_dict()
    w = sd["visual_encoder.patch_embed.proj.weight"].reshape(768, -1).half().float()
    patches = imgs.half().float().unfold(2, 16, 16).unfold(3, 16, 16).permute(0, 2, 3, 1, 4, 5).reshape(6, 16, -1)
    x0 = torch.cat([sd["visual_encoder.cls_token"].expand(6, -1, -1), patches @ w.T + sd["visual_encoder.patch_embed.proj.bias"]], 1) + sd["visual_encoder.pos_embed"]
    ref2 = F.layer_norm(x0[2], (768,), sd["visual_encoder.norm.weight"], sd["visual_encoder.norm.bias"], 1e-6)
    assert (out[2] - ref2).abs().max().item() < 2e-3
    assert (out[4] - kept[4]).abs().max().item() > 1e-2 and (out[4] - F.layer_norm(x0[4], (768,), sd["visual_encoder.norm.weight"], sd["visual_encoder.norm.bias"], 1e-6)).abs().max().item() > 1e-2


def test_img_embed_draws_drop_path_in_train_mode_only(model):
    from candidate_reranking_cir_amd import synthetic
    m = model
    imgs = synthetic.scene_images(range(48), 64).cuda()
    m.eval()
    with torch.no_grad():
        e1, e2 = m.img_embed(imgs), m.img_embed(imgs)
    assert torch.equal(e1, e2)                                  # eval: deterministic inference engine
    m.train()
    torch.manual_seed(123)
    with torch.no_grad():
        t1 = m.img_embed(imgs)
        t2 = m.img_embed(imgs)
    torch.manual_seed(123)
    with torch.no_grad():
        t3 = m.img_embed(imgs)
    assert t1.dtype == torch.float32 and t1.shape == e1.shape and not t1.requires_grad
    assert torch.equal(t1, t3) and not torch.equal(t1, t2)      # torch's generator drives the draws, like timm's
    # a sample that keeps every branch still differs from eval: kept branches are scaled by 1 / keep (timm: x.div(keep_prob) * mask).  Against
    # the all-kept result: depth 6, rates linspace(0, 0.1, 6) -> P(a sample keeps every branch) = prod keep_i^2 = 0.54
    geo = m.vit_geometry
    keep = 1.0 - torch.linspace(0.0, geo.drop_path_rate, geo.depth)
    all_kept = m.engines(text=False)[0].forward_drop_path(imgs, (1.0 / keep)[:, None, None].expand(geo.depth, 2, imgs.shape[0]).contiguous().cuda())
    same = ((t1 - all_kept).abs().amax(dim=(1, 2)) == 0).float().mean().item()
    assert 0.3 < same < 0.8, same
    assert (all_kept - e1).abs().amax(dim=(1, 2)).min().item() > 1e-3          # (the scaled branches: no sample equals its eval tokens)
    m.vit_geometry.drop_path_rate = 0.0
    try:
        with torch.no_grad():
            assert torch.equal(m.img_embed(imgs), e1)           # rate 0: the inference engine
    finally:
        m.vit_geometry.drop_path_rate = 0.1
    m.eval()
