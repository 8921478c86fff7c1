"""bench.py's accounting helpers (no GPU): the algorithmic-flop formula must reproduce SURVEY.md section 8(d)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_gflop_matches_survey():
    b = _bench()
    a = b.algorithmic_gflop(197, 32, 100)
    assert abs(a["vit"] - 35.13) < 0.02 and abs(a["fuse"] - 24.83) < 0.02 and abs(a["s1"] - 12.19) < 0.02
    assert abs(a["per_triplet"] - 60.43) < 0.02
    assert abs(b.algorithmic_gflop(197, 32, 50)["per_triplet"] - 60.91) < 0.02
    assert abs(b.algorithmic_gflop(197, 32, 10)["per_triplet"] - 64.69) < 0.02
    a384 = b.algorithmic_gflop(577, 32, 100)
    assert abs(a384["vit"] - 110.97) < 0.05 and abs(a384["fuse"] - 47.25) < 0.05 and abs(a384["per_triplet"] - 159.56) < 0.05


def test_usable_cpus_is_positive_and_bounded():
    b = _bench()
    n = b.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_launcher_starts_one_child_per_rank_and_relays_failure(tmp_path, monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE becomes a launcher (bench.launch_ranks): N children with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR=127.0.0.1 / one shared MASTER_PORT, the same argument vector, exit code = a failed rank's."""
    import sys
    b = _bench()
    script = tmp_path / "child.py"
    script.write_text(
        "import os, sys\n"
        "r = os.environ['RANK']\n"
        "open(os.path.join(os.path.dirname(__file__), 'rank' + r), 'w').write(' '.join([os.environ[k] for k in "
        "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')] + sys.argv[1:]))\n"
        "sys.exit(int(os.environ.get('FAIL_RANK', '-1')) == int(r) and 7 or 0)\n")
    monkeypatch.setattr(b, "__file__", str(script))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3", "--steps", "1"])
    assert b.launch_ranks(3) == 0
    got = [(tmp_path / f"rank{r}").read_text().split() for r in range(3)]
    assert [g[0] for g in got] == ["0", "1", "2"] and [g[1] for g in got] == ["0", "1", "2"]
    assert all(g[2] == "3" and g[3] == "127.0.0.1" and g[5:] == ["--gpus", "3", "--steps", "1"] for g in got)
    assert len({g[4] for g in got}) == 1 and int(got[0][4]) > 0
    monkeypatch.setenv("FAIL_RANK", "1")
    assert b.launch_ranks(3) == 7


def test_precision_flags_map_to_model_settings():
    """--dtype / --stream-dtype -> BLIP_NLVR settings (no GPU: the engines are only packed on first use) and the names the JSON line prints."""
    import torch
    from candidate_reranking_cir_amd import config
    from candidate_reranking_cir_amd.blip_stage2 import BLIP_NLVR
    b = _bench()
    m = BLIP_NLVR(med_config=config.BertGeometry(hidden_size=128, num_attention_heads=2, num_hidden_layers=2, intermediate_size=256, encoder_width=128),
                  vit_geometry=config.VitGeometry(image_size=64, width=128, depth=2, num_heads=2), tokenizer=None)
    assert (m.precision, m.stream_dtype, m.vit_stream_dtype) == ("f16", torch.float16, torch.float16) and b.DEFAULT_DTYPE == "f16"   # library defaults
    want = {("f16", "auto"): ("f16", torch.float16, None, "f16"), ("bf16", "auto"): ("bf16", torch.bfloat16, None, "f16"),
            ("mixed", "auto"): ("mixed", torch.float16, torch.bfloat16, "f16"), ("f16", "split"): ("f16", torch.float16, None, "split"),
            ("f16", "f32"): ("f16", torch.float16, None, "f32"), ("bf16", "f32"): ("bf16", torch.bfloat16, None, "f32")}
    for (dtype, stream), (prec, cdt, idt, sname) in want.items():
        b.apply_precision(m, dtype, stream)
        assert (m.precision, m.compute_dtype, m.image_dtype, b.stream_name(m)) == (prec, cdt, idt, sname)
        assert m.token_dtype == (idt or cdt)
    # the exact mode (round 5): fp32 everywhere, whatever --stream-dtype says; leaving it restores the automatic streams
    b.apply_precision(m, "exact", "f16")
    assert (m.precision, m.compute_dtype, m.image_dtype, m.stream_dtype, m.vit_stream_dtype, m.token_dtype) == \
        ("exact", torch.float32, None, torch.float32, torch.float32, torch.float32) and b.stream_name(m) == "f32"
    # text32 (round 5): the text side in the exact mode's arithmetic over the fp16 ViT / cross block
    b.apply_precision(m, "text32", "f32")
    assert (m.precision, m.compute_dtype, m.image_dtype, m.stream_dtype, m.vit_stream_dtype, m.token_dtype) == \
        ("text32", torch.float32, torch.float16, torch.float32, torch.float16, torch.float16) and b.stream_name(m) == "split"
    assert m.text_split3 == 8 and m.text_arithmetic.startswith("split8")                     # round 6: fp16 + 2 scaled-fp8 products
    b.apply_precision(m, "text32x3", "auto")                                                  # round 5's three fp16 products, same mode otherwise
    assert m.precision == "text32" and m.text_split3 == 3 and m.text_arithmetic.startswith("three") and b.stream_name(m) == "split"
    b.apply_precision(m, "f16", "auto")
    assert (m.precision, m.stream_dtype, m.vit_stream_dtype) == ("f16", torch.float16, torch.float16)
    assert set(b.PRECISION_NOTE) == {"f16", "bf16", "mixed", "text32", "text32x3", "exact"} == set(b.PEAK_TFLOPS) and b.PEAK_TFLOPS["exact"] == 157.3
