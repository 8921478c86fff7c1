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
