import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _usable_cpus() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


# The GPU box shows every host core but grants a cgroup quota of a few: torch's default intra-op pool (one thread per visible
# core) then thrashes inside the quota - weight synthesis and fp32 -> 16-bit packing of a full-size model took 25-45 s instead
# of a few (the same work under torchrun, which exports OMP_NUM_THREADS=1, took 7 s).  Child processes inherit the setting.
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, _usable_cpus())))


def pytest_configure(config):
    import torch
    torch.set_num_threads(min(16, _usable_cpus()))
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
