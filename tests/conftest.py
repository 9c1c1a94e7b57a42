import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _usable_cores():
    """threads this process may really use: scheduler affinity and the cgroup CPU quota, not os.cpu_count() -- the GPU box
    reports 256 CPUs but owns 16; torch's default thread count then oversubscribes the oracle's CPU convs 16x"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            n = min(n, max(1, int(float(q[0]) / float(q[1]))))
    except Exception:
        pass
    return max(1, min(n, 32))


def pytest_configure(config):
    import torch
    torch.set_num_threads(_usable_cores())
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")
    config.addinivalue_line("markers", "extended: simulator twin of a test the GPU suite always runs; skipped in the default CPU run to keep it "
                                       "to a few minutes (BCP_EXTENDED=1 runs them)")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("BCP_EXTENDED") == "1":
        return
    skip = pytest.mark.skip(reason="simulator twin of a GPU-suite test (BCP_EXTENDED=1 to run)")
    for it in items:
        if "extended" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
