import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ant-multi-modal-framework_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_xdist_auto_num_workers(config):
    """`-n auto` (pytest.ini): 4 workers for the CPU suite, none for the GPU suite (-m gpu) or when ANTMMF_TEST_WORKERS says otherwise."""
    if os.environ.get("ANTMMF_TEST_WORKERS"):
        return int(os.environ["ANTMMF_TEST_WORKERS"])
    mark = config.getoption("markexpr", "") or ""
    if "gpu" in mark and "not gpu" not in mark:
        return 0
    return min(4, os.cpu_count() or 1)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import torch

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = torch.load(os.path.join(GOLDEN, name), map_location="cpu")
        return cache[name]

    return load
