import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    # The CPU oracle is plain torch; on a GPU box with hundreds of hardware threads torch's default thread count makes its small
    # ops crawl (bench.py measured 170x between 32 and 256 threads on the EPYC 9575F host), and the oracle is most of the GPU
    # suite's wall time.  EIGHT threads, always: the float32 oracle's own distance from float64 — the yardstick of
    # helpers.close64 — moves by 2x with the thread count (CPU GEMM blocking = summation order), every fixture under tests/golden
    # was recorded at 8 threads (its `torch_num_threads` field), and a yardstick must not depend on the box.
    import torch
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 8)))
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a GPU: skip the gpu-marked tests instead of failing them."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a GPU (MI355X); run with -m gpu on the GPU box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
