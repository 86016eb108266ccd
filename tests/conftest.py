import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _bounded_torch_threads():
    """The oracle is thousands of tiny CPU tensor operations per search step: on a many-core GPU host the default
    intra-op thread count (every core) makes each of them a fork-join over 100+ threads and the checker legs of the GPU
    tests several times slower (round 5: a 6-minute test file ran into a 25-minute limit).  32 threads at most - what
    bench.py's cpu_baseline legs use."""
    import torch

    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))))
    yield
