"""pytest configuration: `gpu` marker = needs a real MI355X (run with `-m gpu`); everything else runs on CPU."""

import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "tests")):  # repo root (package, oracle) and tests/ (helpers shared between test modules)
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs an AMD MI355X GPU (HIP kernels through the C ABI)")


def pytest_collection_modifyitems(config, items):
    # multi-process tests (gloo / RCCL rendezvous, spawned workers) must FAIL rather than hang a whole run if a rank never shows up
    try:
        import pytest_timeout  # noqa: F401

        for item in items:
            if item.get_closest_marker("timeout") is None:
                item.add_marker(pytest.mark.timeout(900))
    except ImportError:  # pragma: no cover
        pass
    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
