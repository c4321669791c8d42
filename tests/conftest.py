import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long sweeps (forced tile variants, one child process each): still part of `-m gpu`, but run LAST")


def _have_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    items.sort(key=lambda it: 1 if "slow" in it.keywords else 0)      # stable: everything else keeps its order, the sweeps go last
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def engine():
    from fashionern_aaai2024_amd.engine import FernEngine
    eng = FernEngine("cuda:0")
    yield eng
    eng.close()
