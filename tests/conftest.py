import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch  # plumbing only: device discovery

        return torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def has_gpu() -> bool:
    return _has_gpu()


@pytest.fixture(scope="session")
def ctx():
    """One orama_ctx on GPU 0 for the whole session (GPU tests only)."""
    import oramacore_amd as oa

    c = oa.Context(0)
    yield c
    c.close()
