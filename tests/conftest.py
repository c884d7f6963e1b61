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
    # device discovery through the product library itself (no torch import: it costs minutes on a fresh box)
    try:
        import ctypes as C

        from oramacore_amd import _native as N

        lib = N.load()
        h = C.c_void_p()
        if lib.orama_ctx_create(0, C.byref(h)) != 0:
            return False
        lib.orama_ctx_destroy(h)
        return True
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
