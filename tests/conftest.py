import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def native_lib():
    """libserhip.so, built on demand here (CPU box); on the GPU box the prebuilt file is used."""
    from ser_amd import _native

    if not _native.LIB_PATH.exists():
        import __graft_entry__

        __graft_entry__.build()
    return _native.load_library()
