import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / "flow-rl_amd", ROOT / "oracle", ROOT / "tests", ROOT):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """The C-ABI library, built in-tree beforehand (python flow-rl_amd/build.py)."""
    from flowrl_hip import _lib
    return _lib.load()
