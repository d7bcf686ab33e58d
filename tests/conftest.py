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


def pytest_sessionfinish(session, exitstatus):
    """GPU sessions leave the measured parity margins (helpers.record_margin) beside their other outputs."""
    import json
    try:
        import helpers
    except Exception:
        return
    if not helpers.MARGINS:
        return
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    worst = {}
    for m in helpers.MARGINS:
        for k, v in m.items():
            if k.endswith("_frac_of_bar") and v is not None:
                worst[m["suite"]] = max(worst.get(m["suite"], 0.0), v)
    (out / "parity_margins.json").write_text(json.dumps({"worst_fraction_of_bar_by_suite": worst, "cases": helpers.MARGINS}, indent=1))
