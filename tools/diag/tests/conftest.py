"""Tests of the diagnostic build (tools/diag/build_diag.py).  Run with
    I8IE_LIB=tools/diag/libi8ie_hip_diag.so python -m pytest tools/diag/tests -m gpu -q
They reuse the harness of tests/ (abi.py, synth.py, the oracle fixture)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X")


def pytest_collection_modifyitems(config, items):
    if "diag" not in os.path.basename(os.environ.get("I8IE_LIB", "")):
        skip = pytest.mark.skip(reason="set I8IE_LIB to tools/diag/libi8ie_hip_diag.so (python tools/diag/build_diag.py)")
        for it in items:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def orc():
    import orc as _orc

    _orc.lib()
    return _orc
