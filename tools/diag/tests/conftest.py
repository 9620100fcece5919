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


# tests/ modules do `from conftest import load_cases, GOLDEN`: when both directories are collected in one run this file may be
# the module that name resolves to, so re-export the harness's helpers
import importlib.util as _ilu

_spec = _ilu.spec_from_file_location("_i8ie_tests_conftest", os.path.join(ROOT, "tests", "conftest.py"))
_tc = _ilu.module_from_spec(_spec)
_spec.loader.exec_module(_tc)
load_cases, GOLDEN = _tc.load_cases, _tc.GOLDEN


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X")


HERE = os.path.dirname(os.path.abspath(__file__))


def pytest_collection_modifyitems(config, items):
    # only the tests of THIS directory need the diagnostic library: a run collected from the repo root must not lose tests/
    if "diag" not in os.path.basename(os.environ.get("I8IE_LIB", "")):
        skip = pytest.mark.skip(reason="set I8IE_LIB to tools/diag/libi8ie_hip_diag.so (python tools/diag/build_diag.py)")
        for it in items:
            if os.path.abspath(str(it.fspath)).startswith(HERE + os.sep):
                it.add_marker(skip)


@pytest.fixture(scope="session")
def orc():
    import orc as _orc

    _orc.lib()
    return _orc
