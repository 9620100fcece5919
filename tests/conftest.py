import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_cases(name):
    """Yield dicts of arrays from a tests/golden/*.npz fixture (data only, no pickle)."""
    import numpy as np

    d = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    n = int(d["n_cases"])
    out = []
    for i in range(n):
        pre = "%d_" % i
        out.append({k[len(pre):]: d[k] for k in d.files if k.startswith(pre)})
    return out


@pytest.fixture(scope="session")
def orc():
    import orc as _orc

    _orc.lib()
    return _orc
