import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def blosum62():
    from aligner_amd.matrices import get_blosum62
    return get_blosum62()


@pytest.fixture(scope="session")
def kat():
    """The reference's own known-answer vectors (src/tests/test_alignment.rs), as committed fixture data."""
    with open(os.path.join(ROOT, "tests", "golden", "legacy_kat.json")) as f:
        k = json.load(f)
    k["matrix"] = np.array(k["blosum50_sub"], dtype=np.float64)
    return k


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.build()
    return oracle
