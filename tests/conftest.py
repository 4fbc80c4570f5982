import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_arrays():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "golden_ids.npz"))


@pytest.fixture(scope="session")
def corpora():
    from tests import fixtures
    return fixtures.Corpora()


@pytest.fixture(scope="session")
def oracle():
    from tests import oraclelib
    return oraclelib.OracleLib()
