import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

GOLDEN = os.path.join(ROOT, "tests", "golden")
KARATE = os.path.join(GOLDEN, "karate.txt")
TESTGRAPH = os.path.join(GOLDEN, "testgraph.txt")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.lib()
    return oracle_py
