import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with `-m gpu` via gpurun)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (oracle/liboracle.so).  Test infrastructure only."""
    from tests import oracle_util
    return oracle_util.load_oracle()


@pytest.fixture(scope="session")
def ref():
    """The real reference compiled into oracle/_ref (if present)."""
    from tests import oracle_util
    r = oracle_util.load_ref()
    if r is None:
        pytest.skip("oracle/_ref/libdeflate_ref.so not built")
    return r
