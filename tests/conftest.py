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


_LDA_SWITCHES = ("LDA_INFLATE_PAR", "LDA_INFLATE_LPW", "LDA_INFLATE_WAVES_PER_CU",
                 "LDA_NO_SMALL", "LDA_NO_SEGMENTS", "LDA_HOST_THREADS",
                 "LDA_NO_STREAM_PAR", "LDA_STREAM_PAR_MIN", "LDA_STREAM_WINDOW",
                 "LDA_STREAM_CHUNK", "LDA_DEVICES", "LDA_FANOUT_OVERSUB")


@pytest.fixture(autouse=True)
def _restore_library_switches():
    """The library reads its tuning switches once; tests that change them call
    binding.reload_env().  After every test the switches are put back and the
    library re-reads them, so the tests that follow run the default kernels
    again (monkeypatch may already have restored the environment by the time
    this teardown runs, which is why the re-read is unconditional)."""
    before = {k: os.environ.get(k) for k in _LDA_SWITCHES}
    yield
    for k, v in before.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    from libdeflate_amd import binding
    if binding._lib is not None:
        binding.reload_env()
