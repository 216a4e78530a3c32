import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    from ladi_vton_amd import _lib
    return _lib.load()


def pytest_sessionstart(session):
    """a GPU session starts its parity record from scratch (tests/util.py record_parity): the committed profiles/r06_parity.json is one run"""
    import time
    os.environ["LADI_PYTEST_SESSION"] = time.strftime("%Y%m%dT%H%M%S")
    markexpr = getattr(session.config.option, "markexpr", "") or ""
    if "gpu" in markexpr and "not gpu" not in markexpr:
        from tests import util as U
        try:
            os.remove(U.parity_path())
        except OSError:
            pass
