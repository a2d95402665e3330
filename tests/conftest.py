import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _fresh_library_env():
    """libmtn_hip caches its MTN_* environment switches per call site (mtn_reload_env): start every test from the current
    environment and drop whatever the test set once it is over.  Tests that flip a switch mid-test call lib.reload_env()."""
    from mtn_amd import lib
    lib.reload_env()
    yield
    lib.reload_env()
