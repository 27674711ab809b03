import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# built models are cached (sspslam_amd/modelcache.py): the suite gets a cache of its own, shared by the processes it spawns
if "SSN_CACHE_DIR" not in os.environ:
    import tempfile
    os.environ["SSN_CACHE_DIR"] = tempfile.mkdtemp(prefix="ssn_test_cache_")
    import atexit
    import shutil
    atexit.register(shutil.rmtree, os.environ["SSN_CACHE_DIR"], ignore_errors=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")
    config.addinivalue_line("markers", "needs_nengo: cross-check against the real nengo package (skipped unless it is installed)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load
