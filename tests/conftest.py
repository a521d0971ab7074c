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
    # (pytest-timeout registers this marker itself when installed -- it is in this image; declared here so that a box without the
    # plugin collects the threaded-pipeline tests instead of failing on an unknown marker)
    config.addinivalue_line("markers", "timeout(seconds): per-test limit of the threaded input-pipeline tests (pytest-timeout)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
