import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# The library picks the games per wave by batch size (one or two below 7 168 games); the suite's small batches keep exercising the
# four-games-per-wave layout every production batch runs in - tests/test_gpu_rows.py covers the other two.
os.environ.setdefault("RMJ_ROWS", "4")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
