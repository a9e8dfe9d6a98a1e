import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Every engine the tests create starts from an arena filled with 0xFF bytes (NaN as float): a kernel that reads
# memory the engine never initialised -- e.g. a factor column outside its stored tiles -- fails loudly instead of
# passing on freshly zeroed pages.
os.environ.setdefault("IPP_POISON_ARENA", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture
def golden():
    return load_golden
