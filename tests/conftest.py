import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the synthetic fixtures have no vocab.txt: the hashing stand-in tokenizer must be asked for (memvul_amd/tokenizer.py)
os.environ.setdefault("MEMVUL_ALLOW_HASH_TOKENIZER", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
