import os
import sys

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
WEIGHTS = os.path.join(ROOT, "hifimeth_amd", "weights")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import hm_oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def oracle_models(oracle):
    return [oracle.Model(os.path.join(WEIGHTS, n + ".hmw")) for n in ("CpG", "CHG", "CHH")]
