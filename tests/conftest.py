import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def g1():
    return load_golden("sed_g1.npz")


@pytest.fixture(scope="session")
def g2():
    return load_golden("sed_g2.npz")


@pytest.fixture(scope="session")
def g3():
    return load_golden("sed_g3.npz")


@pytest.fixture(scope="session")
def g4():
    return load_golden("sed_g4.npz")


G3_CASES = ["m1n_block", "m1n_arp220", "m1n_loga_noblock", "m2n_block", "m2n_fixedbeta",
            "m2n_loga_noblock", "m1_marg", "thick", "thick_loga"]
