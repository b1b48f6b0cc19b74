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


@pytest.fixture(scope="session")
def weights():
    from pdb2reaction_amd import weights as W

    return W.make_synthetic_weights(0)


@pytest.fixture(scope="session")
def oracle(weights):
    import torch
    from oracle.escn_md_oracle import Oracle

    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    return Oracle(weights)


@pytest.fixture(scope="session")
def engine(weights):
    """One HIP engine for the whole GPU session (fails loudly if the library or GPU is missing)."""
    from pdb2reaction_amd.engine import Engine

    eng = Engine(0)
    eng.load_weights(weights)
    yield eng
    eng.close()


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: d[k] for k in d.files}
