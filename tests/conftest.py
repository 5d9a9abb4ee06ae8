import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (MI355X); run with `-m gpu` on the GPU box")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_golden.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def seeded_sd():
    """Deterministic weights with the reference's key surface (a function of the seed only)."""
    from caspr_amd.models import CaSPR
    from caspr_amd.utils.synthetic import seeded_state_dict
    return seeded_state_dict(CaSPR().state_dict(), 0)


@pytest.fixture(scope="session")
def stress_sd():
    """The same key surface with dynamics that are HARD to integrate (synthetic.stress_state_dict): time-switching gates, saturated
    softplus tails, T = 1, a latent field that moves."""
    from caspr_amd.models import CaSPR
    from caspr_amd.utils.synthetic import stress_state_dict
    return stress_state_dict(CaSPR().state_dict(), 0)
