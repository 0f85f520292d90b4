import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_frames():
    import numpy as np
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_frames.npz"))


@pytest.fixture(scope="session")
def golden_counters():
    import numpy as np
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_counters.npz"))


@pytest.fixture(scope="session")
def golden_sim():
    import json
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_sim.json")))
