import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "marl-uavs-targets-tracking_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    import json
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    return z, meta


@pytest.fixture(scope="session")
def pmi_state_dict():
    z = np.load(os.path.join(GOLDEN, "pmi_h128.npz"))
    return {k: z[k] for k in z.files if k != "meta"}


@pytest.fixture(scope="session")
def pmi_state_dict_h64():
    """PMINetwork at its class-default width (PMINet.py:21), recorded from the reference (gen_golden.py: gen_h64)."""
    z = np.load(os.path.join(GOLDEN, "pmi_h64.npz"))
    return {k: z[k] for k in z.files if k != "meta"}

