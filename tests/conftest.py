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


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def vo():
    """The product package with the HIP library loaded; GPU tests only."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    import visual_odometry_amd as pkg
    from visual_odometry_amd import _lib
    _lib.load()           # raises if the HIP shared library is missing
    return pkg


def rot_angle(Ra, Rb):
    """Geodesic angle between two rotation matrices, radians."""
    # chord length instead of arccos(trace): accurate down to 1e-16 for tiny angles
    chord = np.linalg.norm(np.asarray(Ra) - np.asarray(Rb)) / (2.0 * np.sqrt(2.0))
    return float(2.0 * np.arcsin(min(1.0, chord)))
