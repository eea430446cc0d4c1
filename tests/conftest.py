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
def gray_kat():
    return dict(np.load(os.path.join(GOLDEN, "gray_kat.npz")))


@pytest.fixture(scope="session")
def cd_bgr():
    return np.load(os.path.join(GOLDEN, "cd.npz"))["bgr"]


@pytest.fixture(scope="session")
def login_bgr():
    return np.load(os.path.join(GOLDEN, "login.npz"))["bgr"]


def uml_like(seed, h, w, noise=0):
    """Small UML-like gray/BGR raster for parity tests (boxes, connectors, glyph clutter)."""
    from imageprocessing_amd.synth import synth_diagram

    return synth_diagram(seed, h, w, noise=noise)
