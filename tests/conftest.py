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


@pytest.fixture(scope="session")
def arrow_pin():
    """thresh_image.png / arrow_image.png of the reference (written by cv2, newVersion3/detect_arrowv2.py:126,146)
    as {0,255} planes."""
    d = np.load(os.path.join(GOLDEN, "arrow_pin.npz"))
    h, w = (int(v) for v in d["shape"])
    thresh = np.unpackbits(d["thresh_bits"], axis=1)[:, :w].astype(np.uint8) * 255
    arrow = np.unpackbits(d["arrow_bits"], axis=1)[:, :w].astype(np.uint8) * 255
    assert thresh.shape == (h, w)
    return thresh, arrow


def arrow_invariant_mismatches(dilated, arrow):
    """Pixels where fill_holes(8-connected components of `dilated` that meet `arrow`) differs from `arrow`.

    arrow_image.png holds filled contours (cv2.drawContours(..., -1, 255, -1), detect_arrowv2.py:33) of whole
    components of cv2.dilate(thresh, rect(2,2)) (detect_arrowv2.py:10-13), so the count is 0 exactly when
    `dilated` is what cv2.dilate produced."""
    from scipy import ndimage

    lab, _ = ndimage.label(dilated > 0, structure=np.ones((3, 3), np.uint8))
    ids = np.unique(lab[arrow > 0])
    sel = np.isin(lab, ids[ids > 0])
    return int((ndimage.binary_fill_holes(sel) != (arrow > 0)).sum())


# the parsers' thresholds (umlClassdiagram.py:18, umlSequencediagram.py:21, umlActivitydiagram.py:33,
# umlStateDiagram.py:18) by the kind of diagram a file of the reference's images/ directory shows
PARSER_THRESHOLDS = (("cd", 150), ("class", 150), ("no_", 150), ("sq", 128), ("login", 128), ("act", 128),
                     ("order", 128), ("atm", 128), ("std", 100))


def reference_diagrams():
    """[(name, thresholds)] for the 41 input diagrams of the reference (tests/golden/images)."""
    out = []
    for f in sorted(os.listdir(os.path.join(GOLDEN, "images"))):
        t = [v for p, v in PARSER_THRESHOLDS if f.startswith(p)]
        out.append((f, tuple(t[:1]) if t else (150, 128, 100)))
    return out


def load_reference_diagram(name):
    from imageprocessing_amd.batching import decode_bgr

    with open(os.path.join(GOLDEN, "images", name), "rb") as fh:
        return decode_bgr(fh.read())
