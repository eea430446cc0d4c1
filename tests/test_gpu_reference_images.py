"""GPU parity on the reference's own files: the cv2-written thresh_image.png / arrow_image.png pair, which pins
cv2.dilate with an even-sized kernel (origin ``cv2-artefact``), and the 41 input diagrams of the reference's
images/ directory through the calls its scripts make on them (origin ``restatement``: expected values come from
oracle/, no cv2 capture of them exists)."""
import numpy as np
import pytest
import torch

from oracle import frontref as fr
from tests.conftest import arrow_invariant_mismatches, load_reference_diagram, reference_diagrams

pytestmark = pytest.mark.gpu

cv2 = None


@pytest.fixture(scope="module", autouse=True)
def _mod():
    global cv2
    import imageprocessing_amd as m

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    cv2 = m
    yield


def test_dilate_2x2_matches_the_cv2_written_arrow_image(arrow_pin):
    # newVersion3/detect_arrowv2.py:10-11 (the call), :33 (filled contours), :126,146 (the two files)
    thresh, arrow = arrow_pin
    kernel = cv2.getStructuringElement(cv2.MORPH_RECT, (2, 2))
    d = cv2.dilate(thresh, kernel, iterations=1)
    assert arrow_invariant_mismatches(d, arrow) == 0
    assert np.array_equal(d, fr.morphology_ex(thresh, 1, (2, 2), 1))
    # the same call on the device tensor and through morphologyEx(MORPH_DILATE)
    dt = cv2.dilate(torch.from_numpy(thresh).cuda(), kernel)
    assert np.array_equal(dt.cpu().numpy(), d)
    assert np.array_equal(cv2.morphologyEx(thresh, cv2.MORPH_DILATE, kernel), d)
    # an explicit anchor (0, 0) is the other window and must NOT reproduce the artefact
    assert arrow_invariant_mismatches(cv2.dilate(thresh, kernel, anchor=(0, 0)), arrow) > 1000


@pytest.mark.parametrize("name,thresholds", reference_diagrams())
def test_parser_prefix_and_line_detection_on_reference_diagram(name, thresholds):
    # umlClassdiagram.py:12-30, umlSequencediagram.py:15-33, umlActivitydiagram.py:22-44, umlStateDiagram.py:12-30
    # (threshold of the parser the diagram belongs to) and lineDetection.py:5-9 in one fused pass
    bgr = load_reference_diagram(name)
    g = fr.bgr2gray(bgr)
    e = fr.canny(g, 50, 150)
    acc = fr.hough_accum(e, 1, np.pi / 180)
    for t in thresholds:
        res = cv2.front_end(bgr, thresh_val=t)
        th = fr.threshold(g, t, 255, 1)[1]
        assert np.array_equal(res.gray, g)
        assert np.array_equal(res.thresh, th)
        assert np.array_equal(res.vlines, fr.morphology_ex(th, 2, (1, 15), 2))
        assert np.array_equal(res.hlines, fr.morphology_ex(th, 2, (15, 1), 2))
        assert np.array_equal(res.edges, e)
        assert np.array_equal(res.accum.cpu().numpy(), acc)


@pytest.mark.parametrize("name", [n for n, _ in reference_diagrams()])
def test_comparison_script_calls_on_reference_diagram(name):
    # "New Version/comparison.py":7-30: gray, threshold(127, BINARY_INV), Canny(50,150,3), HoughLines(.., 2) on both
    bgr = load_reference_diagram(name)
    gray = cv2.cvtColor(bgr, cv2.COLOR_BGR2GRAY)
    _, binary = cv2.threshold(gray, 127, 255, cv2.THRESH_BINARY_INV)
    edges = cv2.Canny(gray, 50, 150, apertureSize=3)
    g0 = fr.bgr2gray(bgr)
    assert np.array_equal(gray, g0)
    assert np.array_equal(binary, fr.threshold(g0, 127, 255, 1)[1])
    assert np.array_equal(edges, fr.canny(g0, 50, 150))
    for src in (binary, edges):
        a = cv2.HoughLines(src, 1, np.pi / 180, 2)
        b = fr.hough_lines(src, 1, np.pi / 180, 2)
        assert (a is None) == (b is None)
        if a is not None:
            assert a.shape == b.shape and np.array_equal(a, b)


def test_reference_diagrams_as_one_upload(tmp_path):
    # receiveFile.py:22: the files of one upload, here all 41, through the batched loop
    from imageprocessing_amd.batching import process_images

    names = [n for n, _ in reference_diagrams()]
    imgs = [load_reference_diagram(n) for n in names]
    out = process_images(imgs, thresh_val=128)
    assert len(out) == len(imgs)
    for n, bgr, r in zip(names, imgs, out):
        g = fr.bgr2gray(bgr)
        th = fr.threshold(g, 128, 255, 1)[1]
        assert np.array_equal(r.gray, g) and np.array_equal(r.thresh, th), n
        assert np.array_equal(r.vlines, fr.morphology_ex(th, 2, (1, 15), 2)), n
        assert np.array_equal(r.hlines, fr.morphology_ex(th, 2, (15, 1), 2)), n
