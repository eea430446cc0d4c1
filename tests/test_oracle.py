"""CPU-only checks of the oracle itself: cv2-written artefact, hand-derived known
answers, and C restatement vs the independent numpy/scipy restatement.

Origin labels: ``cv2-artefact`` = produced by real cv2 and shipped in the
reference tree; ``analytic`` = derived by hand from the OpenCV definitions;
``restatement`` = agreement of two independently written restatements.
"""
import numpy as np
import pytest
from scipy import ndimage

from oracle import frontref as fr
from oracle import np_ref as nr


# ---------------------------------------------------------------- cv2-artefact
def test_gray_kat_window(gray_kat):
    assert np.array_equal(fr.bgr2gray(gray_kat["bgr"]), gray_kat["gray"])
    assert np.array_equal(nr.bgr2gray(gray_kat["bgr"]), gray_kat["gray"])


def test_gray_kat_all_colours(gray_kat):
    cols = gray_kat["colours_bgr"][None, :, :]
    assert np.array_equal(fr.bgr2gray(cols)[0], gray_kat["colours_gray"])


def test_bitwise_not_kat(gray_kat):
    assert np.array_equal(fr.bitwise_not(gray_kat["bgr"]), gray_kat["inverted_bgr"])


def test_dilate_2x2_anchor_and_border_pinned_by_cv2_artefact(arrow_pin):
    # cv2-artefact: newVersion3/detect_arrowv2.py:10-11,33,126-146 (see tests/golden/make_golden.py)
    from tests.conftest import arrow_invariant_mismatches

    thresh, arrow = arrow_pin
    for d in (fr.morphology_ex(thresh, 1, (2, 2), 1), nr.morph(thresh, 1, (2, 2), 1)):
        assert set(np.unique(d)) <= {0, 255}
        assert arrow_invariant_mismatches(d, arrow) == 0
    # control: the other possible window of an even kernel (offsets {0, +1}) is NOT what cv2 did
    other = fr.morphology_ex(thresh, 1, (2, 2), 1, anchor=(0, 0))
    assert arrow_invariant_mismatches(other, arrow) > 1000
    # the border rule: out-of-image samples never win (a dilation never lights a pixel from outside)
    assert fr.morphology_ex(np.zeros((5, 7), np.uint8), 1, (2, 2), 1).sum() == 0


def test_restatements_agree_on_the_reference_diagrams():
    # restatement: the C and the numpy statements of the parsers' prefix on a sample of the reference's own inputs
    from tests.conftest import load_reference_diagram, reference_diagrams

    for name, ts in reference_diagrams()[::6]:
        bgr = load_reference_diagram(name)
        g = fr.bgr2gray(bgr)
        assert np.array_equal(g, nr.bgr2gray(bgr)), name
        for t in ts:
            th = fr.threshold(g, t, 255, 1)[1]
            assert np.array_equal(th, nr.threshold(g, t, 255, 1)[1]), name
            assert np.array_equal(fr.morphology_ex(th, 2, (1, 15), 2), nr.morph(th, 2, (1, 15), 2)), name
            assert np.array_equal(fr.morphology_ex(th, 2, (15, 1), 2), nr.morph(th, 2, (15, 1), 2)), name


# ---------------------------------------------------------------- analytic
def test_gray_primaries():
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0]]], np.uint8)
    assert fr.bgr2gray(px).tolist() == [[29, 150, 76, 255, 0]]


@pytest.mark.parametrize("t", [100, 127, 128, 150, 200, 240])
def test_threshold_truth_table(t):
    g = np.array([[0, t - 1, t, t + 1, 255]], np.uint8)
    r, d = fr.threshold(g, t, 255, 1)
    assert r == float(t) and d.tolist() == [[255, 255, 255, 0, 0]]
    r, d = fr.threshold(g, t, 255, 0)
    assert d.tolist() == [[0, 0, 0, 255, 255]]
    r, d = fr.threshold(g, t + 0.7, 200.4, 0)
    assert r == float(t) and d.tolist() == [[0, 0, 0, 200, 200]]


def test_open_keeps_runs_of_29():
    # open((15,1), it=2) == erode 29 then dilate 29: keeps horizontal runs >= 29
    img = np.zeros((5, 200), np.uint8)
    img[1, 10:38] = 255  # 28 long -> removed
    img[3, 10:39] = 255  # 29 long -> kept
    out = fr.morphology_ex(img, 2, (15, 1), iterations=2)
    assert out[1].sum() == 0
    assert np.array_equal(out[3], img[3])
    # a run touching the border counts outside pixels as foreground for the erosion
    img2 = np.zeros((1, 100), np.uint8)
    img2[0, :15] = 255
    out2 = fr.morphology_ex(img2, 2, (15, 1), iterations=2)
    assert np.array_equal(out2, img2)
    img3 = np.zeros((1, 100), np.uint8)
    img3[0, :14] = 255
    assert fr.morphology_ex(img3, 2, (15, 1), iterations=2).sum() == 0


def test_even_kernel_anchor():
    # 2x2 all-ones, anchor (1,1): window offsets {-1, 0}
    img = np.zeros((5, 5), np.uint8)
    img[2, 2] = 255
    d = fr.morphology_ex(img, 1, (2, 2))
    exp = np.zeros((5, 5), np.uint8)
    exp[2:4, 2:4] = 255
    assert np.array_equal(d, exp)
    e = fr.morphology_ex(255 - img, 0, (2, 2))
    assert np.array_equal(e, 255 - exp)


def test_gaussian_5x5_impulse():
    img = np.zeros((9, 9), np.uint8)
    img[4, 4] = 255
    out = fr.gaussian_blur(img, (5, 5), 0)
    k = np.array([1, 4, 6, 4, 1])
    exp = (np.outer(k, k) * 255 + 128) >> 8
    assert np.array_equal(out[2:7, 2:7], exp)
    assert np.array_equal(fr.gaussian_kernel_q8(5, 0), [16, 64, 96, 64, 16])
    assert fr.gaussian_kernel_q8(5, 1.0).sum() == 256


def test_median_blur_known_answers_and_c_vs_numpy():
    # analytic: an isolated pixel vanishes, a 3-wide bar survives a 3x3 median except at its ends, the replicated
    # border keeps a constant edge row; restatement: C vs numpy (scipy's median_filter with mode="nearest" as a third)
    img = np.zeros((7, 9), np.uint8)
    img[3, 4] = 200
    assert fr.median_blur(img, 3).sum() == 0 and fr.median_blur(img, 5).sum() == 0
    bar = np.zeros((7, 9), np.uint8)
    bar[2:5, :] = 90
    assert np.array_equal(fr.median_blur(bar, 3), bar)
    edge = np.zeros((6, 6), np.uint8)
    edge[0, :] = 255  # rows -1 (copy of row 0), 0 give 6 of 9 values
    assert fr.median_blur(edge, 3)[0].tolist() == [255] * 6 and fr.median_blur(edge, 3)[1].sum() == 0
    rng = np.random.default_rng(5)
    for h, w in [(1, 1), (1, 17), (23, 1), (2, 2), (40, 53), (64, 128)]:
        g = rng.integers(0, 256, (h, w), dtype=np.uint8)
        for k in (3, 5):
            a = fr.median_blur(g, k)
            assert np.array_equal(a, nr.median_blur(g, k))
            assert np.array_equal(a, ndimage.median_filter(g, size=k, mode="nearest"))
    with pytest.raises(ValueError):
        fr.median_blur(np.zeros((4, 4), np.uint8), 4)


def test_canny_vertical_step():
    g = np.zeros((8, 12), np.uint8)
    g[:, 6:] = 255
    e = fr.canny(g, 50, 150)
    # |dx| = 1020 on columns 5 and 6; NMS keeps m > left && m >= right -> column 5 only
    exp = np.zeros_like(g)
    exp[:, 5] = 255
    assert np.array_equal(e, exp)


def test_canny_weak_chain_needs_a_seed():
    # horizontal ramp giving magnitude between low and high: weak only -> nothing survives
    g = np.zeros((8, 12), np.uint8)
    g[:, 6:] = 20  # |dx| = 80: 50 < 80 <= 150
    assert fr.canny(g, 50, 150).sum() == 0
    g2 = g.copy()
    g2[:4, 6:] = 255  # top half strong, bottom half weak but connected
    e = fr.canny(g2, 50, 150)
    # rows 5..7 have magnitude 80 only (weak) yet survive through the chain to the strong top half
    assert (e[5:, 5] == 255).all() and (e[:3, 5] == 255).all()


def test_hough_single_pixel_and_line():
    img = np.zeros((40, 60), np.uint8)
    img[7, 11] = 255
    acc = fr.hough_accum(img, 1, np.pi / 180)
    na, nrho = acc.shape[0] - 2, acc.shape[1] - 2
    assert (na, nrho) == (180, 201)
    assert acc.sum() == 180 and acc.max() == 1
    assert acc[1, 11 + (nrho - 1) // 2 + 1] == 1  # theta 0: rho = x
    assert acc[91, 7 + (nrho - 1) // 2 + 1] == 1  # theta 90 deg: rho = y
    img[:] = 0
    img[20, 5:55] = 255
    lines = fr.hough_lines(img, 1, np.pi / 180, 40)
    assert lines is not None
    rho, theta = lines[0, 0]
    assert rho == 20.0 and theta == np.float32(90) * np.float32(np.pi / 180)
    assert fr.hough_lines(np.zeros((10, 10), np.uint8), 1, np.pi / 180, 2) is None


def test_otsu_bimodal():
    g = np.concatenate([np.full(500, 40, np.uint8), np.full(300, 200, np.uint8)]).reshape(20, 40)
    r, d = fr.threshold(g, 0, 255, 8)
    assert r == 40.0  # first strict maximum of the between-class variance
    assert nr.otsu(g) == 40
    assert d.sum() == 300 * 255


# ---------------------------------------------------------------- restatement
def _rand_imgs():
    rng = np.random.default_rng(7)
    yield rng.integers(0, 256, (37, 53), dtype=np.uint8)
    yield (rng.random((64, 131)) < 0.3).astype(np.uint8) * 255
    sm = ndimage.uniform_filter(rng.integers(0, 256, (90, 77)).astype(np.float32), 5)
    yield sm.astype(np.uint8)


@pytest.mark.parametrize("op", [0, 1, 2, 3])
@pytest.mark.parametrize("ks,it", [((3, 3), 1), ((3, 3), 2), ((15, 1), 2), ((1, 15), 2), ((2, 2), 1), ((4, 3), 2), ((1, 1), 3), ((5, 7), 0)])
def test_morph_c_vs_numpy(op, ks, it):
    for img in _rand_imgs():
        assert np.array_equal(fr.morphology_ex(img, op, ks, it), nr.morph(img, op, ks, it))


def test_blur_c_vs_numpy():
    for img in _rand_imgs():
        assert np.array_equal(fr.gaussian_blur(img, (5, 5), 0), nr.gaussian_blur_5x5_sigma0(img))
        assert np.array_equal(fr.gaussian_blur(img, (3, 3), 0), nr.gaussian_blur_3x3_sigma0(img))


def test_canny_c_vs_numpy(cd_bgr, login_bgr):
    imgs = list(_rand_imgs()) + [fr.bgr2gray(cd_bgr), fr.bgr2gray(login_bgr)[:300, :400]]
    for g in imgs:
        for lo, hi in [(50, 150), (50, 50), (200, 100)]:
            e, m = fr.canny(g, lo, hi, return_map=True)
            assert np.array_equal(m, nr.canny_map(g, lo, hi))
            assert np.array_equal(e, nr.canny(g, lo, hi))


def test_hough_c_vs_numpy(cd_bgr):
    e = fr.canny(fr.bgr2gray(cd_bgr), 50, 150)
    acc = fr.hough_accum(e, 1, np.pi / 180)
    assert np.array_equal(acc, nr.hough_accum(e, 1, np.pi / 180))
    assert acc.sum() == 180 * np.count_nonzero(e)
    for thr in (2, 60):
        a, b = fr.hough_lines(e, 1, np.pi / 180, thr), nr.hough_lines(e, 1, np.pi / 180, thr)
        assert a.shape == b.shape and np.array_equal(a, b)


def test_adaptive_c_vs_numpy():
    for img in _rand_imgs():
        for typ in (0, 1):
            for block in (3, 11, 31):
                assert np.array_equal(fr.adaptive_threshold(img, 255, 1, typ, block, 2),
                                      nr.adaptive_threshold_gaussian_ref(img, 255, typ, block, 2))
                assert np.array_equal(fr.adaptive_threshold(img, 255, 0, typ, block, 2),
                                      nr.adaptive_threshold_mean_ref(img, 255, typ, block, 2))


def test_otsu_c_vs_numpy(cd_bgr):
    for img in list(_rand_imgs()) + [fr.bgr2gray(cd_bgr)]:
        assert fr.threshold(img, 0, 255, 8)[0] == nr.otsu(img)
        t, d = fr.threshold(img, 0, 255, 9)
        assert np.array_equal(d, nr.threshold(img, 0, 255, 9)[1])


# ------------------------------------------------------------------ external contours -> rectangles
def _rings(h, w):
    """Nested square rings around (h/2, w/2) plus satellites inside and outside of them."""
    m = np.zeros((h, w), np.uint8)
    cy, cx = h // 2, w // 2
    for r in range(3, min(cy, cx) - 1, 4):
        m[cy - r : cy + r + 1, cx - r : cx + r + 1] = 0
        m[cy - r, cx - r : cx + r + 1] = 255
        m[cy + r, cx - r : cx + r + 1] = 255
        m[cy - r : cy + r + 1, cx - r] = 255
        m[cy - r : cy + r + 1, cx + r] = 255
    m[cy, cx] = 255
    m[0, 0] = 255
    m[h - 1, w - 2 : w] = 255
    return m


def test_external_boxes_known_answers():
    # get_segments (umlClassdiagram.py:33-36): two bars, listed bottom-most first like cv2 does
    m = np.zeros((20, 30), np.uint8)
    m[2, 3:25] = 255
    m[10:18, 7] = 255
    assert fr.external_boxes(m).tolist() == [[7, 10, 1, 8], [3, 2, 22, 1]]
    # a ring hides what is inside it; a diagonal chain is one component
    m = np.zeros((12, 12), np.uint8)
    m[1:10, 1:10] = 255
    m[2:9, 2:9] = 0
    m[5, 5] = 255
    for i in range(3):
        m[9 + i, 9 + i] = 255
    assert fr.external_boxes(m).tolist() == [[1, 1, 11, 11]]
    # a ring open through a diagonal gap still encloses (background is 4-connected)
    m = np.zeros((9, 9), np.uint8)
    m[1, 1:7] = m[7, 1:8] = 255
    m[1:8, 1] = 255
    m[2:8, 7] = 255  # (1,7) missing: corner gap closed diagonally by (1,6)-(2,7)
    m[4, 4] = 255
    assert fr.external_boxes(m).tolist() == [[1, 1, 7, 7]]
    m[2, 7] = 0  # now a real gap
    assert fr.external_boxes(m).tolist() == [[4, 4, 1, 1], [1, 1, 7, 7]]
    assert fr.external_boxes(np.zeros((5, 7), np.uint8)).shape == (0, 4)
    assert fr.external_boxes(np.full((5, 7), 9, np.uint8)).tolist() == [[0, 0, 7, 5]]
    assert fr.external_boxes(np.ones((1, 1), np.uint8)).tolist() == [[0, 0, 1, 1]]


def test_external_boxes_c_vs_numpy():
    rng = np.random.default_rng(11)
    masks = [_rings(41, 57), _rings(64, 64), _rings(9, 70)]
    for p in (0.05, 0.3, 0.5, 0.62, 0.8):
        for h, w in ((1, 40), (33, 1), (17, 31), (48, 65), (70, 129)):
            masks.append(((rng.random((h, w)) < p) * rng.integers(1, 256, (h, w))).astype(np.uint8))
    for m in masks:
        a, b = fr.external_boxes(m), nr.external_boxes(m)
        assert a.shape == b.shape and np.array_equal(a, b)


def test_external_boxes_on_line_masks(cd_bgr):
    t = fr.threshold(fr.bgr2gray(cd_bgr), 150, 255, 1)[1]
    for ks in ((1, 15), (15, 1)):
        lines = fr.morphology_ex(t, 2, ks, 2)
        a = fr.external_boxes(lines)
        assert len(a) >= 2 and np.array_equal(a, nr.external_boxes(lines))
    assert np.array_equal(fr.external_boxes(t), nr.external_boxes(t))


# ------------------------------------------------------------------ HoughLinesP (restatement; parity unpinned)
def _segs(l):
    return [] if l is None else sorted(tuple(int(v) for v in s) for s in l.reshape(-1, 4))


def _norm(seg):
    x0, y0, x1, y1 = seg
    return (x0, y0, x1, y1) if (x0, y0) <= (x1, y1) else (x1, y1, x0, y0)


def test_hough_lines_p_known_answers():
    # lineDetection.py:12: HoughLinesP(edges, 1, pi/180, threshold, minLineLength=, maxLineGap=)
    m = np.zeros((40, 60), np.uint8)
    m[10, 5:50] = 255
    m[5:35, 20] = 255
    got = {_norm(s) for s in _segs(fr.hough_lines_p(m, 1, np.pi / 180, 15, 10, 2))}
    assert got == {(5, 10, 49, 10), (20, 5, 20, 34)}
    assert fr.hough_lines_p(np.zeros((20, 20), np.uint8), 1, np.pi / 180, 5, 3, 1) is None
    # a gap of 3 pixels: bridged with maxLineGap >= 3, two segments below
    m = np.zeros((30, 80), np.uint8)
    m[12, 4:30] = 255
    m[12, 33:70] = 255
    one = {_norm(s) for s in _segs(fr.hough_lines_p(m, 1, np.pi / 180, 10, 5, 3))}
    two = {_norm(s) for s in _segs(fr.hough_lines_p(m, 1, np.pi / 180, 10, 5, 2))}
    assert one == {(4, 12, 69, 12)}
    assert two == {(4, 12, 29, 12), (33, 12, 69, 12)}
    # minLineLength drops the short bar, and its pixels are used up all the same
    assert {_norm(s) for s in _segs(fr.hough_lines_p(m, 1, np.pi / 180, 10, 30, 2))} == {(33, 12, 69, 12)}
    # a 45-degree diagonal
    m = np.zeros((64, 64), np.uint8)
    for k in range(8, 56):
        m[k, k] = 255
    assert {_norm(s) for s in _segs(fr.hough_lines_p(m, 1, np.pi / 180, 20, 10, 0))} == {(8, 8, 55, 55)}


def test_hough_lines_p_uses_every_pixel_once(cd_bgr):
    # every segment lies inside the image, is at least minLineLength long on one axis, and a second
    # run gives the same list (the pseudo-random visiting order is fixed)
    e = fr.canny(fr.bgr2gray(cd_bgr), 50, 150)
    a = fr.hough_lines_p(e, 1, np.pi / 180, 50, 50, 10)
    b = fr.hough_lines_p(e, 1, np.pi / 180, 50, 50, 10)
    assert a is not None and np.array_equal(a, b)
    h, w = e.shape
    s = a.reshape(-1, 4)
    assert (s[:, [0, 2]] >= 0).all() and (s[:, [0, 2]] < w).all() and (s[:, [1, 3]] >= 0).all() and (s[:, [1, 3]] < h).all()
    assert (np.maximum(abs(s[:, 2] - s[:, 0]), abs(s[:, 3] - s[:, 1])) >= 50).all()
