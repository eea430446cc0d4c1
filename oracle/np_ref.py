"""Independent numpy/scipy restatement of the raster front-end (TEST INFRASTRUCTURE ONLY).

A second, differently-written statement of the same OpenCV 4.x semantics as
``oracle/frontref.c``, used by the tests to cross-check the C restatement
(and through it the HIP kernels).  Nothing in the product package imports it.

Parity status: the reference ships no tests and ``cv2`` is not installed here,
so apart from ``bgr2gray`` / ``bitwise_not`` (pinned by the cv2-written pair
``NewVersion2/inverted.png`` / ``NewVersion2/gray_img.png`` of the reference)
every function below is "parity unpinned".

Each function names the cv2 call it restates; reference call sites are listed
next to the matching function of ``oracle/frontref.c``.
"""
from __future__ import annotations

import math

import numpy as np
from scipy import ndimage


def bgr2gray(bgr: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(bgr, COLOR_BGR2GRAY) (umlClassdiagram.py:14)."""
    b = bgr[..., 0].astype(np.int64)
    g = bgr[..., 1].astype(np.int64)
    r = bgr[..., 2].astype(np.int64)
    return ((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15).astype(np.uint8)


def threshold(gray: np.ndarray, thresh: float, maxval: float, typ: int):
    """cv2.threshold for THRESH_BINARY(0) / THRESH_BINARY_INV(1) [| THRESH_OTSU(8)]."""
    if typ & 8:
        t = otsu(gray)
    else:
        t = math.floor(thresh)
    mv = int(np.clip(round_half_even(maxval), 0, 255))
    above = gray.astype(np.int32) > t
    if (typ & 7) == 0:
        out = np.where(above, mv, 0)
    else:
        out = np.where(above, 0, mv)
    return float(t), out.astype(np.uint8)


def round_half_even(x: float) -> int:
    return int(np.rint(x))


def otsu(gray: np.ndarray) -> int:
    """Otsu's threshold, OpenCV's sequential double-precision scan."""
    hist = np.bincount(gray.ravel(), minlength=256).astype(np.float64)
    n = gray.size
    scale = 1.0 / n
    mu = float((np.arange(256) * hist).sum()) * scale
    eps = float(np.finfo(np.float32).eps)
    mu1 = q1 = 0.0
    best, best_sigma = 0, 0.0
    for i in range(256):
        p = hist[i] * scale
        mu1 *= q1
        q1 += p
        q2 = 1.0 - q1
        if min(q1, q2) < eps or max(q1, q2) > 1.0 - eps:
            continue
        mu1 = (mu1 + i * p) / q1
        mu2 = (mu - q1 * mu1) / q2
        sigma = q1 * q2 * (mu1 - mu2) ** 2
        if sigma > best_sigma:
            best_sigma, best = sigma, i
    return best


def _rect_minmax(img: np.ndarray, kw: int, kh: int, ax: int, ay: int, dilate: bool) -> np.ndarray:
    """min/max over the window [x-ax, x-ax+kw-1] x [y-ay, y-ay+kh-1], outside ignored."""
    pad = 0 if dilate else 255
    h, w = img.shape
    left, right = ax, kw - 1 - ax
    top, bottom = ay, kh - 1 - ay
    p = np.pad(img, ((top, bottom), (left, right)), constant_values=pad)
    # scipy's origin convention: window centred at size//2 + origin; use explicit slicing instead
    view = np.lib.stride_tricks.sliding_window_view(p, (kh, kw))
    return (view.max(axis=(2, 3)) if dilate else view.min(axis=(2, 3))).astype(np.uint8)[:h, :w]


def morph(img: np.ndarray, op: int, ksize, iterations: int = 1, anchor=(-1, -1)) -> np.ndarray:
    """cv2.erode(0) / dilate(1) / morphologyEx OPEN(2) / CLOSE(3), all-ones kernel.

    Deliberately ITERATES instead of folding the iterations into one larger
    kernel, so that the folding used by the C restatement and the HIP kernels
    is itself under test.
    """
    kw, kh = ksize
    ax = kw // 2 if anchor[0] < 0 else anchor[0]
    ay = kh // 2 if anchor[1] < 0 else anchor[1]

    def rep(a, dil):
        for _ in range(iterations):
            a = _rect_minmax(a, kw, kh, ax, ay, dil)
        return a

    if op == 0:
        return rep(img, False)
    if op == 1:
        return rep(img, True)
    if op == 2:
        return rep(rep(img, False), True)
    if op == 3:
        return rep(rep(img, True), False)
    raise ValueError(op)


def gaussian_blur_5x5_sigma0(gray: np.ndarray) -> np.ndarray:
    """cv2.GaussianBlur(gray, (5,5), 0): exact [1 4 6 4 1]^2 / 256, round half up, REFLECT_101."""
    k = np.array([1, 4, 6, 4, 1], dtype=np.int64)
    k2 = np.outer(k, k)
    acc = ndimage.correlate(gray.astype(np.int64), k2, mode="mirror")
    return ((acc + 128) >> 8).astype(np.uint8)


def median_blur(gray: np.ndarray, ksize: int) -> np.ndarray:
    """cv2.medianBlur(gray, ksize): median of the ksize x ksize window, BORDER_REPLICATE."""
    r = ksize // 2
    p = np.pad(gray, r, mode="edge")
    h, w = gray.shape
    stack = np.stack([p[dy : dy + h, dx : dx + w] for dy in range(ksize) for dx in range(ksize)])
    return np.sort(stack, axis=0)[ksize * ksize // 2].astype(np.uint8)


def gaussian_blur_3x3_sigma0(gray: np.ndarray) -> np.ndarray:
    k = np.array([1, 2, 1], dtype=np.int64)
    acc = ndimage.correlate(gray.astype(np.int64), np.outer(k, k), mode="mirror")
    return ((acc + 8) >> 4).astype(np.uint8)


def sobel3(gray: np.ndarray):
    """Sobel 3x3 dx, dy (int), BORDER_REPLICATE."""
    g = gray.astype(np.int32)
    kx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=np.int32)
    dx = ndimage.correlate(g, kx, mode="nearest")
    dy = ndimage.correlate(g, kx.T, mode="nearest")
    return dx, dy


def canny_map(gray: np.ndarray, t1: float, t2: float) -> np.ndarray:
    """Pre-hysteresis map of cv2.Canny(gray, t1, t2, 3): 2 strong, 0 weak, 1 suppressed."""
    if t1 > t2:
        t1, t2 = t2, t1
    low, high = math.floor(t1), math.floor(t2)
    dx, dy = sobel3(gray)
    mag = np.abs(dx) + np.abs(dy)
    h, w = gray.shape
    mp = np.pad(mag, 1)

    def nb(oy, ox):
        return mp[1 + oy : 1 + oy + h, 1 + ox : 1 + ox + w]

    ax = np.abs(dx).astype(np.int64)
    ay = np.abs(dy).astype(np.int64) << 15
    tg22 = ax * 13573
    tg67 = tg22 + (ax << 16)
    horiz = ay < tg22
    vert = (~horiz) & (ay > tg67)
    diag = ~(horiz | vert)
    same_sign = (dx ^ dy) >= 0
    is_max = np.zeros_like(mag, dtype=bool)
    is_max |= horiz & (mag > nb(0, -1)) & (mag >= nb(0, 1))
    is_max |= vert & (mag > nb(-1, 0)) & (mag >= nb(1, 0))
    is_max |= diag & same_sign & (mag > nb(-1, -1)) & (mag > nb(1, 1))
    is_max |= diag & ~same_sign & (mag > nb(-1, 1)) & (mag > nb(1, -1))
    is_max &= mag > low
    out = np.ones_like(gray, dtype=np.uint8)
    out[is_max & (mag > high)] = 2
    out[is_max & ~(mag > high)] = 0
    return out


def canny(gray: np.ndarray, t1: float, t2: float) -> np.ndarray:
    """cv2.Canny(gray, t1, t2, apertureSize=3, L2gradient=False) (lineDetection.py:9).

    Hysteresis as a connected-component statement: a candidate survives iff its
    8-connected component of {weak, strong} pixels contains a strong pixel.
    """
    m = canny_map(gray, t1, t2)
    cand = m != 1
    lab, n = ndimage.label(cand, structure=np.ones((3, 3), dtype=bool))
    keep = np.zeros(n + 1, dtype=bool)
    keep[np.unique(lab[m == 2])] = True
    keep[0] = False
    return np.where(keep[lab], 255, 0).astype(np.uint8)


def hough_dims(w: int, h: int, rho: float, theta: float):
    ft = np.float32(theta)
    na = int(math.floor(math.pi / float(ft))) + 1
    if na > 1 and abs(math.pi - (na - 1) * float(ft)) < float(ft) / 2:
        na -= 1
    numrho = int(np.rint(np.float32(2 * (w + h) + 1) / np.float32(rho)))
    return na, numrho


def hough_trig(rho: float, theta: float, numangle: int):
    irho = np.float32(1) / np.float32(rho)
    ft = np.float32(theta)
    ang = np.float32(0)
    ts = np.empty(numangle, np.float32)
    tc = np.empty(numangle, np.float32)
    for n in range(numangle):
        ts[n] = np.float32(math.sin(float(ang)) * float(irho))
        tc[n] = np.float32(math.cos(float(ang)) * float(irho))
        ang = np.float32(ang + ft)
    return ts, tc


def hough_accum(img: np.ndarray, rho: float, theta: float) -> np.ndarray:
    """Vote accumulator of cv2.HoughLines: int32[numangle+2, numrho+2] ("New Version/comparison.py":20)."""
    h, w = img.shape
    na, nr = hough_dims(w, h, rho, theta)
    ts, tc = hough_trig(rho, theta, na)
    ys, xs = np.nonzero(img)
    acc = np.zeros((na + 2, nr + 2), np.int32)
    xf = xs.astype(np.float32)
    yf = ys.astype(np.float32)
    for n in range(na):
        a = xf * tc[n]  # float32 product, rounded
        b = yf * ts[n]
        r = np.rint(a + b).astype(np.int64) + (nr - 1) // 2
        acc[n + 1] += np.bincount(r + 1, minlength=nr + 2).astype(np.int32)
    return acc


def hough_lines(img: np.ndarray, rho: float, theta: float, threshold: int):
    """cv2.HoughLines(img, rho, theta, threshold) -> float32[N,1,2] or None."""
    acc = hough_accum(img, rho, theta)
    na, nr = acc.shape[0] - 2, acc.shape[1] - 2
    c = acc[1:-1, 1:-1]
    peak = (
        (c > threshold)
        & (c > acc[1:-1, 0:-2])
        & (c >= acc[1:-1, 2:])
        & (c > acc[0:-2, 1:-1])
        & (c >= acc[2:, 1:-1])
    )
    n_idx, r_idx = np.nonzero(peak)
    if n_idx.size == 0:
        return None
    base = (n_idx + 1) * (nr + 2) + r_idx + 1
    order = np.lexsort((base, -c[n_idx, r_idx].astype(np.int64)))
    n_idx, r_idx = n_idx[order], r_idx[order]
    rho_out = (r_idx.astype(np.float32) - np.float32((nr - 1) * 0.5)) * np.float32(rho)
    theta_out = n_idx.astype(np.float32) * np.float32(theta)
    return np.stack([rho_out, theta_out], axis=1).astype(np.float32)[:, None, :]


def adaptive_threshold_mean_ref(gray, maxv, typ, block, C):
    """cv2.adaptiveThreshold(..., ADAPTIVE_THRESH_MEAN_C, ...): integer box mean, REPLICATE."""
    s = ndimage.correlate(gray.astype(np.int64), np.ones((block, block), np.int64), mode="nearest")
    mean = np.clip(np.rint(s * (1.0 / (block * block))), 0, 255).astype(np.int32)
    return _adaptive_compare(gray, mean, maxv, typ, C)


def _adaptive_compare(gray, mean, maxv, typ, C):
    mv = int(np.clip(np.rint(maxv), 0, 255))
    d = gray.astype(np.int32) - mean
    if typ == 0:
        return np.where(d > -math.ceil(C), mv, 0).astype(np.uint8)
    return np.where(d <= -math.floor(C), mv, 0).astype(np.uint8)


def gaussian_kernel_f32(n: int) -> np.ndarray:
    """float32 Gaussian taps for sigma <= 0 (sigma derived from n), as cv2.getGaussianKernel."""
    tabs = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
            7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}
    if n in tabs:
        return np.array(tabs[n], np.float32)
    sigma = ((n - 1) * 0.5 - 1) * 0.3 + 0.8
    x = np.arange(n, dtype=np.float64) - (n - 1) * 0.5
    k = np.exp(-0.5 / (sigma * sigma) * x * x)
    k = k * (1.0 / k.sum())
    return k.astype(np.float32)


def adaptive_threshold_gaussian_ref(gray, maxv, typ, block, C):
    """cv2.adaptiveThreshold(..., ADAPTIVE_THRESH_GAUSSIAN_C, ...) (NewVersion2/img2UML.py:29).

    float32 arithmetic with every multiply and add rounded separately: rows
    accumulate taps left to right, columns as k0*x0 + sum k_i*(x_+i + x_-i).
    """
    k = gaussian_kernel_f32(block)
    r = block // 2
    h, w = gray.shape
    src = np.pad(gray, ((0, 0), (r, r)), mode="edge").astype(np.float32)
    acc = np.zeros((h, w), np.float32)
    for i in range(block):
        acc = acc + src[:, i : i + w] * k[i]
    tp = np.pad(acc, ((r, r), (0, 0)), mode="edge")
    out = tp[r : r + h] * k[r]
    for i in range(1, r + 1):
        out = out + (tp[r + i : r + i + h] + tp[r - i : r - i + h]) * k[r + i]
    mean = np.clip(np.rint(out), 0, 255).astype(np.int32)
    return _adaptive_compare(gray, mean, maxv, typ, C)


def external_boxes(mask: np.ndarray) -> np.ndarray:
    """Topological statement of findContours(RETR_EXTERNAL) + boundingRect: the 8-connected
    components of the non-zero pixels that are not enclosed by another component (they touch
    the image border or the 4-connected background region that reaches the border), as
    (x, y, w, h), ordered by their first pixel in raster order, last first (cv2's order)."""
    from scipy import ndimage

    m = np.asarray(mask) != 0
    h, w = m.shape
    if h == 0 or w == 0:
        return np.zeros((0, 4), np.int32)
    fg, nf = ndimage.label(m, structure=np.ones((3, 3), int))
    # background with a one-pixel frame, 4-connected; the frame's region is the outside
    bgm = np.pad(~m, 1, constant_values=True)
    bg, _ = ndimage.label(bgm, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    outer = bg == bg[0, 0]
    # pixels of the image that have an outside pixel among their 4 neighbours
    near = np.zeros_like(m)
    o = outer
    near |= o[1:-1, :-2] | o[1:-1, 2:] | o[:-2, 1:-1] | o[2:, 1:-1]
    ext = np.unique(fg[m & near])
    boxes = []
    objs = ndimage.find_objects(fg)
    first = {}
    ys, xs = np.nonzero(m)
    lab = fg[ys, xs]
    _, idx = np.unique(lab, return_index=True)  # raster order: first occurrence
    for k, i in zip(lab[idx], idx):
        first[int(k)] = int(ys[i]) * w + int(xs[i])
    for k in ext:
        sl = objs[int(k) - 1]
        boxes.append((first[int(k)], sl[1].start, sl[0].start, sl[1].stop - sl[1].start, sl[0].stop - sl[0].start))
    boxes.sort(key=lambda t: -t[0])
    return np.array([b[1:] for b in boxes], np.int32).reshape(-1, 4)
