"""ctypes binding of ``oracle/libfrontref.so`` (TEST INFRASTRUCTURE ONLY).

Loads the C restatement built by ``oracle/Makefile`` and exposes numpy-in /
numpy-out wrappers.  Only tests, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` import this module; the product package
never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfrontref.so")
_lib = None

u8p = C.POINTER(C.c_uint8)


def build() -> str:
    """Compile the C restatement if the shared object is missing or stale."""
    src = os.path.join(_HERE, "frontref.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libfrontref.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.fr_threshold.restype = C.c_double
        _lib.fr_otsu_from_hist.restype = C.c_int
        _lib.fr_hough_peaks.restype = C.c_int
        _lib.fr_gaussian_blur_u8.restype = C.c_int
        _lib.fr_adaptive_threshold.restype = C.c_int
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(u8p)


def _c(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a)
    assert a.dtype == np.uint8
    return a


def bgr2gray(bgr):
    bgr = _c(bgr)
    h, w, _ = bgr.shape
    out = np.empty((h, w), np.uint8)
    lib().fr_bgr2gray(_p(bgr), C.c_size_t(3 * w), _p(out), C.c_size_t(w), w, h)
    return out


def gray2bgr(gray):
    gray = _c(gray)
    h, w = gray.shape
    out = np.empty((h, w, 3), np.uint8)
    lib().fr_gray2bgr(_p(gray), C.c_size_t(w), _p(out), C.c_size_t(3 * w), w, h)
    return out


def swap_rb(img):
    img = _c(img)
    h, w, _ = img.shape
    out = np.empty_like(img)
    lib().fr_swap_rb(_p(img), C.c_size_t(3 * w), _p(out), C.c_size_t(3 * w), w, h)
    return out


def bitwise_not(a):
    a = _c(a)
    h = a.shape[0]
    rb = a.size // max(h, 1)
    out = np.empty_like(a)
    lib().fr_bitwise_not(_p(a), C.c_size_t(rb), _p(out), C.c_size_t(rb), rb, h)
    return out


def bitwise_or(a, b):
    a, b = _c(a), _c(b)
    h = a.shape[0]
    rb = a.size // max(h, 1)
    out = np.empty_like(a)
    lib().fr_bitwise_or(_p(a), C.c_size_t(rb), _p(b), C.c_size_t(rb), _p(out), C.c_size_t(rb), rb, h)
    return out


def histogram(gray):
    gray = _c(gray)
    h, w = gray.shape
    hist = np.zeros(256, np.int32)
    lib().fr_histogram(_p(gray), C.c_size_t(w), w, h, hist.ctypes.data_as(C.POINTER(C.c_int)))
    return hist


def otsu_from_hist(hist, npix):
    hist = np.ascontiguousarray(hist, np.int32)
    return lib().fr_otsu_from_hist(hist.ctypes.data_as(C.POINTER(C.c_int)), C.c_longlong(npix))


def threshold(gray, thresh, maxval, typ):
    gray = _c(gray)
    h, w = gray.shape
    out = np.empty((h, w), np.uint8)
    r = lib().fr_threshold(_p(gray), C.c_size_t(w), _p(out), C.c_size_t(w), w, h,
                           C.c_double(thresh), C.c_double(maxval), int(typ))
    return float(r), out


def morphology_ex(img, op, ksize, iterations=1, anchor=(-1, -1)):
    img = _c(img)
    h, w = img.shape
    out = np.empty((h, w), np.uint8)
    lib().fr_morphology_ex(_p(img), C.c_size_t(w), _p(out), C.c_size_t(w), w, h, int(op),
                           int(ksize[0]), int(ksize[1]), int(anchor[0]), int(anchor[1]), int(iterations))
    return out


def gaussian_kernel_q8(n, sigma):
    k = np.zeros(n, np.int32)
    lib().fr_gaussian_kernel_q8(int(n), C.c_double(sigma), k.ctypes.data_as(C.POINTER(C.c_int)))
    return k


def gaussian_blur(gray, ksize, sigmaX, sigmaY=0.0, border=4):
    gray = _c(gray)
    h, w = gray.shape
    out = np.empty((h, w), np.uint8)
    rc = lib().fr_gaussian_blur_u8(_p(gray), C.c_size_t(w), _p(out), C.c_size_t(w), w, h,
                                   int(ksize[0]), int(ksize[1]), C.c_double(sigmaX), C.c_double(sigmaY), int(border))
    if rc != 0:
        raise ValueError("bad GaussianBlur arguments")
    return out


def median_blur(gray, ksize):
    """cv2.medianBlur(gray, ksize) for ksize 3 or 5."""
    gray = _c(gray)
    h, w = gray.shape
    out = np.empty((h, w), np.uint8)
    rc = lib().fr_median_blur_u8(_p(gray), C.c_size_t(w), _p(out), C.c_size_t(w), w, h, int(ksize))
    if rc != 0:
        raise ValueError("bad medianBlur arguments")
    return out


def adaptive_threshold(gray, maxv, method, typ, block, Cc, return_mean=False):
    gray = _c(gray)
    h, w = gray.shape
    out = np.empty((h, w), np.uint8)
    mean = np.empty((h, w), np.uint8)
    rc = lib().fr_adaptive_threshold(_p(gray), C.c_size_t(w), _p(out), C.c_size_t(w), w, h,
                                     C.c_double(maxv), int(method), int(typ), int(block), C.c_double(Cc), _p(mean))
    if rc != 0:
        raise ValueError("bad adaptiveThreshold arguments")
    return (out, mean) if return_mean else out


def canny(gray, t1, t2, return_map=False):
    gray = _c(gray)
    h, w = gray.shape
    out = np.empty((h, w), np.uint8)
    m = np.empty((h, w), np.uint8)
    lib().fr_canny(_p(gray), C.c_size_t(w), _p(out), C.c_size_t(w), w, h, C.c_double(t1), C.c_double(t2), _p(m))
    return (out, m) if return_map else out


def hough_dims(w, h, rho, theta):
    na, nr = C.c_int(), C.c_int()
    lib().fr_hough_dims(int(w), int(h), C.c_double(rho), C.c_double(theta), C.byref(na), C.byref(nr))
    return na.value, nr.value


def hough_trig(rho, theta, numangle):
    ts = np.empty(numangle, np.float32)
    tc = np.empty(numangle, np.float32)
    fp = C.POINTER(C.c_float)
    lib().fr_hough_trig(C.c_double(rho), C.c_double(theta), int(numangle), ts.ctypes.data_as(fp), tc.ctypes.data_as(fp))
    return ts, tc


def hough_accum(img, rho, theta):
    img = _c(img)
    h, w = img.shape
    na, nr = hough_dims(w, h, rho, theta)
    acc = np.empty((na + 2, nr + 2), np.int32)
    lib().fr_hough_accum(_p(img), C.c_size_t(w), w, h, C.c_double(rho), C.c_double(theta),
                         acc.ctypes.data_as(C.POINTER(C.c_int)), na, nr)
    return acc


def hough_peaks(acc, rho, theta, threshold, max_lines=1 << 22):
    acc = np.ascontiguousarray(acc, np.int32)
    na, nr = acc.shape[0] - 2, acc.shape[1] - 2
    lines = np.empty((max_lines, 2), np.float32)
    n = lib().fr_hough_peaks(acc.ctypes.data_as(C.POINTER(C.c_int)), na, nr, C.c_double(rho), C.c_double(theta),
                             int(threshold), lines.ctypes.data_as(C.POINTER(C.c_float)), int(max_lines))
    if n == 0:
        return None
    return lines[: min(n, max_lines)].reshape(-1, 1, 2).copy()


def hough_lines(img, rho, theta, threshold):
    return hough_peaks(hough_accum(img, rho, theta), rho, theta, threshold)


def external_boxes(mask, cap=1 << 20):
    """Bounding rectangles (x, y, w, h) of cv2.findContours(mask, RETR_EXTERNAL, ...) in cv2's order."""
    m = _c(np.asarray(mask, dtype=np.uint8))
    h, w = m.shape
    out = np.zeros((max(cap, 1), 4), np.int32)
    f = lib().fr_external_boxes
    f.restype = C.c_int
    n = f(_p(m), C.c_size_t(m.strides[0] if h else 0), C.c_int(w), C.c_int(h), out.ctypes.data_as(C.c_void_p), C.c_int(cap))
    if n < 0:
        raise MemoryError("fr_external_boxes")
    return out[: min(n, cap)].copy()


def hough_lines_p(img, rho, theta, threshold, min_line_length=0, max_line_gap=0, cap=1 << 16):
    """cv2.HoughLinesP(img, rho, theta, threshold, minLineLength=, maxLineGap=): int32[N,1,4] or None."""
    m = _c(np.asarray(img, dtype=np.uint8))
    h, w = m.shape
    out = np.zeros((max(cap, 1), 4), np.int32)
    f = lib().fr_hough_lines_p
    f.restype = C.c_int
    n = f(_p(m), C.c_size_t(m.strides[0] if h else 0), C.c_int(w), C.c_int(h), C.c_double(rho), C.c_double(theta),
          C.c_int(int(threshold)), C.c_int(int(min_line_length)), C.c_int(int(max_line_gap)),
          out.ctypes.data_as(C.c_void_p), C.c_int(cap))
    if n < 0:
        raise MemoryError("fr_hough_lines_p")
    if n > cap:
        return hough_lines_p(img, rho, theta, threshold, min_line_length, max_line_gap, cap=n)
    return out[:n].reshape(n, 1, 4).copy() if n else None


def front_end(bgr, t=128, line_len=15, iterations=2, c_lo=50.0, c_hi=150.0, with_hough=True):
    """gray, thresh, vlines, hlines, edges, accum for one BGR image (CPU baseline unit of work)."""
    bgr = _c(bgr)
    h, w, _ = bgr.shape
    planes = [np.empty((h, w), np.uint8) for _ in range(5)]
    na, nr = hough_dims(w, h, 1.0, np.pi / 180)
    acc = np.empty((na + 2, nr + 2), np.int32) if with_hough else None
    lib().fr_front_end(_p(bgr), w, h, int(t), int(line_len), int(iterations), C.c_double(c_lo), C.c_double(c_hi),
                       *[_p(p) for p in planes],
                       acc.ctypes.data_as(C.POINTER(C.c_int)) if with_hough else None, na, nr)
    return (*planes, acc)
