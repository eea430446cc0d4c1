"""cv2-style call signatures over the HIP kernels.

The reference's diagram parsers do ``import cv2`` and call free functions on
numpy arrays (e.g. umlClassdiagram.py:12-30, lineDetection.py:5-12).  This
module exposes the subset of those functions that forms their raster front-end,
with the same names, argument order, return conventions and error behaviour, so
that a script can use ``import imageprocessing_amd as cv2`` for that prefix.

Inputs may be numpy ``uint8`` arrays (results come back as numpy arrays) or
ROCm ``torch.uint8`` tensors (results stay on the device).  All pixel work runs
in ``libipamd_hip.so``; there is no CPU implementation behind these names.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _native
from ._device import DevImage, check, error, stream_ptr, upload, workspace

# ---- constants (values as in cv2) ------------------------------------------------
COLOR_BGR2RGB = 4
COLOR_RGB2BGR = 4
COLOR_BGR2GRAY = 6
COLOR_GRAY2BGR = 8
COLOR_GRAY2RGB = 8  # (the same conversion: three copies of the gray value)
THRESH_BINARY = 0
THRESH_BINARY_INV = 1
THRESH_OTSU = 8
ADAPTIVE_THRESH_MEAN_C = 0
ADAPTIVE_THRESH_GAUSSIAN_C = 1
MORPH_RECT = 0
MORPH_ERODE = 0
MORPH_DILATE = 1
MORPH_OPEN = 2
MORPH_CLOSE = 3
BORDER_CONSTANT = 0
BORDER_REPLICATE = 1
BORDER_REFLECT = 2
BORDER_REFLECT_101 = 4
BORDER_REFLECT101 = 4
BORDER_DEFAULT = 4


def _is_torch(x) -> bool:
    return isinstance(x, torch.Tensor)


def _ret(img: DevImage, like):
    return img.to_tensor(True) if _is_torch(like) else img.to_numpy(True)


# ---- colour ----------------------------------------------------------------------
def cvtColor(src, code, dst=None, dstCn=0):
    """cv2.cvtColor for COLOR_BGR2GRAY, COLOR_GRAY2BGR, COLOR_BGR2RGB / COLOR_RGB2BGR
    (umlClassdiagram.py:14,1103; NewVersion2/text_detectionv2.py:17)."""
    L = _native.lib()
    if code == COLOR_BGR2GRAY:
        s = upload(src, 3)
        d = DevImage.empty(1, s.height, s.width, 1)
        check(L.ipa_bgr2gray_u8(s.ptr, s.pitch, s.stride, d.ptr, d.pitch, d.stride, s.width, s.height, 1, stream_ptr()))
    elif code == COLOR_GRAY2BGR:
        s = upload(src, 1)
        d = DevImage.empty(1, s.height, s.width, 3)
        check(L.ipa_gray2bgr_u8(s.ptr, s.pitch, s.stride, d.ptr, d.pitch, d.stride, s.width, s.height, 1, stream_ptr()))
    elif code == COLOR_BGR2RGB:
        s = upload(src, 3)
        d = DevImage.empty(1, s.height, s.width, 3)
        check(L.ipa_swap_rb_u8(s.ptr, s.pitch, s.stride, d.ptr, d.pitch, d.stride, s.width, s.height, 1, stream_ptr()))
    else:
        raise error(f"cvtColor: unsupported conversion code {code}")
    return _ret(d, src)


def _bitwise_shape(a):
    nd = a.dim() if _is_torch(a) else np.asarray(a).ndim
    if nd == 2:
        return 1
    if nd == 3 and a.shape[-1] == 3:
        return 3
    if nd == 3 and a.shape[-1] == 1:
        return 1
    raise error(f"bitwise op: unsupported array shape {tuple(a.shape)}")


def bitwise_not(src, dst=None, mask=None):
    """cv2.bitwise_not (NewVersion2/text_detectionv2.py:22)."""
    if mask is not None:
        raise error("bitwise_not: mask is not supported")
    ch = _bitwise_shape(src)
    s = upload(src, ch)
    d = DevImage.empty(1, s.height, s.width, ch)
    check(_native.lib().ipa_bitwise_not_u8(s.ptr, s.pitch, s.stride, d.ptr, d.pitch, d.stride, s.row_bytes, s.height, 1,
                                           stream_ptr()))
    return _ret(d, src)


def bitwise_or(src1, src2, dst=None, mask=None):
    """cv2.bitwise_or ("New Version/textDetection copy.py":94)."""
    if mask is not None:
        raise error("bitwise_or: mask is not supported")
    ch = _bitwise_shape(src1)
    a, b = upload(src1, ch), upload(src2, _bitwise_shape(src2))
    if (a.height, a.width, a.channels) != (b.height, b.width, b.channels):
        raise error("bitwise_or: the input arrays differ in size or channels")
    d = DevImage.empty(1, a.height, a.width, ch)
    check(_native.lib().ipa_bitwise_or_u8(a.ptr, a.pitch, a.stride, b.ptr, b.pitch, b.stride, d.ptr, d.pitch, d.stride,
                                          a.row_bytes, a.height, 1, stream_ptr()))
    return _ret(d, src1)


# ---- thresholds ------------------------------------------------------------------
def threshold(src, thresh, maxval, type, dst=None):
    """cv2.threshold -> (retval, dst) for THRESH_BINARY / THRESH_BINARY_INV, optionally
    | THRESH_OTSU (umlClassdiagram.py:19; "New Version/textDetection.py":23)."""
    L = _native.lib()
    base = int(type) & 7
    if base not in (THRESH_BINARY, THRESH_BINARY_INV) or int(type) & ~(7 | THRESH_OTSU):
        raise error(f"threshold: unsupported threshold type {type}")
    s = upload(src, 1)
    d = DevImage.empty(1, s.height, s.width, 1)
    if int(type) & THRESH_OTSU:
        buf = torch.empty(257, dtype=torch.int32, device=s.t.device)  # histogram and the threshold (both written by the call)
        hist, otsu = buf[:256], buf[256:]
        check(L.ipa_threshold_otsu_u8(s.ptr, s.pitch, s.stride, d.ptr, d.pitch, d.stride, s.width, s.height, 1,
                                      float(maxval), base, hist.data_ptr(), otsu.data_ptr(), stream_ptr()))
        ret = float(otsu.item()) if s.width * s.height else 0.0  # (an empty image: nothing is written)
    else:
        check(L.ipa_threshold_u8(s.ptr, s.pitch, s.stride, d.ptr, d.pitch, d.stride, s.width, s.height, 1,
                                 float(thresh), float(maxval), base, stream_ptr()))
        ret = float(math.floor(thresh))
    return ret, _ret(d, src)


def adaptiveThreshold(src, maxValue, adaptiveMethod, thresholdType, blockSize, C, dst=None):
    """cv2.adaptiveThreshold (NewVersion2/img2UML.py:29; newVersion3/detect_arrow.py:19)."""
    L = _native.lib()
    if blockSize % 2 != 1 or blockSize <= 1:
        raise error("adaptiveThreshold: blockSize % 2 == 1 && blockSize > 1")
    if adaptiveMethod not in (ADAPTIVE_THRESH_MEAN_C, ADAPTIVE_THRESH_GAUSSIAN_C):
        raise error("adaptiveThreshold: Unknown/unsupported adaptive threshold method")
    if thresholdType not in (THRESH_BINARY, THRESH_BINARY_INV):
        raise error("adaptiveThreshold: Unknown/unsupported threshold type")
    s = upload(src, 1)
    d = DevImage.empty(1, s.height, s.width, 1)
    ws = workspace(L.ipa_adaptive_threshold_workspace_bytes(s.width, s.height, 1))
    check(L.ipa_adaptive_threshold_u8(s.ptr, s.pitch, s.stride, d.ptr, d.pitch, d.stride, s.width, s.height, 1,
                                      float(maxValue), int(adaptiveMethod), int(thresholdType), int(blockSize), float(C),
                                      ws.data_ptr(), ws.numel(), stream_ptr()))
    return _ret(d, src)


# ---- smoothing -------------------------------------------------------------------
def GaussianBlur(src, ksize, sigmaX, dst=None, sigmaY=0, borderType=BORDER_DEFAULT):
    """cv2.GaussianBlur on 8-bit single-channel images (NewVersion2/img2UML.py:26;
    "New Version/detectArrow.py":7)."""
    L = _native.lib()
    kw, kh = int(ksize[0]), int(ksize[1])
    s = upload(src, 1)
    d = DevImage.empty(1, s.height, s.width, 1)
    ws = workspace(L.ipa_gaussian_blur_workspace_bytes(s.width, s.height, 1))
    check(L.ipa_gaussian_blur_u8(s.ptr, s.pitch, s.stride, d.ptr, d.pitch, d.stride, s.width, s.height, 1, kw, kh,
                                 float(sigmaX), float(sigmaY), int(borderType) & ~16, ws.data_ptr(), ws.numel(),
                                 stream_ptr()))
    return _ret(d, src)


def medianBlur(src, ksize, dst=None):
    """cv2.medianBlur on 8-bit single-channel images, ksize 3 or 5 ("New Version/textDetection.py":16,
    "New Version/detectCircle.py":14, "New Version/detectShape.py":11)."""
    L = _native.lib()
    k = int(ksize)
    if k % 2 != 1 or k <= 1:
        raise error("medianBlur: ksize must be odd and greater than 1")
    if k not in (3, 5):
        raise error("medianBlur: only ksize 3 and 5 are supported")
    s = upload(src, 1)
    d = DevImage.empty(1, s.height, s.width, 1)
    check(L.ipa_median_blur_u8(s.ptr, s.pitch, s.stride, d.ptr, d.pitch, d.stride, s.width, s.height, 1, k, stream_ptr()))
    return _ret(d, src)


# ---- morphology ------------------------------------------------------------------
def getStructuringElement(shape, ksize, anchor=(-1, -1)):
    """cv2.getStructuringElement(MORPH_RECT, (cols, rows)) -> uint8[rows, cols] of ones
    (umlClassdiagram.py:24-25)."""
    if shape != MORPH_RECT:
        raise error("getStructuringElement: only MORPH_RECT is supported")
    kw, kh = int(ksize[0]), int(ksize[1])
    if kw <= 0 or kh <= 0:
        raise error("getStructuringElement: ksize must be positive")
    return np.ones((kh, kw), np.uint8)


def _rect_kernel(kernel):
    """(kw, kh) of an all-ones kernel; None/empty -> 3x3 as in cv2."""
    if kernel is None:
        return 3, 3
    k = kernel.cpu().numpy() if _is_torch(kernel) else np.asarray(kernel)
    if k.size == 0:
        return 3, 3
    if k.ndim == 1:
        k = k[:, None]
    if k.ndim != 2:
        raise error("morphology: kernel must be 2-D")
    if not np.all(k != 0):
        raise error("morphology: only all-ones (rectangular) structuring elements are supported")
    return int(k.shape[1]), int(k.shape[0])


def _morph(src, op, kernel, anchor, iterations, borderType, borderValue):
    L = _native.lib()
    if borderType not in (None, BORDER_CONSTANT) or borderValue is not None:
        raise error("morphology: only the default border (BORDER_CONSTANT, morphologyDefaultBorderValue) is supported")
    kw, kh = _rect_kernel(kernel)
    ax, ay = (int(anchor[0]), int(anchor[1])) if anchor is not None else (-1, -1)
    if ax >= kw or ay >= kh:
        raise error("morphology: anchor is outside the kernel")
    s = upload(src, 1)
    d = DevImage.empty(1, s.height, s.width, 1)
    ws = workspace(L.ipa_morph_workspace_bytes(s.width, s.height, 1))
    # Content unknown (assume_binary = 0): the library tries its one-pass kernel for two-valued images first, which
    # gives up on an image as soon as it meets a third value, and runs the gray-level kernels only then.
    # Rectangles beyond what that kernel covers (longer than 65 or higher than 31 after folding the iterations):
    # two-valued images take the bit-plane kernels, which is worth a pass that looks at the pixels.
    n = max(int(iterations), 1)
    fx, fy = kw + (n - 1) * (kw - 1), kh + (n - 1) * (kh - 1)
    cx = (ax if ax >= 0 else kw // 2) * n
    binary = 0
    if cx > 32 or fx - 1 - cx > 32 or fy > 31:
        flags = torch.zeros(1, dtype=torch.int32, device=s.t.device)
        check(L.ipa_binary_check_u8(s.ptr, s.pitch, s.stride, s.width, s.height, 1, flags.data_ptr(), stream_ptr()))
        binary = 1 if int(flags.item()) >= 0 else 0
    check(L.ipa_morph_u8(s.ptr, s.pitch, s.stride, d.ptr, d.pitch, d.stride, s.width, s.height, 1, int(op), kw, kh,
                         ax, ay, int(iterations), binary, ws.data_ptr(), ws.numel(), stream_ptr()))
    return _ret(d, src)


def erode(src, kernel, dst=None, anchor=(-1, -1), iterations=1, borderType=None, borderValue=None):
    """cv2.erode ("New Version/classDiagram copy.py":317)."""
    return _morph(src, MORPH_ERODE, kernel, anchor, iterations, borderType, borderValue)


def dilate(src, kernel, dst=None, anchor=(-1, -1), iterations=1, borderType=None, borderValue=None):
    """cv2.dilate ("New Version/detectArrow.py":10; NewVersion2/detectArrowv2.py:24)."""
    return _morph(src, MORPH_DILATE, kernel, anchor, iterations, borderType, borderValue)


def morphologyEx(src, op, kernel, dst=None, anchor=(-1, -1), iterations=1, borderType=None, borderValue=None):
    """cv2.morphologyEx for MORPH_ERODE/DILATE/OPEN/CLOSE (umlClassdiagram.py:27-28)."""
    if op not in (MORPH_ERODE, MORPH_DILATE, MORPH_OPEN, MORPH_CLOSE):
        raise error(f"morphologyEx: unsupported operation {op}")
    return _morph(src, op, kernel, anchor, iterations, borderType, borderValue)


# ---- Canny / Hough ----------------------------------------------------------------
def Canny(image, threshold1, threshold2, edges=None, apertureSize=3, L2gradient=False):
    """cv2.Canny(image, threshold1, threshold2, apertureSize=3, L2gradient=False)
    (lineDetection.py:9)."""
    L = _native.lib()
    if apertureSize != 3:
        raise error("Canny: only apertureSize=3 is supported")
    if L2gradient:
        raise error("Canny: only L2gradient=False is supported")
    s = upload(image, 1)
    d = DevImage.empty(1, s.height, s.width, 1)
    ws = workspace(L.ipa_canny_workspace_bytes(s.width, s.height, 1))
    check(L.ipa_canny_u8(s.ptr, s.pitch, s.stride, d.ptr, d.pitch, d.stride, s.width, s.height, 1, float(threshold1),
                         float(threshold2), ws.data_ptr(), ws.numel(), stream_ptr()))
    return _ret(d, image)


def hough_dims(width, height, rho, theta):
    na, nr = C.c_int(), C.c_int()
    check(_native.lib().ipa_hough_dims(int(width), int(height), float(rho), float(theta), C.byref(na), C.byref(nr)))
    return na.value, nr.value


def hough_accumulator(image, rho, theta, with_seg_max=False):
    """Vote accumulator of cv2.HoughLines as a device tensor int32[numangle+2, numrho+2]
    (and, on request, the per-segment maxima int32[numangle, segments] that speed up the peak search).

    cv2 does not expose it; it is returned here because parity is defined on its counts."""
    L = _native.lib()
    s = upload(image, 1)
    na, nr = hough_dims(s.width, s.height, rho, theta)
    acc = torch.empty((na + 2, nr + 2), dtype=torch.int32, device=s.t.device)
    seg = torch.empty((na, L.ipa_hough_segments(nr)), dtype=torch.int32, device=s.t.device)
    counts = torch.zeros(1, dtype=torch.int32, device=s.t.device)
    ws = workspace(L.ipa_hough_workspace_bytes(s.width, s.height, 1, 0))
    check(L.ipa_hough_accum_u8(s.ptr, s.pitch, s.stride, s.width, s.height, 1, float(rho), float(theta), acc.data_ptr(),
                               acc.numel(), seg.data_ptr(), 0, counts.data_ptr(), ws.data_ptr(), ws.numel(), stream_ptr()))
    return (acc, seg) if with_seg_max else acc


def lines_from_keys(keys: np.ndarray, numrho: int, rho: float, theta: float):
    """Sorted peak keys -> float32[N,1,2] (rho, theta) exactly as cv2 emits them."""
    keys = np.sort(keys)[::-1]
    idx = (0xFFFFFFFF - (keys & np.uint64(0xFFFFFFFF))).astype(np.int64)
    scale = 1.0 / (numrho + 2)
    n = np.floor(idx * scale).astype(np.int64) - 1
    r = idx - (n + 1) * (numrho + 2) - 1
    rho32, theta32 = np.float32(rho), np.float32(theta)
    out_rho = (r.astype(np.float32) - np.float32((numrho - 1) * 0.5)) * rho32
    out_theta = np.float32(0) + n.astype(np.float32) * theta32
    return np.stack([out_rho, out_theta], axis=1).astype(np.float32)[:, None, :]


def hough_peaks(acc: torch.Tensor, threshold: int, capacity: int = 4096, seg_max: torch.Tensor = None) -> np.ndarray:
    """Unsorted 64-bit peak keys of one accumulator (device), growing the buffer as needed.
    ``seg_max`` = the per-segment maxima produced together with ``acc`` (optional)."""
    L = _native.lib()
    na, nr = acc.shape[0] - 2, acc.shape[1] - 2
    while True:
        keys = torch.empty(max(capacity, 1), dtype=torch.int64, device=acc.device)
        cnt = torch.zeros(1, dtype=torch.int32, device=acc.device)
        check(L.ipa_hough_peaks_i32(acc.data_ptr(), acc.numel(), seg_max.data_ptr() if seg_max is not None else None, na, nr, 1,
                                    int(threshold), keys.data_ptr(), capacity,
                                    cnt.data_ptr(), stream_ptr()))
        n = int(cnt.item())
        if n <= capacity:
            return keys[:n].cpu().numpy().view(np.uint64)
        capacity = n


def hough_peaks_batch(acc: torch.Tensor, threshold: int, capacity: int = 4096, seg_max: torch.Tensor = None):
    """Unsorted peak keys of a batch of accumulators int32[N, numangle+2, numrho+2] in ONE device call:
    list of N uint64 arrays."""
    L = _native.lib()
    n, na, nr = acc.shape[0], acc.shape[1] - 2, acc.shape[2] - 2
    if n == 0:
        return []
    while True:
        keys = torch.empty((n, max(capacity, 1)), dtype=torch.int64, device=acc.device)
        cnt = torch.zeros(n, dtype=torch.int32, device=acc.device)
        check(L.ipa_hough_peaks_i32(acc.data_ptr(), acc.stride(0), seg_max.data_ptr() if seg_max is not None else None, na, nr, n,
                                    int(threshold), keys.data_ptr(), capacity, cnt.data_ptr(), stream_ptr()))
        c = cnt.cpu().numpy()
        if c.max(initial=0) <= capacity:
            host = keys.cpu().numpy().view(np.uint64)
            return [host[i, : int(c[i])].copy() for i in range(n)]
        capacity = int(c.max())


def HoughLines(image, rho, theta, threshold, lines=None, srn=0, stn=0, min_theta=0, max_theta=np.pi):
    """cv2.HoughLines(image, rho, theta, threshold) -> float32[N,1,2] or None
    ("New Version/comparison.py":20,30)."""
    if srn != 0 or stn != 0:
        raise error("HoughLines: the multi-scale variant (srn/stn) is not supported")
    if min_theta != 0 or abs(max_theta - np.pi) > 1e-12:
        raise error("HoughLines: only min_theta=0, max_theta=pi is supported")
    if rho <= 0 or theta <= 0:
        raise error("HoughLines: rho and theta must be greater 0")
    acc, seg = hough_accumulator(image, rho, theta, with_seg_max=True)
    keys = hough_peaks(acc, int(threshold), seg_max=seg)
    if keys.size == 0:
        return None
    out = lines_from_keys(keys, acc.shape[1] - 2, rho, theta)
    return torch.from_numpy(out).to(acc.device) if _is_torch(image) else out


def hough_lines_p_batch(images, rho, theta, threshold, minLineLength=0, maxLineGap=0, max_lines=4096):
    """cv2.HoughLinesP for a batch uint8[N,H,W] (numpy or ROCm tensor): per image int32[n,1,4]
    (x0, y0, x1, y1) in the order cv2 finds the segments, or None."""
    L = _native.lib()
    if rho <= 0 or theta <= 0:
        raise error("HoughLinesP: rho and theta must be greater 0")
    s = upload(images, 1, batched=True)
    n = s.batch
    if n == 0:
        return []
    dev = s.t.device
    ws = workspace(L.ipa_hough_lines_p_workspace_bytes(s.width, s.height, n, float(rho), float(theta), 0))
    while True:
        lines = torch.empty((n, max(max_lines, 1), 4), dtype=torch.int32, device=dev)
        counts = torch.zeros(n, dtype=torch.int32, device=dev)
        check(L.ipa_hough_lines_p_u8(s.ptr, s.pitch, s.stride, s.width, s.height, n, float(rho), float(theta), int(threshold),
                                     int(minLineLength), int(maxLineGap), 0, lines.data_ptr(), int(max_lines),
                                     counts.data_ptr(), ws.data_ptr(), ws.numel(), stream_ptr()))
        c = counts.cpu().numpy()
        if (c < 0).any():
            raise error("HoughLinesP: internal list capacity exceeded")
        if c.max(initial=0) > max_lines:
            max_lines = int(c.max())
            continue
        out = []
        for i in range(n):
            if c[i] == 0:
                out.append(None)
            else:
                seg = lines[i, : int(c[i])].reshape(-1, 1, 4)
                out.append(seg.clone() if _is_torch(images) else seg.cpu().numpy())
        return out


def HoughLinesP(image, rho, theta, threshold, lines=None, minLineLength=0, maxLineGap=0):
    """cv2.HoughLinesP(image, rho, theta, threshold, minLineLength=, maxLineGap=) -> int32[N,1,4] or None
    (lineDetection.py:12-19, "New Version/comparison.py":14)."""
    m = image if _is_torch(image) else np.asarray(image)
    if m.ndim != 2:
        raise error("HoughLinesP: a single-channel 8-bit image is required")
    return hough_lines_p_batch(m[None], rho, theta, threshold, minLineLength, maxLineGap)[0]
