"""The fused raster front-end: what the reference's parsers run before contour logic.

``front_end`` mirrors the caller wrappers ``load_image_from_file`` /
``threshold_image`` / ``extract_lines`` of the reference
(umlClassdiagram.py:12-30, umlSequencediagram.py:15-33,
umlActivitydiagram.py:22-44, umlStateDiagram.py:12-30) followed by the
line-detection prefix ``cvtColor -> [GaussianBlur] -> Canny -> HoughLines``
(lineDetection.py:6-9, "New Version/comparison.py":30), with the same defaults
(``vert_len = horiz_len = 15``, ``iterations = 2``).  The batched form runs the
whole batch through one C call (`ipa_front_end_u8`): per-image Python overhead
would otherwise dominate.
"""
from __future__ import annotations

import dataclasses
from typing import List, Optional

import numpy as np
import torch

from . import _native, cv2api
from ._device import DevImage, check, error, stream_ptr, upload, workspace


def set_option(name: str, value: int) -> None:
    """Schedule switch of the fused pass for the calls that follow ("pipeline_chunks", "overlap",
    "phased", "head_start_ticks"; see include/ipamd.h).  Results never depend on them."""
    check(_native.lib().ipa_set_option(name.encode(), int(value)))


def get_option(name: str) -> int:
    import ctypes as C

    v = C.c_int()
    check(_native.lib().ipa_get_option(name.encode(), C.byref(v)))
    return v.value


@dataclasses.dataclass
class FrontEndResult:
    gray: object
    thresh: object
    vlines: object
    hlines: object
    edges: object
    accum: Optional[object] = None         # int32[batch, numangle+2, numrho+2] (device) when requested
    lines: Optional[List[object]] = None   # per image: float32[N,1,2] or None


class FrontEnd:
    """Pre-planned front-end for a fixed (batch, height, width): owns the five output
    planes, the Hough accumulators and the workspace, so that ``run`` allocates nothing."""

    def __init__(self, batch: int, height: int, width: int, *, thresh_val: int = 128, vert_len: int = 15,
                 horiz_len: int = 15, iterations: int = 2, canny=(50, 150), blur: bool = False, hough: bool = True,
                 max_points: int = 0, edges: bool = True, line_bits: bool = False):
        L = _native.lib()
        self.batch, self.height, self.width = int(batch), int(height), int(width)
        self.thresh_val, self.vert_len, self.horiz_len = int(thresh_val), int(vert_len), int(horiz_len)
        self.iterations, self.canny, self.blur, self.hough = int(iterations), canny, bool(blur), bool(hough)
        self.edges = bool(edges) or self.hough  # False: no Canny, the edges plane stays unwritten
        if self.vert_len < 1 or self.horiz_len < 1 or self.iterations < 1:
            raise error("front_end: vert_len, horiz_len and iterations must be >= 1")
        if max_points <= 0:
            # edge maps of diagrams are sparse: keep room for 1/4 of the pixels, retry on overflow
            max_points = max(1024, (self.height * self.width) // 4)
        self.max_points = int(min(max_points, self.height * self.width)) if self.height * self.width else 0
        self.planes = DevImage.empty(5 * max(self.batch, 1), self.height, self.width, 1)
        self.na, self.nr = cv2api.hough_dims(self.width, self.height, 1.0, np.pi / 180)
        dev = self.planes.t.device
        if self.hough:
            self.accum = torch.empty((max(self.batch, 1), self.na + 2, self.nr + 2), dtype=torch.int32, device=dev)
            self.point_counts = torch.zeros(max(self.batch, 1), dtype=torch.int32, device=dev)
            self.seg_max = torch.zeros((max(self.batch, 1), self.na, L.ipa_hough_segments(self.nr)), dtype=torch.int32,
                                       device=dev)
        else:
            self.accum = None
            self.point_counts = None
            self.seg_max = None
        # line_bits: the two line masks also as 1-bit planes, the form get_segments works on (line_segments())
        self.bits_pitch = int(L.ipa_bits_pitch_words(self.width))
        self.line_bits = (torch.empty((2, max(self.batch, 1), max(self.height, 1), self.bits_pitch), dtype=torch.int32, device=dev)
                          if line_bits else None)
        self._unchecked = False
        self._alloc_ws()

    def _alloc_ws(self):
        L = _native.lib()
        self.ws = workspace(L.ipa_front_end_workspace_bytes(self.width, self.height, self.batch,
                                                            self.max_points if self.hough else 1))

    def plane(self, k: int) -> DevImage:
        """k: 0 gray, 1 thresh, 2 vlines, 3 hlines, 4 edges."""
        b = max(self.batch, 1)
        return DevImage(self.planes.t[k * b : (k + 1) * b], self.batch, self.height, self.width, 1)

    def enqueue(self, bgr: DevImage) -> None:
        """One pass over a device-resident batch.  Synchronises the stream while the Canny
        hysteresis converges; Hough point-list overflow is checked by ``run``."""
        L = _native.lib()
        if (bgr.batch, bgr.height, bgr.width, bgr.channels) != (self.batch, self.height, self.width, 3):
            raise error("front_end: input batch does not match the planned shape")
        p = [self.plane(k) for k in range(5)]
        args = (
            bgr.ptr, bgr.pitch, bgr.stride, p[0].ptr, p[1].ptr, p[2].ptr, p[3].ptr, p[4].ptr if self.edges else None,
            p[0].pitch, p[0].stride,
            self.accum.data_ptr() if self.hough else None, self.accum.stride(0) if self.hough else 0,
            self.seg_max.data_ptr() if self.hough else None,
            self.max_points if self.hough else 1, self.point_counts.data_ptr() if self.hough else None,
            self.width, self.height, self.batch, self.thresh_val, self.vert_len, self.horiz_len, self.iterations,
            float(self.canny[0]), float(self.canny[1]), 1 if self.blur else 0,
            self.ws.data_ptr(), self.ws.numel(), stream_ptr())
        if self.line_bits is not None:
            check(L.ipa_front_end_bits_u8(*args, self.line_bits[0].data_ptr(), self.line_bits[1].data_ptr()))
        else:
            check(L.ipa_front_end_u8(*args))
        self._unchecked = self.hough and self.max_points < self.height * self.width

    def line_segments(self, which: int, max_boxes: int = 4096):
        """get_segments (umlClassdiagram.py:33-36) of the vertical (which = 0) or horizontal (1) line masks of
        the last pass, from their 1-bit planes (FrontEnd(..., line_bits=True)): per image a list of (x, y, w, h)."""
        if self.line_bits is None:
            raise error("front_end: plan the FrontEnd with line_bits=True")
        return _segments_run(self.batch, self.height, self.width, max_boxes, bits=self.line_bits[which], bits_pitch=self.bits_pitch)

    def check_overflow(self) -> bool:
        """True when an image of the last ``enqueue`` had more edge pixels than ``max_points``: its
        accumulator is incomplete (``run`` re-plans and repeats the pass by itself).  Reads the counts
        back, i.e. waits for the pass."""
        over = False
        if self.hough and self.batch and self.max_points < self.height * self.width:
            over = int(self.point_counts[: self.batch].max().item()) > self.max_points
        self._unchecked = False
        return over

    def run(self, bgr: DevImage) -> None:
        """``enqueue`` plus the overflow check of the Hough point lists (re-plans and re-runs
        once with full-size lists if an image had more edge pixels than planned for)."""
        self.enqueue(bgr)
        if self.check_overflow():
            self.max_points = self.height * self.width
            self._alloc_ws()
            self.enqueue(bgr)
            self._unchecked = False

    def lines(self, threshold: int):
        """Per-image cv2.HoughLines output from the accumulators of the last run."""
        if self._unchecked and self.check_overflow():
            raise error("front_end: an image has more edge pixels than max_points, its accumulator is incomplete "
                        "(use run(), or plan the FrontEnd with a larger max_points)")
        allkeys = cv2api.hough_peaks_batch(self.accum[: self.batch], int(threshold), seg_max=self.seg_max[: self.batch])
        return [cv2api.lines_from_keys(k, self.nr, 1.0, np.pi / 180) if k.size else None for k in allkeys]


def _result(fe: FrontEnd, like, squeeze: bool, hough_threshold) -> FrontEndResult:
    to = (lambda d: d.to_tensor(squeeze)) if isinstance(like, torch.Tensor) else (lambda d: d.to_numpy(squeeze))
    planes = [to(fe.plane(k)) if (k < 4 or fe.edges) else None for k in range(5)]
    acc = None
    if fe.hough:
        acc = fe.accum[0] if squeeze else fe.accum[: fe.batch]
    lines = fe.lines(hough_threshold) if (fe.hough and hough_threshold is not None) else None
    if lines is not None and squeeze:
        lines = lines[0]
    return FrontEndResult(*planes, accum=acc, lines=lines)


def front_end(img, thresh_val=128, vert_len=15, horiz_len=15, iterations=2, canny=(50, 150), blur=False,
              hough_threshold=None, hough=True) -> FrontEndResult:
    """Full front-end of ONE BGR image (numpy uint8[H,W,3] or ROCm tensor)."""
    s = upload(img, 3)
    fe = FrontEnd(1, s.height, s.width, thresh_val=thresh_val, vert_len=vert_len, horiz_len=horiz_len,
                  iterations=iterations, canny=canny, blur=blur, hough=hough)
    fe.run(s)
    return _result(fe, img, True, hough_threshold)


def front_end_batch(images, thresh_val=128, vert_len=15, horiz_len=15, iterations=2, canny=(50, 150), blur=False,
                    hough_threshold=None, hough=True, edges=True) -> FrontEndResult:
    """Full front-end of a batch uint8[N,H,W,3] in one device call.  ``edges=False`` (with ``hough=False``)
    leaves Canny out: the parsers only use gray, thresh and the two line masks; ``edges`` is None then."""
    s = upload(images, 3, batched=True)
    fe = FrontEnd(s.batch, s.height, s.width, thresh_val=thresh_val, vert_len=vert_len, horiz_len=horiz_len,
                  iterations=iterations, canny=canny, blur=blur, hough=hough, edges=edges)
    fe.run(s)
    return _result(fe, images, False, hough_threshold)


def edges_and_lines(img, lo=50, hi=150, hough_threshold=100, blur=False):
    """The prefix of lineDetection.py:6-9 with the standard accumulator instead of HoughLinesP:
    gray -> [GaussianBlur 5x5] -> Canny(lo, hi) -> HoughLines(edges, 1, pi/180, hough_threshold)."""
    r = front_end(img, canny=(lo, hi), blur=blur, hough_threshold=hough_threshold)
    return r.gray, r.edges, r.lines


def erase_rects(bgr: DevImage, rects, bg_color=(255, 255, 255)) -> None:
    """``image_rect[y1:y2, x1:x2] = bg_color`` of draw_rectangles (umlClassdiagram.py:107-116) for a
    device-resident batch, in place.  ``rects``: rows of (image index, x1, y1, x2, y2) with numpy
    slice semantics; a (n, 4) array addresses image 0."""
    L = _native.lib()
    r = np.asarray(rects, dtype=np.int64)
    if r.size == 0:
        return
    if r.ndim == 1:
        r = r[None, :]
    if r.shape[1] == 4:
        r = np.concatenate([np.zeros((r.shape[0], 1), np.int64), r], axis=1)
    if r.shape[1] != 5:
        raise error("erase_rects: rects must have 4 or 5 columns")
    if bgr.channels != 3:
        raise error("erase_rects: a 3-channel image is required")
    if np.abs(r).max() > 2 ** 31 - 1:
        raise error("erase_rects: coordinate out of range")
    dev = torch.from_numpy(np.ascontiguousarray(r.astype(np.int32))).to(bgr.t.device)
    c = [int(v) for v in bg_color]
    if len(c) != 3 or any(v < 0 or v > 255 for v in c):
        raise error("erase_rects: bg_color must be three values in 0..255")
    check(L.ipa_fill_rects_c3_u8(bgr.ptr, bgr.pitch, bgr.stride, bgr.width, bgr.height, bgr.batch, dev.data_ptr(),
                                 int(r.shape[0]), c[0], c[1], c[2], stream_ptr()))


def rebinarize(img, rects, thresh_val=128, bg_color=(255, 255, 255), maxval=255, thresh_type=cv2api.THRESH_BINARY_INV):
    """The second binarisation of the parsers (umlClassdiagram.py:1117-1121): erase the rectangle
    line areas from the BGR image, then gray + threshold, without leaving the device.

    ``img``: numpy uint8[H,W,3] / [N,H,W,3] or ROCm tensor (a tensor is modified in place, like the
    reference's ``image_rect``; numpy input is copied).  Returns (erased BGR, gray, thresh) of the
    same kind as ``img``."""
    L = _native.lib()
    batched = getattr(img, "ndim", 3) == 4
    s = upload(img, 3, batched=batched)
    erase_rects(s, rects, bg_color)
    planes = DevImage.empty(2 * s.batch, s.height, s.width, 1)
    g = DevImage(planes.t[: s.batch], s.batch, s.height, s.width, 1)
    t = DevImage(planes.t[s.batch :], s.batch, s.height, s.width, 1)
    check(L.ipa_gray_threshold_u8(s.ptr, s.pitch, s.stride, g.ptr, t.ptr, g.pitch, g.stride, s.width, s.height, s.batch,
                                  float(thresh_val), float(maxval), int(thresh_type), stream_ptr()))
    to = (lambda d: d.to_tensor(not batched)) if isinstance(img, torch.Tensor) else (lambda d: d.to_numpy(not batched))
    return to(s), to(g), to(t)


def _segments_run(n, height, width, max_boxes, mask: DevImage = None, bits=None, bits_pitch=0):
    """The segment pass on byte masks (``mask``) or on 1-bit planes int32[n, height, bits_pitch] (``bits``)."""
    L = _native.lib()
    if n == 0:
        return []
    dev = mask.t.device if mask is not None else bits.device
    max_segments = 0
    while True:
        boxes = torch.empty((n, max(max_boxes, 1), 4), dtype=torch.int32, device=dev)
        counts = torch.zeros(n, dtype=torch.int32, device=dev)
        ws = workspace(L.ipa_segments_workspace_bytes(width, height, n, max_segments))
        if mask is not None:
            check(L.ipa_segments_u8(mask.ptr, mask.pitch, mask.stride, width, height, n, max_segments, boxes.data_ptr(),
                                    max_boxes, counts.data_ptr(), ws.data_ptr(), ws.numel(), stream_ptr()))
        else:
            check(L.ipa_segments_bits_u32(bits.data_ptr(), bits_pitch, bits_pitch * height, width, height, n, max_segments,
                                          boxes.data_ptr(), max_boxes, counts.data_ptr(), ws.data_ptr(), ws.numel(),
                                          stream_ptr()))
        c = counts.cpu().numpy()
        if (c < 0).any():  # more runs than planned for: allow every pixel to be its own run
            max_segments = width * height
            continue
        if c.max(initial=0) > max_boxes:
            max_boxes = int(c.max())
            continue
        b = boxes.cpu().numpy()
        return [[tuple(int(v) for v in b[i, k]) for k in range(int(c[i]))] for i in range(n)]


def segments_batch(masks, max_boxes=4096):
    """get_segments of the parsers for a batch of masks uint8[N,H,W] (numpy or ROCm tensor):
    per image the list of (x, y, w, h) of cv2.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)
    + cv2.boundingRect, in cv2's order (umlClassdiagram.py:33-36)."""
    s = upload(masks, 1, batched=True)
    return _segments_run(s.batch, s.height, s.width, max_boxes, mask=s)


def get_segments(lines, max_boxes=4096):
    """get_segments(lines) of the parsers (umlClassdiagram.py:33-36) for one mask: list of (x, y, w, h)."""
    m = lines if isinstance(lines, torch.Tensor) else np.asarray(lines)
    if m.ndim != 2:
        raise error("get_segments: a single-channel 2-D mask is required")
    return segments_batch(m[None], max_boxes)[0]
