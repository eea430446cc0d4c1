"""Synthetic UML-like rasters for parity tests and benchmarks.

The reference ships ~40 sample diagrams (``images/*.png``: white page, dark
strokes 2-5 px wide, boxes with separators, elbow connectors, small text,
foreground 2-3 % of the pixels).  There is no dataset to download, so the
benchmark inputs are generated: same look, any size, reproducible from a seed.

Pure numpy on the host; nothing here is part of the measured path.
"""
from __future__ import annotations

import numpy as np

_GLYPH_BANK = None


def _glyph_bank():
    """64 fixed 14x8 glyph bitmaps (strokes of a pseudo alphabet)."""
    global _GLYPH_BANK
    if _GLYPH_BANK is None:
        rng = np.random.default_rng(12345)
        bank = np.zeros((64, 14, 8), dtype=bool)
        for g in range(64):
            for _ in range(int(rng.integers(2, 5))):
                if rng.random() < 0.5:  # vertical stroke
                    x = int(rng.integers(0, 7))
                    y0 = int(rng.integers(0, 6))
                    y1 = int(rng.integers(y0 + 4, 15))
                    bank[g, y0:y1, x : x + 2] = True
                else:  # horizontal stroke
                    y = int(rng.integers(0, 13))
                    x0 = int(rng.integers(0, 3))
                    x1 = int(rng.integers(x0 + 3, 9))
                    bank[g, y : y + 2, x0:x1] = True
        _GLYPH_BANK = bank
    return _GLYPH_BANK


class _Canvas:
    def __init__(self, h, w, rng, antialias):
        self.h, self.w = h, w
        self.img = np.full((h, w, 3), 255, np.uint8)
        self.rng = rng
        self.aa = antialias

    def _colour(self):
        v = int(self.rng.integers(0, 71))
        if self.rng.random() < 0.12:  # a few tinted strokes
            tint = self.rng.integers(0, 60, size=3)
            return np.clip(v + tint, 0, 255).astype(np.uint8)
        return np.array([v, v, v], np.uint8)

    def rect(self, y0, x0, y1, x1, col):
        """Filled rectangle [y0,y1) x [x0,x1), with a 1-px mid-gray fringe when anti-aliasing."""
        y0, x0 = max(y0, 0), max(x0, 0)
        y1, x1 = min(y1, self.h), min(x1, self.w)
        if y1 <= y0 or x1 <= x0:
            return
        if self.aa:
            fy0, fx0 = max(y0 - 1, 0), max(x0 - 1, 0)
            fy1, fx1 = min(y1 + 1, self.h), min(x1 + 1, self.w)
            fringe = (col.astype(np.int32) + 255 * 2) // 3
            reg = self.img[fy0:fy1, fx0:fx1]
            np.minimum(reg, fringe.astype(np.uint8), out=reg)
        self.img[y0:y1, x0:x1] = col

    def hline(self, y, x0, x1, t, col):
        if x1 < x0:
            x0, x1 = x1, x0
        self.rect(y, x0, y + t, x1 + t, col)

    def vline(self, x, y0, y1, t, col):
        if y1 < y0:
            y0, y1 = y1, y0
        self.rect(y0, x, y1 + t, x + t, col)

    def text(self, y, x, n_glyphs, height, col):
        bank = _glyph_bank()
        idx = self.rng.integers(0, 64, size=n_glyphs)
        for k, g in enumerate(idx):
            gx = x + k * 10
            if gx + 8 > self.w or y + height > self.h or gx < 0 or y < 0:
                break
            m = bank[g, :height]
            reg = self.img[y : y + height, gx : gx + 8]
            reg[m] = col

    def arrow_head(self, y, x, direction, size, col):
        # filled triangle pointing along +x / -x / +y / -y
        for i in range(size):
            half = i // 2
            if direction == 0:
                self.rect(y - half, x - i, y + half + 1, x - i + 1, col)
            elif direction == 1:
                self.rect(y - half, x + i, y + half + 1, x + i + 1, col)
            elif direction == 2:
                self.rect(y - i, x - half, y - i + 1, x + half + 1, col)
            else:
                self.rect(y + i, x - half, y + i + 1, x + half + 1, col)


def synth_diagram(seed: int, h: int, w: int, *, n_boxes: int | None = None, antialias: bool | None = None,
                  noise: int = 0, gradient: bool = False) -> np.ndarray:
    """One BGR uint8[h, w, 3] class/sequence-diagram-like raster.

    ``n_boxes`` defaults to 12 at 1920x1080 and grows with area^0.75.  ``noise`` adds uniform
    +-noise paper grain and ``gradient`` a smooth illumination ramp (the
    "scanned whiteboard" look).
    """
    rng = np.random.default_rng(seed)
    if antialias is None:
        antialias = bool(rng.random() < 0.5)
    if n_boxes is None:
        n_boxes = max(2, int(round(12 * ((h * w) / (1920 * 1080)) ** 0.75)))
    cv = _Canvas(h, w, rng, antialias)
    boxes = []
    scale = max(1.0, min(h, w) / 1080.0) ** 0.5
    for _ in range(n_boxes):
        bw = int(rng.integers(min(120, max(w // 4, 9)), max(min(int(380 * scale), w // 2), min(120, max(w // 4, 9)) + 1)))
        bh = int(rng.integers(min(80, max(h // 4, 9)), max(min(int(300 * scale), h // 2), min(80, max(h // 4, 9)) + 1)))
        x0 = int(rng.integers(0, max(w - bw, 1)))
        y0 = int(rng.integers(0, max(h - bh, 1)))
        t = int(rng.integers(2, 6))
        col = cv._colour()
        cv.hline(y0, x0, x0 + bw, t, col)
        cv.hline(y0 + bh, x0, x0 + bw, t, col)
        cv.vline(x0, y0, y0 + bh, t, col)
        cv.vline(x0 + bw, y0, y0 + bh, t, col)
        nsep = int(rng.integers(1, 4))
        ys = sorted(int(v) for v in rng.integers(y0 + 18, max(y0 + bh - 4, y0 + 19), size=nsep))
        for sy in ys:
            cv.hline(sy, x0, x0 + bw, max(2, t - 1), col)
        # text lines inside the box
        ty = y0 + t + 4
        while ty + 14 < y0 + bh - 2:
            gh = int(rng.integers(8, 15))
            n = int(rng.integers(3, max(4, (bw - 16) // 26)))
            cv.text(ty, x0 + t + 6, n, gh, cv._colour())
            ty += gh + int(rng.integers(10, 26))
        boxes.append((y0, x0, bh, bw))
    # connectors: elbows between box centres, some with arrow heads
    n_conn = max(1, int(n_boxes * 1.2))
    for _ in range(n_conn):
        a = boxes[int(rng.integers(0, len(boxes)))]
        b = boxes[int(rng.integers(0, len(boxes)))]
        ya, xa = a[0] + a[2] // 2, a[1] + a[3]
        yb, xb = b[0], b[1] + b[3] // 2
        t = int(rng.integers(2, 5))
        col = cv._colour()
        cv.hline(ya, xa, xb, t, col)
        cv.vline(xb, ya, yb, t, col)
        if rng.random() < 0.6:
            cv.arrow_head(yb, xb + t // 2, 3 if yb > ya else 2, int(rng.integers(8, 16)), col)
    # free-standing labels
    for _ in range(n_boxes):
        cv.text(int(rng.integers(0, max(h - 16, 1))), int(rng.integers(0, max(w - 120, 1))), int(rng.integers(3, 11)),
                int(rng.integers(8, 15)), cv._colour())
    img = cv.img
    if gradient or noise:
        f = img.astype(np.int16)
        if gradient:
            gy = np.linspace(0, -28, h, dtype=np.float32)[:, None]
            gx = np.linspace(0, -20, w, dtype=np.float32)[None, :]
            f += (gy + gx).astype(np.int16)[:, :, None]
        if noise:
            f += rng.integers(-noise, noise + 1, size=(h, w, 1), dtype=np.int16)
        img = np.clip(f, 0, 255).astype(np.uint8)
    return img


def synth_batch(seed0: int, n: int, h: int, w: int, out: np.ndarray | None = None, **kw) -> np.ndarray:
    """``n`` diagrams with seeds seed0 .. seed0+n-1 as uint8[n, h, w, 3]."""
    if out is None:
        out = np.empty((n, h, w, 3), np.uint8)
    for i in range(n):
        out[i] = synth_diagram(seed0 + i, h, w, **kw)
    return out
