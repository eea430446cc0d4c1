"""Randomised fused-front-end cases against the CPU restatement (test infrastructure, not collected by
pytest): python tests/fuzz_gpu.py [cases] [first seed] for the fused pass, python tests/fuzz_gpu.py ops
[cases] [first seed] for every single operation, python tests/fuzz_gpu.py batch [cases] [first seed] for
the batched helpers, strided / device inputs, rectangle erase and large shapes, python tests/fuzz_gpu.py abi
[cases] [first seed] for the C entry points with minimal (16-byte) pitches, strides and base alignment,
python tests/fuzz_gpu.py threads [cases] [first seed] for concurrent calls from four host threads.  Shapes are biased towards the tile, strip and
word boundaries of the kernels (multiples of 32 / 64 / 128 / 248 / 256 +-1), batches towards the sizes
at which the fused pass switches schedules (1, 63, 64, 65, 128+), parameters over the ranges the
parsers use and beyond.  Prints one line per failing case and a summary; exit status 1 on any failure."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import imageprocessing_amd as ipa  # noqa: E402
from oracle import frontref as fr  # noqa: E402

EDGES = [1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 29, 31, 32, 33, 63, 64, 65, 127, 128, 129, 247, 248, 249, 255, 256, 257,
         495, 496, 497, 511, 512, 513, 743, 744, 745, 1023, 1024, 1025]


def pick_dim(rng, hi):
    if rng.random() < 0.6:
        v = int(rng.choice(EDGES)) + int(rng.integers(-1, 2)) * int(rng.random() < 0.3)
    else:
        v = int(rng.integers(1, hi))
    return max(1, min(v, hi))


def diagram(rng, n, h, w):
    """White pages with dark strokes, boxes, speckle and, sometimes, a gradient or noise."""
    img = np.full((n, h, w, 3), 255, np.uint8)
    for b in range(n):
        kind = rng.integers(0, 4)
        if kind == 3:
            img[b] = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            continue
        for _ in range(int(rng.integers(0, 12))):
            x0, x1 = sorted(int(v) for v in rng.integers(0, w + 1, 2))
            y0, y1 = sorted(int(v) for v in rng.integers(0, h + 1, 2))
            t = int(rng.integers(1, 6))
            col = rng.integers(0, 120, 3).astype(np.uint8)
            img[b, y0:y0 + t, x0:x1] = col
            img[b, max(y1 - t, 0):y1, x0:x1] = col
            img[b, y0:y1, x0:x0 + t] = col
            img[b, y0:y1, max(x1 - t, 0):x1] = col
        if kind >= 1:
            m = rng.random((h, w)) < 0.02
            img[b][m] = rng.integers(0, 256, (int(m.sum()), 3), dtype=np.uint8)
        if kind == 2:
            ramp = (np.arange(w)[None, :] * 90 // max(w, 1)).astype(np.int16)
            img[b] = np.clip(img[b].astype(np.int16) - ramp[:, :, None], 0, 255).astype(np.uint8)
    return img


def oracle_case(bgr, t, vl, hl, it, lo, hi, blur, hough, edges):
    g = fr.bgr2gray(bgr)
    _, th = fr.threshold(g, t, 255, 1)
    v = fr.morphology_ex(th, 2, (1, vl), it)
    hz = fr.morphology_ex(th, 2, (hl, 1), it)
    e = acc = None
    if edges:
        src = fr.gaussian_blur(g, (5, 5), 0) if blur else g
        e = fr.canny(src, lo, hi)
        if hough:
            acc = fr.hough_accum(e, 1.0, np.pi / 180)
    return g, th, v, hz, e, acc


def gray_image(rng, kind, h, w):
    if kind == 0:
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind == 1:
        return ((rng.random((h, w)) < rng.uniform(0.02, 0.6)) * 255).astype(np.uint8)
    if kind == 2:
        return fr.bgr2gray(diagram(rng, 1, h, w)[0])
    from scipy import ndimage

    return ndimage.gaussian_filter(rng.integers(0, 256, (h, w)).astype(np.float32), rng.uniform(0.8, 3.0)).astype(np.uint8)


def ops_case(rng):
    """Every single operation of the cv2-style module on one random image; returns the name of the
    first operation whose result differs from the restatement, or None."""
    cv2 = ipa
    w, h = pick_dim(rng, 1100), pick_dim(rng, 700)
    if rng.random() < 0.04:  # ribbons: one side up to the 65535 limit
        long, short = int(rng.choice([4097, 16384, 32769, 65535, int(rng.integers(5000, 65536))])), int(rng.integers(1, 9))
        w, h = (long, short) if rng.random() < 0.5 else (short, long)
    g = gray_image(rng, int(rng.integers(0, 4)), h, w)
    binary = gray_image(rng, int(rng.choice([1, 2])), h, w)
    if binary.max() > 0 and len(np.unique(binary)) > 2:
        binary = fr.threshold(binary, 128, 255, 1)[1]
    bgr = diagram(rng, 1, h, w)[0]
    if not np.array_equal(cv2.cvtColor(bgr, cv2.COLOR_BGR2GRAY), fr.bgr2gray(bgr)):
        return f"cvtColor {w}x{h}"
    t, mv, typ = float(rng.uniform(-20, 280)), float(rng.uniform(0, 300)), int(rng.integers(0, 2))
    if rng.random() < 0.3:
        typ |= cv2.THRESH_OTSU
    rt, d = cv2.threshold(g, t, mv, typ)
    ro, do = fr.threshold(g, t, mv, typ)
    if rt != ro or not np.array_equal(d, do):
        return f"threshold {w}x{h} {t} {mv} {typ}"
    km = int(rng.choice([3, 5]))
    if not np.array_equal(cv2.medianBlur(g, km), fr.median_blur(g, km)):
        return f"medianBlur {w}x{h} {km}"
    for src in (g, binary):
        kw_, kh_ = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        r = rng.random()
        if r < 0.6:
            kw_, kh_ = (kw_, 1) if rng.random() < 0.5 else (1, kh_)
        elif r < 0.8:
            kw_, kh_ = int(rng.integers(1, 8)), int(rng.integers(1, 8))
        if rng.random() < 0.15:  # long lines (the whiteboard configuration uses 101 and 201)
            long = int(rng.choice([95, 96, 97, 101, 127, 128, 129, 201, 255, 256, 257, 300]))
            kw_, kh_ = (long, 1) if rng.random() < 0.5 else (1, long)
        it, op = int(rng.integers(1, 4)), int(rng.integers(0, 4))
        anchor = (-1, -1) if rng.random() < 0.7 else (int(rng.integers(0, kw_)), int(rng.integers(0, kh_)))
        got = cv2.morphologyEx(src, op, np.ones((kh_, kw_), np.uint8), anchor=anchor, iterations=it)
        if not np.array_equal(got, fr.morphology_ex(src, op, (kw_, kh_), it, anchor)):
            return f"morphologyEx {w}x{h} op={op} k=({kw_},{kh_}) it={it} anchor={anchor} binary={src is binary}"
    kx, ky = 2 * int(rng.integers(0, 16)) + 1, 2 * int(rng.integers(0, 16)) + 1
    sg = 0.0 if rng.random() < 0.5 else float(rng.uniform(0.3, 6.0))
    sgy = 0.0 if rng.random() < 0.7 else float(rng.uniform(0.3, 6.0))
    border = int(rng.choice([4, 4, 1, 2, 0]))
    if not np.array_equal(cv2.GaussianBlur(g, (kx, ky), sg, sigmaY=sgy, borderType=border), fr.gaussian_blur(g, (kx, ky), sg, sgy, border)):
        return f"GaussianBlur {w}x{h} ({kx},{ky}) {sg} {sgy} border={border}"
    blk, cc = 2 * int(rng.integers(1, 21)) + 1, float(rng.uniform(-10, 10))
    meth, typ, mv = int(rng.integers(0, 2)), int(rng.integers(0, 2)), float(rng.choice([255, 255, 1, 77.6, 300]))
    if not np.array_equal(cv2.adaptiveThreshold(g, mv, meth, typ, blk, cc), fr.adaptive_threshold(g, mv, meth, typ, blk, cc)):
        return f"adaptiveThreshold {w}x{h} {mv} {meth} {typ} {blk} {cc}"
    lo, hi = float(rng.uniform(0, 300)), float(rng.uniform(0, 300))
    bands, queue = int(rng.choice([-1, 16, 64, 64])), int(rng.random() < 0.8)  # band height / deferred suppression of the strip kernel
    cv2.set_option("canny_bands", bands)
    cv2.set_option("canny_queue", queue)
    try:
        e = cv2.Canny(g, lo, hi)
    finally:
        cv2.set_option("canny_bands", -1)
        cv2.set_option("canny_queue", 1)
    if not np.array_equal(e, fr.canny(g, lo, hi)):
        return f"Canny {w}x{h} {lo} {hi} bands={bands} queue={queue}"
    rho, theta = float(rng.choice([0.5, 1.0, 2.0, 3.0])), float(rng.choice([np.pi / 360, np.pi / 180, np.pi / 90, 0.05]))
    src = e if np.count_nonzero(e) else binary
    if not np.array_equal(cv2.hough_accumulator(src, rho, theta).cpu().numpy(), fr.hough_accum(src, rho, theta)):
        return f"hough_accumulator {w}x{h} {rho} {theta}"
    thr = int(rng.integers(1, 120))
    a, b = cv2.HoughLines(src, rho, theta, thr), fr.hough_lines(src, rho, theta, thr)
    if (a is None) != (b is None) or (a is not None and not np.array_equal(a, b)):
        return f"HoughLines {w}x{h} {rho} {theta} {thr}"
    if np.count_nonzero(src) < 60000:  # the restatement visits pixel by pixel
        thr, mll, gap = int(rng.integers(5, 120)), int(rng.integers(0, 60)), int(rng.integers(0, 12))
        a, b = cv2.HoughLinesP(src, rho, theta, thr, minLineLength=mll, maxLineGap=gap), fr.hough_lines_p(src, rho, theta, thr, mll, gap)
        a = np.zeros((0, 4), np.int32) if a is None else np.asarray(a).reshape(-1, 4)
        b = np.zeros((0, 4), np.int32) if b is None else np.asarray(b).reshape(-1, 4)
        if not np.array_equal(a, b):
            return f"HoughLinesP {w}x{h} {rho} {theta} {thr} {mll} {gap}"
    got = np.asarray(cv2.get_segments(binary)).reshape(-1, 4)
    exp = np.asarray(fr.external_boxes(binary)).reshape(-1, 4)
    if not np.array_equal(got, exp):
        return f"get_segments {w}x{h}"
    return None


def batch_case(rng):
    """Batched helpers, strided and device-tensor inputs, rectangle erase + re-binarisation, and now and
    then a large shape; returns a description of the first difference or None."""
    import torch

    cv2 = ipa
    big = rng.random() < 0.04
    w, h = (pick_dim(rng, 5000), pick_dim(rng, 3000)) if big else (pick_dim(rng, 600), pick_dim(rng, 400))
    n = 1 if big else int(rng.choice([1, 2, 3, 7, 33, 70]))
    while n > 1 and n * w * h > 6_000_000:
        n = max(1, n // 2)
    masks = np.stack([gray_image(rng, int(rng.choice([1, 2])), h, w) for _ in range(n)])
    masks = np.where(masks > 127, 0, 255).astype(np.uint8) if rng.random() < 0.5 else ((masks > 0) * 255).astype(np.uint8)
    # segments of a batch, from numpy and from a device tensor
    src = masks if rng.random() < 0.5 else torch.from_numpy(masks).cuda()
    got = ipa.segments_batch(src)
    for b in range(n):
        exp = fr.external_boxes(masks[b])
        if not np.array_equal(np.asarray(got[b], dtype=np.int64).reshape(-1, 4), exp):
            return f"segments_batch n={n} {w}x{h} image {b}"
    # strided view as input (cv2 accepts any strides)
    wide = np.zeros((h, w + 7), np.uint8)
    wide[:, 3:3 + w] = masks[0]
    view = wide[:, 3:3 + w]
    k = int(rng.integers(1, 40))
    if not np.array_equal(cv2.erode(view, np.ones((1, k), np.uint8)), fr.morphology_ex(masks[0], 0, (k, 1), 1)):
        return f"erode on a strided view {w}x{h} k={k}"
    rev = masks[0][::-1]
    if not np.array_equal(cv2.dilate(rev, np.ones((k, 1), np.uint8)), fr.morphology_ex(np.ascontiguousarray(rev), 1, (1, k), 1)):
        return f"dilate on a flipped view {w}x{h} k={k}"
    # edge maps: batched HoughLinesP and batched peaks
    e = np.stack([fr.canny(gray_image(rng, 2, h, w), 50, 150) for _ in range(min(n, 4))])
    if np.count_nonzero(e) < 40000 * len(e):
        thr, mll, gap = int(rng.integers(10, 100)), int(rng.integers(0, 50)), int(rng.integers(0, 10))
        got = ipa.hough_lines_p_batch(e, 1.0, np.pi / 180, thr, minLineLength=mll, maxLineGap=gap)
        for b in range(len(e)):
            exp = fr.hough_lines_p(e[b], 1.0, np.pi / 180, thr, mll, gap)
            a = np.zeros((0, 4), np.int32) if got[b] is None else np.asarray(got[b]).reshape(-1, 4)
            x = np.zeros((0, 4), np.int32) if exp is None else exp.reshape(-1, 4)
            if not np.array_equal(a, x):
                return f"hough_lines_p_batch {w}x{h} image {b} thr={thr} len={mll} gap={gap}"
    # rectangle erase + second binarisation
    bgr = diagram(rng, n, h, w)
    rects = []
    for _ in range(int(rng.integers(0, 9))):
        x1, x2 = sorted(int(v) for v in rng.integers(-20, w + 20, 2))
        y1, y2 = sorted(int(v) for v in rng.integers(-20, h + 20, 2))
        rects.append((int(rng.integers(0, n)), max(x1, 0), max(y1, 0), max(x2, 0), max(y2, 0)))
    t = int(rng.integers(0, 256))
    col = tuple(int(v) for v in rng.integers(0, 256, 3))
    eb, g, th = ipa.rebinarize(bgr, rects, thresh_val=t, bg_color=col)
    ref = bgr.copy()
    for (b, x1, y1, x2, y2) in rects:
        ref[b, y1:y2, x1:x2] = col
    if not np.array_equal(eb, ref):
        return f"rebinarize (erase) n={n} {w}x{h}"
    for b in sorted(set([0, n - 1])):
        gg = fr.bgr2gray(ref[b])
        if not np.array_equal(g[b], gg) or not np.array_equal(th[b], fr.threshold(gg, t, 255, 1)[1]):
            return f"rebinarize (gray/threshold) n={n} {w}x{h} image {b}"
    # one large or odd-shaped full front-end with the line list
    r = ipa.front_end(bgr[0], thresh_val=t, hough_threshold=int(rng.integers(20, 300)))
    exp = oracle_case(bgr[0], t, 15, 15, 2, 50, 150, False, True, True)
    for name, a, x in zip(("gray", "thresh", "vlines", "hlines", "edges"), (r.gray, r.thresh, r.vlines, r.hlines, r.edges), exp):
        if not np.array_equal(a, x):
            return f"front_end {w}x{h} plane {name}"
    if not np.array_equal(r.accum.cpu().numpy(), exp[5]):
        return f"front_end {w}x{h} accum"
    return None


class Dev:
    """A batch of rows in device memory with a caller-chosen layout: base pointer 16-byte aligned but
    not more, pitch and stride any multiples of 16, the gaps filled with a marker."""

    def __init__(self, rng, n, h, rb, data=None):
        import torch

        self.n, self.h, self.rb = n, h, rb
        self.pitch = ((rb + 15) // 16 + int(rng.integers(0, 4)) * int(rng.random() < 0.6)) * 16
        self.pitch = max(self.pitch, 16)
        self.stride = self.pitch * h + 16 * int(rng.integers(0, 5)) * int(rng.random() < 0.5)
        self.off = 16 * int(rng.integers(0, 16))
        self.t = torch.full((self.off + max(n, 1) * self.stride + 4096,), 0xAB, dtype=torch.uint8, device="cuda")
        if data is not None and n * h * rb:
            self.view().copy_(torch.from_numpy(np.ascontiguousarray(data).reshape(n, h, rb)).cuda())

    def view(self):
        import torch

        return torch.as_strided(self.t, (self.n, self.h, self.rb), (self.stride, self.pitch, 1), self.off)

    @property
    def ptr(self):
        return self.t.data_ptr() + self.off

    def get(self):
        return self.view().cpu().numpy()


def abi_case(rng):
    """The C entry points called directly with pitches, strides and base pointers that only meet the
    documented 16-byte rule."""
    import torch

    from imageprocessing_amd import _native

    L = _native.lib()
    w, h = pick_dim(rng, 700), pick_dim(rng, 400)
    n = int(rng.choice([1, 1, 2, 3, 5, 66]))
    while n > 1 and n * w * h > 4_000_000:
        n = max(1, n // 2)
    sp = torch.cuda.current_stream().cuda_stream

    def ws_for(nbytes):
        return torch.empty(int(nbytes) + 256, dtype=torch.uint8, device="cuda")

    def ok(rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: rc={rc} {_native.last_error()}")

    bgr = diagram(rng, n, h, w)
    gray = np.stack([fr.bgr2gray(bgr[b]) for b in range(n)])
    S = Dev(rng, n, h, 3 * w, bgr)
    # bgr2gray
    G = Dev(rng, n, h, w)
    ok(L.ipa_bgr2gray_u8(S.ptr, S.pitch, S.stride, G.ptr, G.pitch, G.stride, w, h, n, sp), "bgr2gray")
    if not np.array_equal(G.get(), gray):
        return f"ipa_bgr2gray_u8 n={n} {w}x{h} pitches {S.pitch}/{G.pitch}"
    # gray + threshold in one pass
    t = float(rng.integers(0, 256))
    G2, T2 = Dev(rng, n, h, w), None
    T2 = Dev(rng, n, h, w)
    T2.pitch, T2.stride = G2.pitch, G2.stride
    T2.t = torch.full_like(G2.t, 0xAB)
    T2.off = G2.off
    ok(L.ipa_gray_threshold_u8(S.ptr, S.pitch, S.stride, G2.ptr, T2.ptr, G2.pitch, G2.stride, w, h, n, t, 255.0, 1, sp), "gray_threshold")
    th = np.stack([fr.threshold(gray[b], t, 255, 1)[1] for b in range(n)])
    if not np.array_equal(G2.get(), gray) or not np.array_equal(T2.get(), th):
        return f"ipa_gray_threshold_u8 n={n} {w}x{h}"
    # threshold
    D = Dev(rng, n, h, w)
    typ = int(rng.integers(0, 2))
    ok(L.ipa_threshold_u8(G.ptr, G.pitch, G.stride, D.ptr, D.pitch, D.stride, w, h, n, t, 200.0, typ, sp), "threshold")
    if not np.array_equal(D.get(), np.stack([fr.threshold(gray[b], t, 200, typ)[1] for b in range(n)])):
        return f"ipa_threshold_u8 n={n} {w}x{h}"
    # morphology, binary and 8-bit
    TH = Dev(rng, n, h, w, th)
    for src, arr, binary in ((TH, th, 1), (G, gray, 0)):
        kw_, kh_ = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        if rng.random() < 0.6:
            kw_, kh_ = (kw_, 1) if rng.random() < 0.5 else (1, kh_)
        it, op = int(rng.integers(1, 3)), int(rng.integers(0, 4))
        ws = ws_for(L.ipa_morph_workspace_bytes(w, h, n))
        D = Dev(rng, n, h, w)
        ok(L.ipa_morph_u8(src.ptr, src.pitch, src.stride, D.ptr, D.pitch, D.stride, w, h, n, op, kw_, kh_, -1, -1, it, binary,
                          ws.data_ptr(), ws.numel(), sp), "morph")
        if not np.array_equal(D.get(), np.stack([fr.morphology_ex(arr[b], op, (kw_, kh_), it) for b in range(n)])):
            return f"ipa_morph_u8 n={n} {w}x{h} op={op} k=({kw_},{kh_}) it={it} binary={binary} pitches {src.pitch}/{D.pitch}"
    # two-valued and gray images in one batch, content not promised (the one-pass two-valued kernel gives up per image)
    if n >= 2:
        mixed = np.where((np.arange(n) % 2 == 0)[:, None, None], th, gray)
        MX = Dev(rng, n, h, w, mixed)
        kw_, kh_ = int(rng.integers(1, 34)), int(rng.integers(1, 20))
        it, op = int(rng.integers(1, 3)), int(rng.integers(0, 4))
        ws = ws_for(L.ipa_morph_workspace_bytes(w, h, n))
        D = Dev(rng, n, h, w)
        ok(L.ipa_morph_u8(MX.ptr, MX.pitch, MX.stride, D.ptr, D.pitch, D.stride, w, h, n, op, kw_, kh_, -1, -1, it, 0,
                          ws.data_ptr(), ws.numel(), sp), "morph mixed")
        if not np.array_equal(D.get(), np.stack([fr.morphology_ex(mixed[b], op, (kw_, kh_), it) for b in range(n)])):
            return f"ipa_morph_u8 mixed batch n={n} {w}x{h} op={op} k=({kw_},{kh_}) it={it}"
    # medianBlur
    km = int(rng.choice([3, 5]))
    D = Dev(rng, n, h, w)
    ok(L.ipa_median_blur_u8(G.ptr, G.pitch, G.stride, D.ptr, D.pitch, D.stride, w, h, n, km, sp), "median")
    if not np.array_equal(D.get(), np.stack([fr.median_blur(gray[b], km) for b in range(n)])):
        return f"ipa_median_blur_u8 n={n} {w}x{h} {km}"
    # GaussianBlur
    kx, ky = 2 * int(rng.integers(0, 6)) + 1, 2 * int(rng.integers(0, 6)) + 1
    ws = ws_for(L.ipa_gaussian_blur_workspace_bytes(w, h, n))
    D = Dev(rng, n, h, w)
    ok(L.ipa_gaussian_blur_u8(G.ptr, G.pitch, G.stride, D.ptr, D.pitch, D.stride, w, h, n, kx, ky, 0.0, 0.0, 4, ws.data_ptr(), ws.numel(), sp), "blur")
    if not np.array_equal(D.get(), np.stack([fr.gaussian_blur(gray[b], (kx, ky), 0) for b in range(n)])):
        return f"ipa_gaussian_blur_u8 n={n} {w}x{h} ({kx},{ky})"
    # adaptive threshold
    blk = 2 * int(rng.integers(1, 16)) + 1
    ws = ws_for(L.ipa_adaptive_threshold_workspace_bytes(w, h, n))
    D = Dev(rng, n, h, w)
    meth = int(rng.integers(0, 2))
    ok(L.ipa_adaptive_threshold_u8(G.ptr, G.pitch, G.stride, D.ptr, D.pitch, D.stride, w, h, n, 255.0, meth, 0, blk, 2.0, ws.data_ptr(), ws.numel(), sp), "adaptive")
    if not np.array_equal(D.get(), np.stack([fr.adaptive_threshold(gray[b], 255, meth, 0, blk, 2.0) for b in range(n)])):
        return f"ipa_adaptive_threshold_u8 n={n} {w}x{h} {meth} {blk}"
    # Canny
    lo, hi = float(rng.uniform(0, 200)), float(rng.uniform(0, 300))
    ws = ws_for(L.ipa_canny_workspace_bytes(w, h, n))
    E = Dev(rng, n, h, w)
    ok(L.ipa_canny_u8(G.ptr, G.pitch, G.stride, E.ptr, E.pitch, E.stride, w, h, n, lo, hi, ws.data_ptr(), ws.numel(), sp), "canny")
    edges = np.stack([fr.canny(gray[b], lo, hi) for b in range(n)])
    if not np.array_equal(E.get(), edges):
        return f"ipa_canny_u8 n={n} {w}x{h} pitches {G.pitch}/{E.pitch}"
    # Hough accumulator
    na, nr = fr.hough_dims(w, h, 1.0, np.pi / 180)
    cells = (na + 2) * (nr + 2)
    astride = cells + int(rng.integers(0, 3)) * 4
    acc = torch.full((n * astride + 16,), -7, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
    ws = ws_for(L.ipa_hough_workspace_bytes(w, h, n, 0))
    ok(L.ipa_hough_accum_u8(E.ptr, E.pitch, E.stride, w, h, n, 1.0, np.pi / 180, acc.data_ptr(), astride, None, 0, cnt.data_ptr(),
                            ws.data_ptr(), ws.numel(), sp), "hough")
    a = acc.cpu().numpy()
    for b in range(n):
        if not np.array_equal(a[b * astride : b * astride + cells].reshape(na + 2, nr + 2), fr.hough_accum(edges[b], 1.0, np.pi / 180)):
            return f"ipa_hough_accum_u8 n={n} {w}x{h} image {b}"
    # segments
    ws = ws_for(L.ipa_segments_workspace_bytes(w, h, n, w * h))
    boxes = torch.zeros((n, 4096, 4), dtype=torch.int32, device="cuda")
    ok(L.ipa_segments_u8(TH.ptr, TH.pitch, TH.stride, w, h, n, w * h, boxes.data_ptr(), 4096, cnt.data_ptr(), ws.data_ptr(), ws.numel(), sp), "segments")
    c, bx = cnt.cpu().numpy(), boxes.cpu().numpy()
    for b in range(n):
        exp = fr.external_boxes(th[b])
        if c[b] != len(exp) or (c[b] <= 4096 and not np.array_equal(bx[b, : c[b]], exp)):
            return f"ipa_segments_u8 n={n} {w}x{h} image {b}"
    # fused pass, five planes of one layout
    P = [Dev(rng, n, h, w) for _ in range(5)]
    for q in P[1:]:
        q.pitch, q.stride, q.off = P[0].pitch, P[0].stride, P[0].off
        q.t = torch.full_like(P[0].t, 0xAB)
    hough = rng.random() < 0.7
    tv = int(rng.integers(0, 256))
    ws = ws_for(L.ipa_front_end_workspace_bytes(w, h, n, 0 if hough else 1))
    acc.fill_(-7)
    bpw = int(L.ipa_bits_pitch_words(w))
    line_bits = torch.full((2, n, h, bpw), -1, dtype=torch.int32, device="cuda") if rng.random() < 0.5 else None
    fe_args = (S.ptr, S.pitch, S.stride, P[0].ptr, P[1].ptr, P[2].ptr, P[3].ptr, P[4].ptr, P[0].pitch, P[0].stride,
               acc.data_ptr() if hough else None, astride, None, 0 if hough else 1, cnt.data_ptr(), w, h, n, tv, 15, 15, 2,
               lo, hi, 0, ws.data_ptr(), ws.numel(), sp)
    if line_bits is None:
        ok(L.ipa_front_end_u8(*fe_args), "front_end")
    else:
        ok(L.ipa_front_end_bits_u8(*fe_args, line_bits[0].data_ptr(), line_bits[1].data_ptr()), "front_end_bits")
    a = acc.cpu().numpy()
    for b in sorted(set([0, n - 1])):
        exp = oracle_case(bgr[b], tv, 15, 15, 2, lo, hi, False, hough, True)
        for k, name in enumerate(("gray", "thresh", "vlines", "hlines", "edges")):
            if not np.array_equal(P[k].get()[b], exp[k]):
                return f"ipa_front_end_u8 n={n} {w}x{h} image {b} plane {name} pitch {P[0].pitch} stride {P[0].stride}"
        if hough and not np.array_equal(a[b * astride : b * astride + cells].reshape(na + 2, nr + 2), exp[5]):
            return f"ipa_front_end_u8 n={n} {w}x{h} image {b} accum"
    if line_bits is not None:  # the segment pass straight from the bit planes of the two opens
        cnt2 = torch.zeros(n, dtype=torch.int32, device="cuda")
        for which, k in ((0, 2), (1, 3)):
            ws2 = ws_for(L.ipa_segments_workspace_bytes(w, h, n, w * h))
            ok(L.ipa_segments_bits_u32(line_bits[which].data_ptr(), bpw, bpw * h, w, h, n, w * h, boxes.data_ptr(), 4096, cnt2.data_ptr(),
                                       ws2.data_ptr(), ws2.numel(), sp), "segments_bits")
            c2, bx2 = cnt2.cpu().numpy(), boxes.cpu().numpy()
            masks = P[k].get()
            for b in range(n):
                exp2 = fr.external_boxes(masks[b])
                if c2[b] != len(exp2) or (c2[b] <= 4096 and not np.array_equal(bx2[b, : c2[b]], exp2)):
                    return f"ipa_segments_bits_u32 n={n} {w}x{h} plane {k} image {b}"
    # nothing outside the images' rows may change except row tails up to the pitch
    for d in (G, E, P[0]):
        flat = d.t.cpu().numpy()
        if not (flat[: d.off] == 0xAB).all() or not (flat[d.off + d.n * d.stride + 0:][-4000:] == 0xAB).all():
            return f"write outside the destination n={n} {w}x{h}"
    return None


def threads_case(rng):
    """Fused passes (chunked and not), HoughLinesP batches and segment passes issued at the same time
    from four host threads on streams of their own."""
    import threading

    import torch

    w, h = pick_dim(rng, 300), pick_dim(rng, 200)
    jobs = []
    for t in range(4):
        n = int(rng.choice([1, 5, 64, 70, 130]))
        while n > 1 and n * w * h > 3_000_000:
            n = max(1, n // 2)
        jobs.append((diagram(rng, n, h, w), int(rng.integers(0, 256)), bool(rng.random() < 0.3), bool(rng.random() < 0.8)))
    out, errors = [None] * 4, []

    def work(t):
        imgs, tv, blur, hough = jobs[t]
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(4):
                    r = ipa.front_end_batch(imgs, thresh_val=tv, blur=blur, hough=hough)
                    segs = ipa.segments_batch(r.hlines)
                    out[t] = (r.gray, r.thresh, r.vlines, r.hlines, r.edges, r.accum.cpu().numpy() if hough else None, segs)
        except Exception as ex:
            errors.append(f"{type(ex).__name__}: {ex}")

    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    ipa.set_option("pipeline_chunks", int(rng.choice([-1, 2, 3])))
    for x in th:
        x.start()
    for x in th:
        x.join()
    if errors:
        return "; ".join(errors)
    for t in range(4):
        imgs, tv, blur, hough = jobs[t]
        n = len(imgs)
        for b in sorted(set([0, n // 2, n - 1])):
            exp = oracle_case(imgs[b], tv, 15, 15, 2, 50, 150, blur, hough, True)
            for k in range(6):
                if exp[k] is not None and not np.array_equal(out[t][k][b], exp[k]):
                    return f"thread {t} n={n} {w}x{h} image {b} output {k}"
            if not np.array_equal(np.asarray(out[t][6][b], dtype=np.int64).reshape(-1, 4), fr.external_boxes(exp[3])):
                return f"thread {t} n={n} {w}x{h} image {b} segments"
    return None


def main_ops(cases, seed0, case=ops_case, base=99000, name="ops"):
    bad = 0
    t_start = time.time()
    for c in range(cases):
        rng = np.random.default_rng(base + seed0 + c)
        try:
            what = case(rng)
        except Exception as ex:
            what = f"{type(ex).__name__}: {ex}"
        if what:
            bad += 1
            print(f"FAIL {name} case {seed0 + c}: {what}", flush=True)
        if (c + 1) % 25 == 0:
            print(f"{c + 1} cases, {bad} bad, {time.time() - t_start:.0f} s", flush=True)
    print(f"fuzz {name}: {cases} cases from seed {seed0}, {bad} bad")
    return 1 if bad else 0


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "threads":
        return main_ops(int(sys.argv[2]) if len(sys.argv) > 2 else 50, int(sys.argv[3]) if len(sys.argv) > 3 else 0, threads_case, 11000, "threads")
    if len(sys.argv) > 1 and sys.argv[1] in ("ops", "batch", "abi"):
        n, s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100, int(sys.argv[3]) if len(sys.argv) > 3 else 0
        if sys.argv[1] == "ops":
            return main_ops(n, s0)
        return main_ops(n, s0, batch_case, 55000, "batch") if sys.argv[1] == "batch" else main_ops(n, s0, abi_case, 33000, "abi")
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    t_start = time.time()
    for c in range(cases):
        rng = np.random.default_rng(77000 + seed0 + c)
        w, h = pick_dim(rng, 1100), pick_dim(rng, 700)
        if rng.random() < 0.04:  # ribbons
            long, short = int(rng.choice([4097, 16384, 32769, 65535, int(rng.integers(5000, 65536))])), int(rng.integers(1, 9))
            w, h = (long, short) if rng.random() < 0.5 else (short, long)
        n = int(rng.choice([1, 1, 2, 3, 5, 63, 64, 65, 70, 96, 128, 131]))
        while n > 1 and n * w * h > 24_000_000:
            n = max(1, n // 2)
        t = int(rng.integers(0, 256))
        vl, hl, it = int(rng.integers(1, 40)), int(rng.integers(1, 40)), int(rng.integers(1, 4))
        lo, hi = float(rng.uniform(0, 300)), float(rng.uniform(0, 400))
        blur, hough = bool(rng.random() < 0.3), bool(rng.random() < 0.7)
        edges = hough or bool(rng.random() < 0.7)
        imgs = diagram(rng, n, h, w)
        # schedule of the fused pass: one chain, or 2..8 chunks (phased or staggered) whatever the size
        chunks = int(rng.choice([-1, 1, 2, 3, 4, 8])) if n >= 2 else -1
        phased = int(rng.random() < 0.7)
        tag = (f"case {seed0 + c}: n={n} {w}x{h} t={t} v={vl} h={hl} it={it} canny=({lo:.1f},{hi:.1f}) blur={blur} hough={hough} "
               f"edges={edges} chunks={chunks} phased={phased}")
        try:
            ipa.set_option("pipeline_chunks", chunks)
            ipa.set_option("phased", phased)
            ipa.set_option("canny_bands", int(rng.choice([-1, 16, 64, 64])))
            ipa.set_option("canny_queue", int(rng.random() < 0.8))
            # the kernel variants behind options (whole-line Canny kernel, gray conversion on spare vote waves) and
            # the number of hysteresis region launches before the finishing kernel
            ipa.set_option("canny_wide", int(rng.choice([0, 0, 4, 8, -1])))
            ipa.set_option("vote_gray", int(rng.choice([0, 0, 1, 2])))
            ipa.set_option("hyst_rounds", int(rng.choice([6, 6, 1, 2])))
            r = ipa.front_end_batch(imgs, thresh_val=t, vert_len=vl, horiz_len=hl, iterations=it, canny=(lo, hi),
                                    blur=blur, hough=hough, edges=edges)
            acc = r.accum.cpu().numpy() if r.accum is not None else None
            check = sorted(set([0, n - 1, int(rng.integers(0, n)), int(rng.integers(0, n))]))
            for b in check:
                exp = oracle_case(imgs[b], t, vl, hl, it, lo, hi, blur, hough, edges)
                got = (r.gray[b], r.thresh[b], r.vlines[b], r.hlines[b], r.edges[b] if edges else None,
                       acc[b] if acc is not None else None)
                for name, a, e in zip(("gray", "thresh", "vlines", "hlines", "edges", "accum"), got, exp):
                    if e is None:
                        continue
                    if not np.array_equal(np.asarray(a), e):
                        bad += 1
                        print(f"FAIL {tag} image {b} plane {name}: {int((np.asarray(a) != e).sum())} cells differ", flush=True)
                        break
        except Exception as ex:  # a raised error is a failure of the case, the run goes on
            bad += 1
            print(f"ERROR {tag}: {type(ex).__name__}: {ex}", flush=True)
        if (c + 1) % 25 == 0:
            print(f"{c + 1} cases, {bad} bad, {time.time() - t_start:.0f} s", flush=True)
    print(f"fuzz: {cases} cases from seed {seed0}, {bad} bad")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
