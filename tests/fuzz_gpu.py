"""Randomised fused-front-end cases against the CPU restatement (test infrastructure, not collected by
pytest): python tests/fuzz_gpu.py [cases] [first seed] for the fused pass, python tests/fuzz_gpu.py ops
[cases] [first seed] for every single operation, python tests/fuzz_gpu.py batch [cases] [first seed] for
the batched helpers, strided / device inputs, rectangle erase and large shapes.  Shapes are biased towards the tile, strip and
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
        got = cv2.morphologyEx(src, op, np.ones((kh_, kw_), np.uint8), iterations=it)
        if not np.array_equal(got, fr.morphology_ex(src, op, (kw_, kh_), it)):
            return f"morphologyEx {w}x{h} op={op} k=({kw_},{kh_}) it={it} binary={src is binary}"
    kx, ky = 2 * int(rng.integers(0, 8)) + 1, 2 * int(rng.integers(0, 8)) + 1
    sg = 0.0 if rng.random() < 0.5 else float(rng.uniform(0.3, 4.0))
    if not np.array_equal(cv2.GaussianBlur(g, (kx, ky), sg), fr.gaussian_blur(g, (kx, ky), sg)):
        return f"GaussianBlur {w}x{h} ({kx},{ky}) {sg}"
    blk, cc = 2 * int(rng.integers(1, 21)) + 1, float(rng.uniform(-10, 10))
    meth, typ = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    if not np.array_equal(cv2.adaptiveThreshold(g, 255, meth, typ, blk, cc), fr.adaptive_threshold(g, 255, meth, typ, blk, cc)):
        return f"adaptiveThreshold {w}x{h} {meth} {typ} {blk} {cc}"
    lo, hi = float(rng.uniform(0, 300)), float(rng.uniform(0, 300))
    e = cv2.Canny(g, lo, hi)
    if not np.array_equal(e, fr.canny(g, lo, hi)):
        return f"Canny {w}x{h} {lo} {hi}"
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
    if len(sys.argv) > 1 and sys.argv[1] in ("ops", "batch"):
        n, s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100, int(sys.argv[3]) if len(sys.argv) > 3 else 0
        return main_ops(n, s0) if sys.argv[1] == "ops" else main_ops(n, s0, batch_case, 55000, "batch")
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    t_start = time.time()
    for c in range(cases):
        rng = np.random.default_rng(77000 + seed0 + c)
        w, h = pick_dim(rng, 1100), pick_dim(rng, 700)
        n = int(rng.choice([1, 1, 2, 3, 5, 63, 64, 65, 70, 96, 128, 131]))
        while n > 1 and n * w * h > 24_000_000:
            n = max(1, n // 2)
        t = int(rng.integers(0, 256))
        vl, hl, it = int(rng.integers(1, 40)), int(rng.integers(1, 40)), int(rng.integers(1, 4))
        lo, hi = float(rng.uniform(0, 300)), float(rng.uniform(0, 400))
        blur, hough = bool(rng.random() < 0.3), bool(rng.random() < 0.7)
        edges = hough or bool(rng.random() < 0.7)
        imgs = diagram(rng, n, h, w)
        tag = f"case {seed0 + c}: n={n} {w}x{h} t={t} v={vl} h={hl} it={it} canny=({lo:.1f},{hi:.1f}) blur={blur} hough={hough} edges={edges}"
        try:
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
